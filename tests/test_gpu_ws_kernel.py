"""GPU: the experimental warp-specialised conv kernel (HPFG_CONV_WS bit mask, read once per process) must reproduce the default
kernels.  One child process runs a U-Net forward/backward with every layer class routed through conv_ws_kernel and hands its
logits / gradients back through a file; the parent computes the same with the default kernels."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import sys, torch
sys.path.insert(0, {root!r})
from hpfg_amd.model import UNet, reset_dropout_streams
torch.manual_seed(11)
reset_dropout_streams()
m = UNet(1, 4).cuda(); m.train(); m.math = "bf16x3"
x = torch.randn(4, 1, 64, 64, device="cuda")
out = m(x); out.square().mean().backward()
torch.save({{"logits": out.detach().cpu(), "grads": m.flat_grads.detach().cpu()}}, sys.argv[1])
"""


def test_ws_kernel_matches_default(tmp_path):
    path = str(tmp_path / "ws.pt")
    env = dict(os.environ, HPFG_CONV_WS="15")
    r = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT), path], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    ws = torch.load(path)
    from hpfg_amd.model import UNet, reset_dropout_streams
    torch.manual_seed(11)
    reset_dropout_streams()          # the child's network is the first of its process: same dropout stream here
    m = UNet(1, 4).cuda()
    m.train()
    m.math = "bf16x3"
    x = torch.randn(4, 1, 64, 64, device="cuda")
    out = m(x)
    out.square().mean().backward()
    # same tiles, same k order, same fp32 accumulation: the two kernels agree to rounding of the BatchNorm partial-sum order
    assert float((out.detach().cpu() - ws["logits"]).abs().max()) < 1e-4
    g = m.flat_grads.detach().cpu()
    assert float((g - ws["grads"]).norm()) < 1e-3 * float(g.norm())
