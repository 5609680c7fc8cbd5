"""GPU: forward_pair() -- student and teacher layer by layer in shared launches (hpfg_conv_fwd_pair, hpfg_conv3x3_first_fwd_pair,
hpfg_bn_fwd_finalize_pair) -- is bit-identical to the two separate forward calls, BatchNorm running statistics and backward included."""
import copy

import pytest
import torch

from hpfg_amd.model import UNet
from hpfg_amd.model.unet import can_pair, forward_pair, reset_dropout_streams

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _models(seed):
    reset_dropout_streams()
    torch.manual_seed(seed)
    s = UNet(1, 4).to(DEV)
    t = copy.deepcopy(s)
    with torch.no_grad():
        for p in t.parameters():
            p.mul_(0.9)
            p.requires_grad_(False)
    s.train()
    t.train()
    return s, t


@pytest.mark.parametrize("N,H", [(3, 64), (2, 48)])
def test_pair_forward_equals_two_forwards_bitwise(N, H, monkeypatch):
    monkeypatch.setenv("HPFG_PAIR_FWD", "1")      # (off by default: measured slower than the two-stream overlap it replaces)
    monkeypatch.setenv("HPFG_CONV_THIN", "0")     # the paired launches run conv_bf16x3_kernel: compare like with like (BatchNorm sum rows differ otherwise)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, 1, H, H, generator=g).to(DEV)
    dy = torch.randn(N, 4, H, H, generator=g).to(DEV)
    # separate calls
    s0, t0 = _models(1)
    with torch.no_grad():
        t_ref = t0(x)
    o_ref = s0(x)
    o_ref.backward(dy)
    # shared launches
    s1, t1 = _models(1)
    assert can_pair(s1, t1, x, x)
    o, t_out = forward_pair(s1, x, t1, x)
    o.backward(dy)
    assert not t_out.requires_grad
    assert torch.equal(o, o_ref) and torch.equal(t_out, t_ref)
    for (n0, b0), (n1, b1) in zip(list(s0.named_buffers()) + list(t0.named_buffers()), list(s1.named_buffers()) + list(t1.named_buffers())):
        assert torch.equal(b0, b1), n0
    for (n0, p0), (n1, p1) in zip(s0.named_parameters(), s1.named_parameters()):
        assert torch.equal(p0.grad, p1.grad), n0


def test_can_pair_refuses_other_combinations(monkeypatch):
    monkeypatch.setenv("HPFG_PAIR_FWD", "1")
    s, t = _models(2)
    x = torch.randn(2, 1, 32, 32, device=DEV)
    assert can_pair(s, t, x, x)
    assert not can_pair(s, s, x, x)
    t.eval()
    assert not can_pair(s, t, x, x)
    t.train()
    t.math = "f32"
    assert not can_pair(s, t, x, x)
    assert not can_pair(s, t, x, x[:1])
    t.math = "bf16x3"
    monkeypatch.setenv("HPFG_PAIR_FWD", "0")
    assert not can_pair(s, t, x, x)


def test_mean_teacher_step_with_paired_forward_equals_the_two_stream_step(monkeypatch):
    """The Mean-Teacher step with HPFG_PAIR_FWD=1 follows the same trajectory, bit for bit, as the default two-stream step."""
    from copy import deepcopy

    from hpfg_amd.train import MeanTeacherStep
    from hpfg_amd.datasets.synthetic import synth_batch
    from tests.test_gpu_dp_path import _args

    def run():
        torch.manual_seed(7)
        reset_dropout_streams()
        m = UNet(1, 4).to(DEV)
        ema = deepcopy(m)
        for p in ema.parameters():
            p.requires_grad = False
        m.train()
        ema.train()
        st = MeanTeacherStep(m, ema, _args())
        xl, yl = synth_batch(1, 2, 64, 64, 1, 4, 8)
        xu, _ = synth_batch(2, 2, 64, 64, 1, 4, 8)
        xl, yl, xu = xl.to(DEV), yl.to(DEV), xu.to(DEV)
        losses = [float(st.step(xl, yl, xu, k, cons_w=0.05)["loss"]) for k in range(1, 4)]
        return losses, m.flat_params.clone(), ema.flat_params.clone()

    monkeypatch.setenv("HPFG_CONV_THIN", "0")
    monkeypatch.setenv("HPFG_PAIR_FWD", "0")
    l0, p0, e0 = run()
    monkeypatch.setenv("HPFG_PAIR_FWD", "1")
    l1, p1, e1 = run()
    assert l0 == l1 and torch.equal(p0, p1) and torch.equal(e0, e1)


def test_mean_teacher_step_with_interleaved_launches_equals_the_default_step(monkeypatch):
    """HPFG_INTERLEAVE_FWD=1 (student and teacher on two streams, launches issued layer by layer in turn; UNetEngine.forward_interleaved)
    runs the same kernels as the default step: same trajectory bit for bit, eager and captured."""
    from copy import deepcopy

    from hpfg_amd.train import GraphedStep, MeanTeacherStep
    from hpfg_amd.datasets.synthetic import synth_batch
    from tests.test_gpu_dp_path import _args

    def run(graph):
        torch.manual_seed(7)
        reset_dropout_streams()
        m = UNet(1, 4).to(DEV)
        ema = deepcopy(m)
        for p in ema.parameters():
            p.requires_grad = False
        m.train()
        ema.train()
        st = MeanTeacherStep(m, ema, _args())
        xl, yl = synth_batch(1, 2, 64, 64, 1, 4, 8)
        xu, _ = synth_batch(2, 2, 64, 64, 1, 4, 8)
        xl, yl, xu = xl.to(DEV), yl.to(DEV), xu.to(DEV)
        if graph:
            r = GraphedStep(st, [xl, yl, xu], warmup=2)
            losses = [float(r.step([xl, yl, xu], k)["loss"]) for k in range(3, 6)]
        else:
            losses = [float(st.step(xl, yl, xu, k, cons_w=0.05)["loss"]) for k in range(1, 4)]
        torch.cuda.synchronize()
        return losses, m.flat_params.clone(), ema.flat_params.clone()

    for graph in (False, True):
        monkeypatch.setenv("HPFG_INTERLEAVE_FWD", "0")
        l0, p0, e0 = run(graph)
        monkeypatch.setenv("HPFG_INTERLEAVE_FWD", "1")
        l1, p1, e1 = run(graph)
        assert l0 == l1 and torch.equal(p0, p1) and torch.equal(e0, e1), graph
