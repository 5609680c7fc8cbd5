"""GPU: the dZ tensor a 3x3 dgrad stores as a side effect of its staging (HpfgConvArgs.dz_out) and the weight gradient that reads it.

The separate dgrad of a channel-rich layer derives dZ = k1*g + k2*z + k3 for every pixel it stages; with dz_out the workgroups of
output-channel slice 0 store it, and the layer's weight gradient reads that one fp32 tensor as a PLAIN source instead of deriving dZ from
(dA, z) again in every (input-channel slice) workgroup.  Same arithmetic either way, so every parameter gradient must agree with the
path that re-derives dZ to rounding of the bf16 split (the two kernels contract their FMAs independently), and the stored tensor must be
the dZ the oracle's BatchNorm / LeakyReLU / Dropout backward gives."""
import pytest
import torch

from hpfg_amd.datasets.synthetic import synth_batch
from hpfg_amd.model import UNet, reset_dropout_streams
from hpfg_amd.utils.loss import Med_Sup_Loss

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _grads(dz_side, n, hw):
    reset_dropout_streams()
    torch.manual_seed(11)
    m = UNet(1, 4).to(DEV)
    m.train()
    x, lab = synth_batch(5, n, hw, hw, 1, 4, cell=8)
    out = m(x.to(DEV))
    eng = next(iter(m._engines.values()))[0]
    eng.dz_side = dz_side
    Med_Sup_Loss(4)(out, lab.to(DEV)).backward()
    torch.cuda.synchronize()
    return {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}, eng


@pytest.mark.parametrize("n,hw", [(2, 48), (3, 80)])
def test_weight_gradients_from_the_stored_dz(n, hw):
    g1, eng = _grads(True, n, hw)
    assert len(eng.dzbuf) >= 6, sorted(eng.dzbuf)          # every 3x3 layer below the 16-pixel-aligned level took the path
    g0, eng0 = _grads(False, n, hw)
    assert not eng0.dzbuf
    for k in g0:
        d = float((g1[k] - g0[k]).abs().max())
        assert d <= 2e-6 * max(1.0, float(g0[k].abs().max())), (k, d)
    # the stored tensor is dense, finite and covers every pixel (an unwritten pixel would keep the NaN pattern the buffer is created with below)
    for name, t in eng.dzbuf.items():
        assert torch.isfinite(t).all(), name


def test_every_pixel_of_dz_is_written():
    """Fill the side buffers with NaN, run one more backward: no NaN may survive (tiles partition the image; only slice-0 workgroups store)."""
    reset_dropout_streams()
    torch.manual_seed(3)
    m = UNet(1, 4).to(DEV)
    m.train()
    x, lab = synth_batch(9, 2, 112, 112, 1, 4, cell=8)
    for _ in range(2):
        out = m(x.to(DEV))
        eng = next(iter(m._engines.values()))[0]
        for t in eng.dzbuf.values():
            t.fill_(float("nan"))
        m.zero_grad(set_to_none=True)
        Med_Sup_Loss(4)(out, lab.to(DEV)).backward()
    torch.cuda.synchronize()
    assert eng.dzbuf
    for name, t in eng.dzbuf.items():
        assert torch.isfinite(t).all(), name
    for k, p in m.named_parameters():
        assert torch.isfinite(p.grad).all(), k
