"""GPU: the tensors a 3x3 conv / dgrad stores as a side effect of its staging (HpfgConvArgs.stage_out) and the weight gradient that reads them.

The separate dgrad of a channel-rich layer derives dZ = k1*g + k2*z + k3 for every pixel it stages, its forward conv the virtual input
(BatchNorm + LeakyReLU + Dropout, max-pooled or upsampled + concatenated); with stage_out the workgroups of output-channel slice 0 store
what they stage -- as the (hi | lo) bf16 pair of the split-precision products -- and the layer's weight gradient reads the two tensors as
SPLIT16 sources (a copy into LDS) instead of deriving and splitting both again in every
(input slice x output slice) workgroup.  Same arithmetic either way, so every parameter gradient must agree with the path that re-derives
them to rounding (the kernels contract their FMAs independently), the stored input must be the virtual input hpfg_act_materialize gives,
and every pixel must be written."""
import pytest
import torch

from hpfg_amd.datasets.synthetic import synth_batch
from hpfg_amd.model import UNet, reset_dropout_streams
from hpfg_amd.utils.loss import Med_Sup_Loss

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _grads(side, n, hw):
    reset_dropout_streams()
    torch.manual_seed(11)
    m = UNet(1, 4).to(DEV)
    m.train()
    x, lab = synth_batch(5, n, hw, hw, 1, 4, cell=8)
    with torch.no_grad():
        m(x.to(DEV))          # (creates the engine; both variants draw the same masks and move the running statistics alike)
    eng = next(iter(m._engines.values()))[0]
    eng.dz_side = eng.act_side = side
    out = m(x.to(DEV))
    Med_Sup_Loss(4)(out, lab.to(DEV)).backward()
    torch.cuda.synchronize()
    return {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}, eng


@pytest.mark.parametrize("n,hw", [(2, 48), (3, 80)])
def test_weight_gradients_from_the_stored_dz(n, hw):
    g1, eng = _grads(True, n, hw)
    assert len(eng.dzbuf) >= 6 and len(eng.actbuf) >= 6, (sorted(eng.dzbuf), sorted(eng.actbuf))      # every 3x3 layer below the aligned levels
    for name, t in eng.actbuf.items():          # the stored pair == the split of the layer's virtual input (pooled / concatenated sources included)
        ref = eng.materialize_input(name)
        hi, lo = _pair(t, ref.shape)
        rhi = ref.to(torch.bfloat16)
        rlo = (ref - rhi.float()).to(torch.bfloat16)          # the host's split of hpfg_act_materialize: hi = bf16(v), lo = bf16(v - hi)
        # (the staging loader and the materialize kernel contract their FMAs independently: a value one ulp apart may round to the other bf16
        # neighbour -- so "equal" is exact for all but a handful of elements, and the pair always represents the value to 2^-16)
        same = ((hi == rhi) & (lo == rlo)).float().mean()
        assert float(same) >= 0.98 and float((hi == rhi).float().mean()) >= 0.999, (name, float(same), float((hi == rhi).float().mean()))
        err = (hi.float() + lo.float() - ref).abs()
        assert bool((err <= 2.0 ** -15 * ref.abs() + 1e-6 * max(1.0, float(ref.abs().max()))).all()), (name, float(err.max()))
    g0, eng0 = _grads(False, n, hw)
    assert not eng0.dzbuf and not eng0.actbuf
    for k in g0:
        d = float((g1[k] - g0[k]).abs().max())
        assert d <= 2e-6 * max(1.0, float(g0[k].abs().max())), (k, d)
    # the stored tensor is dense, finite and covers every pixel (an unwritten pixel would keep the NaN pattern the buffer is created with below)
    for name, t in eng.dzbuf.items():
        hi, lo = _pair(t, t.shape)
        assert torch.isfinite(hi.float()).all() and torch.isfinite(lo.float()).all(), name


def _pair(t, shape):
    """(hi, lo) bf16 tensors [N,H,W,C] of a side tensor: HpfgConvArgs.stage_out stores bf16 [N][H][W][C / 8][hi 8 | lo 8] in a buffer of N*H*W*C fp32 words."""
    n, h, w, c = shape
    p = t.view(torch.bfloat16).reshape(n, h, w, c // 8, 2, 8)
    return p[..., 0, :].reshape(n, h, w, c), p[..., 1, :].reshape(n, h, w, c)


def test_every_pixel_of_the_side_tensors_is_written():
    """Fill the side buffers with NaN, run one more forward + backward: no NaN may survive (tiles partition the image; only slice-0 workgroups store)."""
    reset_dropout_streams()
    torch.manual_seed(3)
    m = UNet(1, 4).to(DEV)
    m.train()
    x, lab = synth_batch(9, 2, 112, 112, 1, 4, cell=8)
    for it in range(2):
        if it == 1:          # the buffers exist now: poison them, the second forward / backward must overwrite every element
            eng = next(iter(m._engines.values()))[0]
            for t in list(eng.dzbuf.values()) + list(eng.actbuf.values()):
                t.view(torch.bfloat16).fill_(float("nan"))
        out = m(x.to(DEV))
        m.zero_grad(set_to_none=True)
        Med_Sup_Loss(4)(out, lab.to(DEV)).backward()
    torch.cuda.synchronize()
    assert eng.dzbuf and eng.actbuf
    for name, t in list(eng.dzbuf.items()) + list(eng.actbuf.items()):
        assert torch.isfinite(t.view(torch.bfloat16).float()).all(), name
    for k, p in m.named_parameters():
        assert torch.isfinite(p.grad).all(), k


@pytest.mark.parametrize("n,hw", [(2, 48), (2, 112)])
def test_max_pool_backward_in_the_dgrad_epilogue(n, hw):
    """HpfgConvArgs.bwd_stats == 2: the dgrad that would produce dP adds it at the window's arg-max into the gradient of the block output below
    and takes that layer's BatchNorm-backward sums in its epilogue (hpfg_bn_bwd_reduce_pool without its launch).  Same adds on the same values:
    every parameter gradient agrees with the separate-launch path to the rounding of the differently ordered sums."""
    def run(fuse):
        reset_dropout_streams()
        torch.manual_seed(21)
        m = UNet(1, 4).to(DEV)
        m.train()
        x, lab = synth_batch(6, n, hw, hw, 1, 4, cell=8)
        with torch.no_grad():
            m(x.to(DEV))
        eng = next(iter(m._engines.values()))[0]
        eng.pool_fuse = fuse
        out = m(x.to(DEV))
        Med_Sup_Loss(4)(out, lab.to(DEV)).backward()
        torch.cuda.synchronize()
        return {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}, eng

    g1, e1 = run(True)
    assert len(e1._pool_done) == 4, e1._pool_done          # every pooled level here (at 224 x 224 the 112-pixel level runs the fused thin kernel: 3)
    g0, e0 = run(False)
    assert not e0._pool_done
    for k in g0:
        d = float((g1[k] - g0[k]).abs().max())
        assert d <= 5e-5 * max(1e-3, float(g0[k].abs().max())), (k, d, float(g0[k].abs().max()))


@pytest.mark.parametrize("n,hw", [(2, 48), (2, 112), (3, 80)])
def test_upsample_backward_gathered_by_the_1x1_dgrad(n, hw):
    """HPFG_ACT_UPBWD: the dgrad of a decoder block's 1x1 conv evaluates the transposed bilinear interpolation while it stages its tiles (the
    taps and the order of additions of hpfg_upsample2x_bwd), stores the gathered gradient for the 1x1 conv's weight gradient and leaves the
    bias-gradient rows -- instead of the separate launch in front of it.  Every parameter gradient must agree with the two-launch path."""
    def run(fuse):
        reset_dropout_streams()
        torch.manual_seed(31)
        m = UNet(1, 4).to(DEV)
        m.train()
        x, lab = synth_batch(7, n, hw, hw, 1, 4, cell=8)
        with torch.no_grad():
            m(x.to(DEV))
        eng = next(iter(m._engines.values()))[0]
        eng.upb_fuse = fuse
        out = m(x.to(DEV))
        Med_Sup_Loss(4)(out, lab.to(DEV)).backward()
        torch.cuda.synchronize()
        return {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}, {k: v.detach().cpu().clone() for k, v in eng.dU.items()}

    g1, u1 = run(True)
    g0, u0 = run(False)
    for k in u0:          # the gathered gradient itself: the same taps in the same order (the two kernels contract their FMAs independently)
        d = float((u1[k] - u0[k]).abs().max())
        assert d <= 1e-6 * max(1e-6, float(u0[k].abs().max())), (k, d, float(u0[k].abs().max()))
    for k in g0:
        d = float((g1[k] - g0[k]).abs().max())
        assert d <= 5e-6 * max(1e-4, float(g0[k].abs().max())), (k, d, float(g0[k].abs().max()))
