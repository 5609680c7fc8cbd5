"""GPU: the projection necks and Dense_Loss on the HIP library (hpfg_amd/heads.py: hpfg_gemm_f32, hpfg_neck_pool_*, hpfg_l2norm_*,
hpfg_ntxent_rows) against plain PyTorch fp32 restatements of reference model/unet.py:120-152 and utils/loss/dense_loss.py:17-40."""
import pytest
import torch
import torch.nn.functional as F

from hpfg_amd import _lib as L
from hpfg_amd import heads
from hpfg_amd.model.unet import _neck
from hpfg_amd.utils import Dense_Loss
from oracle import losses_ref
from tests.helpers import maxerr

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("m,n,k", [(32, 2048, 256), (512, 128, 2048), (7, 5, 3), (130, 70, 33), (64, 64, 16)])
def test_gemm_f32_all_orientations(m, n, k):
    g = torch.Generator().manual_seed(m * 7 + n)
    x = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g)
    b = torch.randn(n, generator=g)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    ref = x.double() @ w.double().t() + b.double()
    tol = 2e-6 * k * float(ref.abs().max() / k ** 0.5 + 1)
    y = heads.gemm(xd, k, 1, wd, 1, k, m, n, k, bias=bd)                       # X W^T + b
    assert maxerr(y.cpu().double(), ref) < tol
    yr = heads.gemm(xd, k, 1, wd, 1, k, m, n, k, bias=bd, relu=True)
    assert maxerr(yr.cpu().double(), ref.clamp(min=0)) < tol
    dy = torch.randn(m, n, generator=g)
    dyd = dy.to(DEV)
    dx = heads.gemm(dyd, n, 1, wd, k, 1, m, k, n)                               # dY W
    assert maxerr(dx.cpu().double(), dy.double() @ w.double()) < 2e-6 * n * 4
    dw = heads.gemm(dyd, 1, n, xd, k, 1, n, k, m)                               # dY^T X
    assert maxerr(dw.cpu().double(), dy.double().t() @ x.double()) < 2e-6 * m * 4
    db = torch.empty(n, device=DEV)
    L.check(L.load().hpfg_col_sum(L.ptr(dyd), m, n, n, L.ptr(db), torch.cuda.current_stream().cuda_stream), "col_sum")
    assert maxerr(db.cpu().double(), dy.double().sum(0)) < 1e-5 * m


@pytest.mark.parametrize("n,c,h,w", [(3, 256, 14, 14), (2, 4, 224, 224), (2, 4, 64, 48), (2, 2, 30, 30)])
def test_neck_pool_forward_backward(n, c, h, w):
    g = torch.Generator().manual_seed(c + h)
    x = torch.randn(n, h, w, c, generator=g).permute(0, 3, 1, 2)              # NCHW view of NHWC memory, as the engine hands it over
    xd = x.to(DEV).requires_grad_(True)
    gap, pool = heads._NeckPool.apply(xd, 4)
    xr = x.clone().requires_grad_(True)
    rg = F.adaptive_avg_pool2d(xr, 1).flatten(1)
    rp = F.adaptive_avg_pool2d(xr, 4)
    assert maxerr(gap.detach().cpu(), rg.detach()) < 2e-6
    assert maxerr(pool.detach().cpu().view(n, 16, c).permute(0, 2, 1).reshape(n, c, 4, 4), rp.detach()) < 2e-6
    wg, wp = torch.randn(n, c, generator=g), torch.randn(n, c, 4, 4, generator=g)
    ((rg * wg).sum() + (rp * wp).sum()).backward()
    wpd = wp.reshape(n, c, 16).permute(0, 2, 1).reshape(n * 16, c).to(DEV)
    ((gap * wg.to(DEV)).sum() + (pool * wpd).sum()).backward()
    assert maxerr(xd.grad.cpu(), xr.grad) < 1e-7 + 1e-5 * float(xr.grad.abs().max())


@pytest.mark.parametrize("cin,hid,hw", [(256, 2048, 14), (4, 1024, 224)])
def test_projection_neck_matches_torch_modules(cin, hid, hw):
    """projection_conv.forward (unet.py:139-152) and its backward: outputs, input gradient and all eight parameter gradients."""
    torch.manual_seed(3)
    m = _neck(cin, hid).to(DEV)
    n = 4
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, hw, hw, cin, generator=g).permute(0, 3, 1, 2)
    xd = x.to(DEV).requires_grad_(True)
    gg, dd = heads.projection_neck(m, xd)
    assert gg.shape == (n, 128) and dd.shape == (n, 128, 16)
    wg, wd = torch.randn(n, 128, generator=g).to(DEV), torch.randn(n, 128, 16, generator=g).to(DEV)
    ((gg * wg).sum() + (dd * wd).sum()).backward()
    got = {k: p.grad.clone() for k, p in m.named_parameters()}
    gx = xd.grad.clone()
    for p in m.parameters():
        p.grad = None
    xr = x.to(DEV).requires_grad_(True)
    rg = F.adaptive_avg_pool2d(xr, 1).flatten(1)
    rg = F.linear(F.relu(F.linear(rg, m.mlp["0"].weight, m.mlp["0"].bias)), m.mlp["2"].weight, m.mlp["2"].bias)
    rd = F.adaptive_avg_pool2d(xr, 4)
    rd = F.conv2d(F.relu(F.conv2d(rd, m.mlp_conv["0"].weight, m.mlp_conv["0"].bias)), m.mlp_conv["2"].weight, m.mlp_conv["2"].bias).flatten(2)
    assert maxerr(gg.detach().cpu(), rg.detach().cpu()) < 1e-5 and maxerr(dd.detach().cpu(), rd.detach().cpu()) < 1e-5
    ((rg * wg).sum() + (rd * wd).sum()).backward()
    assert maxerr(gx.cpu(), xr.grad.cpu()) < 1e-7 + 2e-5 * float(xr.grad.abs().max())
    for k, p in m.named_parameters():
        assert maxerr(got[k].cpu(), p.grad.cpu()) < 1e-6 + 2e-5 * float(p.grad.abs().max()), k


@pytest.mark.parametrize("n", [4, 8, 32])
def test_dense_loss_matches_reference_law(n):
    """Dense_Loss (dense_loss.py:17-40) value and student gradients against the oracle restatement (pinned to the reference's own
    module by tests/golden/losses.npz), for canonical [N,128,16] tensors and for the position-major views projection_neck produces."""
    g = torch.Generator().manual_seed(n)
    h = (torch.randn(n, 128, generator=g), torch.randn(n, 128, 16, generator=g))
    t = (torch.randn(n, 128, generator=g), torch.randn(n, 128, 16, generator=g))
    hr = tuple(v.clone().requires_grad_(True) for v in h)
    ref = losses_ref.dense_loss(hr, t)
    ref.backward()
    for layout in ("canonical", "position-major"):
        if layout == "canonical":
            hd = tuple(v.to(DEV).requires_grad_(True) for v in h)
            td = tuple(v.to(DEV) for v in t)
        else:
            hd = (h[0].to(DEV).requires_grad_(True), h[1].permute(0, 2, 1).contiguous().to(DEV).permute(0, 2, 1).requires_grad_(True))
            td = (t[0].to(DEV), t[1].permute(0, 2, 1).contiguous().to(DEV).permute(0, 2, 1))
        got = Dense_Loss(n, DEV)(hd, td)
        assert abs(float(got) - float(ref)) < 2e-6, layout
        (3.0 * got).backward()
        for a, b in zip(hd, hr):
            assert maxerr(a.grad.cpu(), 3.0 * b.grad) < 1e-7 + 2e-5 * float(b.grad.abs().max()), layout


def test_losses_fixture_dense_value(golden_dir):
    """The reference's own Dense_Loss output (tests/golden/losses.npz, written from utils/loss/dense_loss.py)."""
    import numpy as np
    d = np.load(f"{golden_dir}/losses.npz")
    hg = (torch.from_numpy(d["hg0"]).to(DEV), torch.from_numpy(d["hg1"]).to(DEV))
    tg = (torch.from_numpy(d["tg0"]).to(DEV), torch.from_numpy(d["tg1"]).to(DEV))
    assert abs(float(Dense_Loss(hg[0].shape[0], DEV)(hg, tg)) - float(d["dense"])) < 1e-5
