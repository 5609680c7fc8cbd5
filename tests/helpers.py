"""Shared helpers for the parity tests: run the oracle with exactly the dropout masks the HIP kernels used."""
import numpy as np
import torch

from hpfg_amd import engine as E
from oracle import rng_ref, unet_ref


def state_from_module(m):
    """Oracle state dict (CPU clones) from an hpfg_amd module."""
    return {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}


def engine_masks(eng, seed_step, n, h, w):
    """Keep-masks (NCHW float) of the five encoder dropout sites for a forward that used `seed_step`."""
    masks = []
    for lvl in range(5):
        s = eng.specs[E.enc_prefix(lvl) + ".0"]
        seed = (eng.layer_seed(s) + seed_step) & 0xFFFFFFFF
        masks.append(torch.from_numpy(rng_ref.keep_mask_nchw(n, s.cout, s.h, s.w, s.drop_p, seed).astype(np.float32)))
    return masks


def nchw(t_nhwc):
    return t_nhwc.permute(0, 3, 1, 2).contiguous()


def maxerr(a, b):
    return float((a.double() - b.double()).abs().max())


# ---- ad-hoc single-layer harness around the C ABI (unit tests of conv / dgrad / wgrad) ----------------------------------
import ctypes as C

from hpfg_amd import _lib as L


def pad16(c):
    return (c + 15) // 16 * 16


class AdHocConv:
    def __init__(self, cin, cout, taps, device, seed=0, hw=(16, 16)):
        g = torch.Generator().manual_seed(seed)
        k = 3 if taps == 9 else 1
        self.cin, self.cout, self.taps, self.k, self.dev = cin, cout, taps, k, device
        self.w = (torch.randn(cout, cin, k, k, generator=g) * 0.2).to(device)
        self.b = torch.randn(cout, generator=g).to(device)
        self.cin_pad, self.cout_pad = pad16(cin), pad16(cout)
        n = taps * self.cin_pad * self.cout_pad
        self.wpk_f = torch.empty(n, device=device)
        self.wpk_d = torch.empty(n, device=device)
        self.bias_pad = torch.empty(self.cout_pad, device=device)
        lib = L.load()
        self.kc = lib.hpfg_conv_kc(hw[0], hw[1], taps)
        self.wpk16_f = torch.empty(lib.hpfg_wpk16_elems(cin, self.cout_pad, taps, self.kc), dtype=torch.bfloat16, device=device)
        self.wpk16_d = torch.empty(lib.hpfg_wpk16_elems(cout, self.cin_pad, taps, self.kc), dtype=torch.bfloat16, device=device)
        d = (L.PackDesc * 1)()
        d[0].wpk16_fwd, d[0].wpk16_dgrad, d[0].kc = L.ptr(self.wpk16_f), L.ptr(self.wpk16_d), self.kc
        d[0].w_oihw, d[0].b, d[0].wpk_fwd, d[0].wpk_dgrad, d[0].bias_pad = (L.ptr(self.w), L.ptr(self.b), L.ptr(self.wpk_f), L.ptr(self.wpk_d),
                                                                          L.ptr(self.bias_pad))
        d[0].Cout, d[0].Cin, d[0].CoutPad, d[0].CinPad, d[0].taps = cout, cin, self.cout_pad, self.cin_pad, taps
        dev_tab = torch.frombuffer(bytearray(bytes(d)), dtype=torch.uint8).to(device)
        self._keep = (d, dev_tab)
        L.check(L.load().hpfg_pack_weights(dev_tab.data_ptr(), d, 1, stream(device)), "pack")

    def conv(self, a0, a1, N, H, W, stats=False, dgrad=False, math=0):
        lib = L.load()
        cout = self.cin if dgrad else self.cout
        cpad = self.cin_pad if dgrad else self.cout_pad
        out = torch.full((N, H, W, cout), float("nan"), device=self.dev)
        ca = L.ConvArgs()
        ca.a0, ca.a1 = a0, (a1 if a1 is not None else L.Act())
        ca.math = math
        if math == L.MATH_BF16X3:
            assert lib.hpfg_conv_kc(H, W, self.taps) == self.kc, "AdHocConv was packed for a different tile class"
            ca.wpk = L.ptr(self.wpk16_d if dgrad else self.wpk16_f)
        else:
            ca.wpk = L.ptr(self.wpk_d if dgrad else self.wpk_f)
        ca.bias = None if dgrad else L.ptr(self.bias_pad)
        ca.out = L.ptr(out)
        part = None
        if stats:
            part = torch.zeros(lib.hpfg_conv_stat_blocks(N, H, W) * 2 * cpad, device=self.dev)   # unused rows stay 0
            ca.stat_partials = L.ptr(part)
        ca.out_pstride, ca.Cout, ca.CoutPad, ca.N, ca.H, ca.W, ca.taps = cout, cout, cpad, N, H, W, self.taps
        L.check(lib.hpfg_conv_fwd(C.byref(ca), stream(self.dev)), "conv_fwd")
        return out, part

    def wgrad(self, a0, a1, g, N, H, W, math=0):
        lib = L.load()
        wa = L.WgradArgs()
        wa.a0, wa.a1, wa.g = a0, (a1 if a1 is not None else L.Act()), g
        S = lib.hpfg_wgrad_splits(N, H, W, self.cin_pad, self.cout_pad, self.taps)
        slab = torch.empty(lib.hpfg_wgrad_slab_floats(N, H, W, self.cin_pad, self.cout_pad, self.taps), device=self.dev)
        dw = torch.full_like(self.w, float("nan"))
        wa.slab, wa.dw_oihw, wa.math = L.ptr(slab), L.ptr(dw), math
        wa.Cin, wa.CinPad, wa.Cout, wa.CoutPad, wa.N, wa.H, wa.W, wa.taps, wa.S = self.cin, self.cin_pad, self.cout, self.cout_pad, N, H, W, self.taps, S
        L.check(lib.hpfg_wgrad(C.byref(wa), stream(self.dev)), "wgrad")
        return dw


def stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def plain_act(t, C_, H, W):
    """PLAIN (or STRIDED when C%4 != 0) source over a contiguous NHWC tensor."""
    a = L.Act()
    a.z, a.mode, a.C, a.Hs, a.Ws, a.pstride = L.ptr(t), L.ACT_PLAIN, C_, H, W, C_
    if C_ % 4:
        a.mode = L.ACT_STRIDED
        a.sn, a.sc, a.sy, a.sx = H * W * C_, 1, W * C_, C_
    return a
