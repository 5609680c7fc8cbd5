"""Kernel schedule of one U-Net forward / backward pass on the HIP library.

The engine owns the device workspace for a fixed (N, H, W) problem: the raw (pre-BatchNorm) output of every conv, the
per-layer BatchNorm tables, packed weights, gradient buffers.  It issues the C-ABI calls of include/hpfg_hip.h in the
order of the reference graph (model/unet.py:61-117); no tensor op of the network runs in PyTorch.

Layer naming follows the reference's state_dict keys (e.g. ``encoder.down1.maxpool_conv.1.conv_conv.0``).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from . import _lib as L

SEED_BUMP = -1      # seed_step value: advance the engine's seed word on the device (inside the forward's first launch)

WIDTHS = (16, 32, 64, 128, 256)          # model/unet.py:161
ENC_DROPOUT = (0.05, 0.1, 0.2, 0.3, 0.5)  # model/unet.py:162
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def _pad16(c: int) -> int:
    return (c + 15) // 16 * 16


@dataclass
class ConvSpec:
    name: str            # state_dict prefix of the conv ("….weight", "….bias")
    bn: Optional[str]    # state_dict prefix of its BatchNorm, or None
    cin: int
    cout: int
    taps: int
    h: int               # output height / width
    w: int
    drop_p: float = 0.0  # dropout applied to this layer's activated output
    idx: int = 0
    cin_pad: int = 0
    cout_pad: int = 0

    def __post_init__(self):
        self.cin_pad, self.cout_pad = _pad16(self.cin), _pad16(self.cout)


def enc_prefix(lvl: int) -> str:
    return "encoder.in_conv.conv_conv" if lvl == 0 else f"encoder.down{lvl}.maxpool_conv.1.conv_conv"


def build_specs(in_ch: int, ncls: int, H: int, W: int) -> Dict[str, ConvSpec]:
    specs: Dict[str, ConvSpec] = {}

    def add(s: ConvSpec):
        s.idx = len(specs)
        specs[s.name] = s

    cprev = in_ch
    for lvl, c in enumerate(WIDTHS):
        p, h, w = enc_prefix(lvl), H >> lvl, W >> lvl
        add(ConvSpec(f"{p}.0", f"{p}.1", cprev, c, 9, h, w, ENC_DROPOUT[lvl]))
        add(ConvSpec(f"{p}.4", f"{p}.5", c, c, 9, h, w))
        cprev = c
    for k in range(1, 5):
        c1, c2 = WIDTHS[5 - k], WIDTHS[4 - k]
        hl, wl = H >> (5 - k), W >> (5 - k)
        add(ConvSpec(f"decoder.up{k}.conv1x1", None, c1, c2, 1, hl, wl))
        p = f"decoder.up{k}.conv.conv_conv"
        add(ConvSpec(f"{p}.0", f"{p}.1", 2 * c2, c2, 9, 2 * hl, 2 * wl))
        add(ConvSpec(f"{p}.4", f"{p}.5", c2, c2, 9, 2 * hl, 2 * wl))
    add(ConvSpec("decoder.out_conv", None, WIDTHS[0], ncls, 9, H, W))
    return specs


class MarkLog:
    """Device time stamps around selected launches (bench.py's per-kernel timings).  A stamp is a one-wave kernel that stores s_memrealtime
    (100 MHz) into a slot of `buf` (hpfg_timestamp); stamps are ordinary stream work, so they are captured into a hipGraph with the launches
    they bracket and every replay refreshes them -- no host events, no eager steps.  `spans`: (tag, first slot, last slot) in capture
    order; a ("calib", ...) span brackets nothing: its length (one stamp kernel plus one kernel boundary) is what a bracket costs by itself."""

    def __init__(self, dev: torch.device, capacity: int = 2048):
        self.buf = torch.zeros(capacity, dtype=torch.int64, device=dev)
        self.spans: List[tuple] = []
        self.n = 0
        self.lib = L.load()

    def stamp(self, stream: int) -> int:
        k = self.n
        if k >= self.buf.numel():
            raise RuntimeError("MarkLog: out of slots")
        self.n += 1
        L.check(self.lib.hpfg_timestamp(self.buf.data_ptr() + 8 * k, stream), "timestamp")
        return k

    def bracket(self, tag: str, stream: int, launch):
        a = self.stamp(stream)
        launch()
        self.spans.append((tag, a, self.stamp(stream)))

    def calib(self, stream: int):
        a = self.stamp(stream)
        self.spans.append(("calib", a, self.stamp(stream)))

    def read_us(self) -> List[tuple]:
        """[(tag, microseconds)] of the last execution (synchronises)."""
        t = self.buf[: self.n].cpu().numpy()
        return [(tag, float(int(t[b]) - int(t[a])) / 100.0) for tag, a, b in self.spans]


class UNetEngine:
    """Forward/backward schedule for one (module, N, H, W).  ``params``: name -> tensor views (flat buffers of the module)."""

    def __init__(self, params: Dict[str, torch.Tensor], buffers: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor],
                 in_ch: int, ncls: int, N: int, H: int, W: int, device: torch.device):
        assert H % 16 == 0 and W % 16 == 0, "U-Net input must be a multiple of 16 (4 poolings)"
        self.lib = L.load()
        self.dev = device
        self.N, self.H, self.W, self.in_ch, self.ncls = N, H, W, in_ch, ncls
        self.params, self.buffers, self.grads = params, buffers, grads
        self.specs = build_specs(in_ch, ncls, H, W)
        self.order = list(self.specs.values())
        f32 = dict(dtype=torch.float32, device=device)
        # raw conv outputs (logits are allocated per forward call)
        self.z: Dict[str, torch.Tensor] = {s.name: torch.empty(N, s.h, s.w, s.cout, **f32) for s in self.order if s.name != "decoder.out_conv"}
        self.bn: Dict[str, torch.Tensor] = {s.name: torch.zeros(L.BN_ROWS, s.cout, **f32) for s in self.order if s.bn}
        nb = max(self.lib.hpfg_conv_stat_blocks(N, s.h, s.w) * 2 * s.cout_pad for s in self.order if s.bn)
        nb = max(nb, max(self.lib.hpfg_bn_bwd_blocks(N, s.h, s.w, s.cout) * 2 * s.cout for s in self.order if s.bn))
        nb = max(nb, max(self.lib.hpfg_bn_bwd_pool_blocks(N, s.h // 2, s.w // 2, s.cout) * 2 * s.cout for s in self.order if s.bn))
        self.partials = torch.empty(nb, **f32)
        self.sums = torch.empty(2 * 256, dtype=torch.float64, device=device)
        # packed weights (the first conv reads OIHW directly)
        self.packed = [s for s in self.order if s.idx > 0]
        self.wpk_f = {s.name: torch.empty(s.taps * s.cin_pad * s.cout_pad, **f32) for s in self.packed}
        self.wpk_d = {s.name: torch.empty(s.taps * s.cin_pad * s.cout_pad, **f32) for s in self.packed if s.idx > 1 or True}
        self.bias_pad = {s.name: torch.empty(s.cout_pad, **f32) for s in self.packed}
        # bf16x3 fragments (hi/lo split weights) for the split-precision matrix-core path
        self.kc = {s.name: self.lib.hpfg_conv_kc(s.h, s.w, s.taps) for s in self.packed}
        self.wpk16_f = {s.name: torch.empty(self.lib.hpfg_wpk16_elems(s.cin, s.cout_pad, s.taps, self.kc[s.name]), dtype=torch.bfloat16, device=device)
                        for s in self.packed}
        self.wpk16_d = {s.name: torch.empty(self.lib.hpfg_wpk16_elems(s.cout, s.cin_pad, s.taps, self.kc[s.name]), dtype=torch.bfloat16, device=device)
                        for s in self.packed}
        self._pack_tables: Dict[tuple, tuple] = {}   # (math, with_dgrad) -> (host descriptors, device copy), built on first use
        # BatchNorm sums through integer atomics (csrc/common.h: hpfg_acc_add): a conv adds its per-channel sums to the layer's accumulator and
        # every forward consumer derives scale / shift from it in its prologue, so NO finalize launch sits between two convs; ONE launch at the
        # end of the forward (hpfg_bn_acc_finalize) writes all 18 tables for the backward kernels and the running statistics.  bf16x3 kernels,
        # per-rank statistics only (the global-batch data-parallel mode exchanges the sums inside its finalize kernels and keeps them).
        self.bn_layers = [s for s in self.order if s.bn]      # (encoder layers first: build_specs order)
        self._n_enc_bn = sum(1 for s in self.bn_layers if s.name.startswith("encoder."))
        self.bn_acc_on = os.environ.get("HPFG_BN_ACC", "1") != "0"      # 0: the per-layer finalize launches (the path the global-batch data-parallel mode keeps)
        # shards: the producers' same-address atomics want many (ONE shard: +0.3 ms per step, two: +0.09), every consumer workgroup's prologue
        # wants few (32 bytes per channel and shard).  Measured in the step (profiles/r04_bn_acc.txt): 8 everywhere beats 4 and beats 8 / 4 / 2 by
        # channel count -- the contention costs more than the prologue reads.
        self.acc_shards = {s.name: L.ACC_MAX_SHARDS for s in self.bn_layers}
        self.acc_all = torch.zeros(sum(self.acc_shards[s.name] * 4 * s.cout for s in self.bn_layers), dtype=torch.int64, device=device)
        self.acc_of, off = {}, 0
        for s in self.bn_layers:
            assert s.cout == s.cout_pad
            self.acc_of[s.name] = self.acc_all[off:off + self.acc_shards[s.name] * 4 * s.cout]
            off += self.acc_shards[s.name] * 4 * s.cout
        self._acc_live, self._acc_dirty = False, False
        self._acc_tables: Dict[bool, tuple] = {}
        # set by the model for the network that runs on the step's ORIGIN stream (the single trainable network of a step): its weight packing
        # (22 us) runs on the side stream beside the first conv, which reads the OIHW weights directly and needs nothing the launch produces
        self.pack_overlap = False
        self.after_layer = None      # (layer index, callable): the step enqueues another network's forward behind that layer (graph submission order)
        # the same for the backward sums (sum g, sum g * xhat): added by the dgrad epilogue / the reduction pass that completes a layer's
        # gradient, read by every dZ consumer's prologue (k1 .. k3 derived there), turned into dgamma / dbeta -- and zeroed again -- by ONE
        # launch per backward (hpfg_bn_acc_bwd_finalize) instead of 18 finalize launches on the critical chain
        self.accb_all = torch.zeros_like(self.acc_all)
        self.accb_of, off = {}, 0
        for s in self.bn_layers:
            self.accb_of[s.name] = self.accb_all[off:off + self.acc_shards[s.name] * 4 * s.cout]
            off += self.acc_shards[s.name] * 4 * s.cout
        self._accb_live, self._accb_dirty = False, False
        self._accb_table = None
        self.seed_dev = torch.zeros(1, dtype=torch.int32, device=device)   # run-time dropout seed word
        self.bump_counters: Optional[torch.Tensor] = None      # set by the module in front of a train-mode forward: int64 counters the pack launch advances
        self.base_seed = 0x1234567
        self.train_stats = True
        self.bwd_ready = False
        self._bwd_alloc = False
        self.x: Optional[torch.Tensor] = None
        self.world = 1
        # BatchNorm-backward sums of the layer below from the dgrad epilogue (bf16x3 kernels) instead of a streaming pass of their own
        self.fuse_bwd_stats = 1      # 0: a streaming pass per layer (tests); 2: also the register-starved 32-channel instantiation
        self._fused_rows: Dict[str, int] = {}
        # thin 3x3 layers (16-pixel-aligned, <= 64 input / 32 output channels, split-bf16 math): ONE kernel produces the input gradient, the
        # weight-gradient slabs and the backward sums of the layer below from a single staging of dZ (hpfg_fused_bwd)
        self.fused_bwd = os.environ.get("HPFG_FUSED_BWD", "1") == "1"
        # HpfgConvArgs.stage_out: the separate dgrad of a 3x3 layer also stores the dZ it stages, and (act_side) its forward conv the virtual
        # input it stages; the layer's weight gradient reads those tensors as PLAIN sources instead of deriving both again -- BatchNorm,
        # LeakyReLU, Dropout, max-pool / bilinear taps forward, their backward for dZ -- in every (input slice x output slice) workgroup
        # the max-pool backward + BatchNorm-backward sums of a block output ride in the epilogue of the dgrad that produces dP (56 / 28 / 14-pixel levels)
        self.pool_fuse = True
        self._pool_done: set = set()
        self.dz_side = True      # (attributes, not environment switches: tests and A/B tools set them on the engine)
        self.act_side = True
        # the upsample backward of a decoder block is gathered by the 1x1 conv's dgrad while it stages its tiles (HPFG_ACT_UPBWD) instead of
        # by a launch of its own in front of it
        self.upb_fuse = os.environ.get("HPFG_UPB_FUSE", "1") == "1"
        self.upb_max_c1 = int(os.environ.get("HPFG_UPB_MAX_C1", "128"))
        self.dzbuf: Dict[str, torch.Tensor] = {}
        self.actbuf: Dict[str, torch.Tensor] = {}
        self._act_live: set = set()      # layers whose actbuf this forward wrote
        # set by the model (UNet.defer_wgrad, which the single-network step objects switch on).  Not with two trainable networks
        # back-propagating on two streams: the fork below would then leave a FORKED stream, and an event wait between two non-origin
        # streams of a capture makes hipStreamEndCapture fault on ROCm 7.2 (tools/nested_fork_probe.py reproduces it without this code)
        self.defer_wgrad = False
        self._deferred = None
        self.fused_grid: Dict[str, int] = {}
        self._last_fused: Dict[str, "L.FusedBwdArgs"] = {}
        self._side, self._side_used = None, False
        self.force_sync = False  # run the data-parallel code path (reduce -> all-reduce -> finalize) even with one rank (tests)
        self.math = L.MATH_F32      # L.MATH_BF16X3 selects the split-bf16 matrix-core kernels for conv forward / dgrad
        self.ext_masks: Dict[str, torch.Tensor] = {}   # conv name -> uint8 NHWC keep-mask (parity tests replaying torch's masks)
        self.allreduce = None    # callable(tensor) -> in-place sum across ranks (data parallel), set by hpfg_amd.parallel
        # peer mailbox exchange (hpfg_amd.parallel.DataParallelContext.enable_peer_exchange): the finalize kernels add the ranks' sums
        # themselves -- no collective between the kernels.  peer = the context, peer_base = first of this engine's 2 x 18 mailbox slots,
        # xepoch = device word counting this engine's train-mode forwards (the epoch of every slot use of that forward / its backward)
        self.peer, self.peer_base, self.xepoch = None, 0, None
        self.marks: Optional[MarkLog] = None      # bench.py: device time stamps around every conv / dgrad / wgrad / BatchNorm launch

    # ---------------------------------------------------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    def _run(self, tag: str, launch, stream: Optional[int] = None):
        """launch() -- bracketed by device time stamps on its stream when a MarkLog is attached."""
        if self.marks is None:
            return launch()
        return self.marks.bracket(tag, self._stream() if stream is None else stream, launch)

    def _bn_index(self, s: ConvSpec) -> int:
        if not hasattr(self, "_bn_idx"):
            self._bn_idx = {t.name: i for i, t in enumerate(t for t in self.order if t.bn)}
        return self._bn_idx[s.name]

    def layer_seed(self, s: ConvSpec) -> int:
        return (self.base_seed * 0x9E3779B1 + 0x85EBCA6B * (s.idx + 1)) & 0xFFFFFFFF

    def _act_bn(self, name: str, mode=L.ACT_BNACT) -> L.Act:
        s = self.specs[name]
        a = L.Act()
        a.z, a.bn, a.mode, a.C = L.ptr(self.z[name]), L.ptr(self.bn[name]), mode, s.cout
        a.Hs, a.Ws, a.pstride, a.bn_stride, a.bn_coff = s.h, s.w, s.cout, s.cout, 0
        if self._acc_live and mode in (L.ACT_BNACT, L.ACT_BNACT_POOL):      # the forward consumers form scale / shift from the sums themselves
            a.bn_acc, a.bn_gamma, a.bn_beta = L.ptr(self.acc_of[name]), L.ptr(self.params[f"{s.bn}.weight"]), L.ptr(self.params[f"{s.bn}.bias"])
            a.bn_count, a.bn_eps, a.bn_shards = float(self.N * s.h * s.w), BN_EPS, self.acc_shards[name]
        if mode == L.ACT_BNACT and s.drop_p > 0 and self.dropout_on:
            a.drop_p, a.drop_seed, a.seed_dev = s.drop_p, self.layer_seed(s), L.ptr(self.seed_dev)
            a.drop_mask = L.ptr(self.ext_masks.get(name)) if self.ext_masks else None
        return a

    def _act_plain(self, t: torch.Tensor, C_: int, h: int, w: int, pstride: Optional[int] = None, mode=L.ACT_PLAIN) -> L.Act:
        a = L.Act()
        a.z, a.mode, a.C, a.Hs, a.Ws = L.ptr(t), mode, C_, h, w
        a.pstride = C_ if pstride is None else pstride
        if C_ % 4 != 0 and mode == L.ACT_PLAIN:
            a.mode = L.ACT_STRIDED
            a.sn, a.sc, a.sy, a.sx = h * w * a.pstride, 1, w * a.pstride, a.pstride
        return a

    def _act_split(self, t: torch.Tensor, C_: int, h: int, w: int) -> L.Act:
        """A side tensor stored by HpfgConvArgs.stage_out: bf16 [N][h][w][C / 8][hi 8 | lo 8] in a buffer of N*h*w*C fp32 words."""
        a = L.Act()
        a.z, a.mode, a.C, a.Hs, a.Ws, a.pstride = L.ptr(t), L.ACT_SPLIT16, C_, h, w, C_
        return a

    def _act_input(self, x: torch.Tensor) -> L.Act:
        a = L.Act()
        a.z, a.mode, a.C, a.Hs, a.Ws = L.ptr(x), L.ACT_STRIDED, self.in_ch, self.H, self.W
        a.sn, a.sc, a.sy, a.sx = x.stride(0), x.stride(1), x.stride(2), x.stride(3)
        return a

    def _act_dz(self, name: str, dA: torch.Tensor, da_pstride: int) -> L.Act:
        s = self.specs[name]
        a = L.Act()
        a.z, a.bn, a.aux, a.mode, a.C = L.ptr(self.z[name]), L.ptr(self.bn[name]), L.ptr(dA), L.ACT_DZ, s.cout
        a.Hs, a.Ws, a.pstride, a.aux_pstride, a.bn_stride = s.h, s.w, s.cout, da_pstride, s.cout
        if self._accb_live:          # the dZ consumers form k1 .. k3 from the backward sums themselves
            a.bn_acc, a.bn_gamma, a.bn_count, a.bn_shards = L.ptr(self.accb_of[name]), L.ptr(self.params[f"{s.bn}.weight"]), float(self.N * s.h * s.w), self.acc_shards[name]
        if s.drop_p > 0 and self.dropout_on:
            a.drop_p, a.drop_seed, a.seed_dev = s.drop_p, self.layer_seed(s), L.ptr(self.seed_dev)
            a.drop_mask = L.ptr(self.ext_masks.get(name)) if self.ext_masks else None
        return a

    def input_acts(self, name: str):
        """(a0, a1) virtual input of conv `name` (forward view)."""
        s = self.specs[name]
        none = L.Act()
        if name == "encoder.in_conv.conv_conv.0":
            return self._act_input(self.x), none
        if name == "decoder.out_conv":
            return self._act_bn("decoder.up4.conv.conv_conv.4"), none
        if name.endswith(".4"):
            return self._act_bn(name[:-2] + ".0"), none
        if name.startswith("encoder.down"):
            lvl = int(name[len("encoder.down")])
            return self._act_bn(enc_prefix(lvl - 1) + ".4", L.ACT_BNACT_POOL), none
        k = int(name[len("decoder.up")])
        prev = enc_prefix(4) + ".4" if k == 1 else f"decoder.up{k - 1}.conv.conv_conv.4"
        if name.endswith("conv1x1"):
            return self._act_bn(prev), none
        # decoder block conv 0: cat([skip, up(conv1x1)])
        skip = self._act_bn(enc_prefix(4 - k) + ".4")
        u = self.specs[f"decoder.up{k}.conv1x1"]
        up = self._act_plain(self.z[u.name], u.cout, u.h, u.w, mode=L.ACT_UP2X)
        return skip, up

    # ---------------------------------------------------------------------------------------------------------
    def _pack_table(self, math: int, with_dgrad: bool):
        """Descriptor table for hpfg_pack_weights: only the fragment layouts this math mode / pass consumes are produced (the
        fp32 fragments in exact-fp32 mode, the split-bf16 ones in bf16x3 mode; the dgrad transposes only if a backward follows)."""
        key = (math, with_dgrad)
        if key not in self._pack_tables:
            b16 = math == L.MATH_BF16X3
            descs = (L.PackDesc * len(self.packed))()
            for d, s in zip(descs, self.packed):
                d.w_oihw, d.b = L.ptr(self.params[f"{s.name}.weight"]), L.ptr(self.params[f"{s.name}.bias"])
                d.bias_pad = L.ptr(self.bias_pad[s.name])
                d.wpk_fwd = None if b16 else L.ptr(self.wpk_f[s.name])
                d.wpk_dgrad = None if (b16 or not with_dgrad) else L.ptr(self.wpk_d[s.name])
                d.wpk16_fwd = L.ptr(self.wpk16_f[s.name]) if b16 else None
                d.wpk16_dgrad = L.ptr(self.wpk16_d[s.name]) if (b16 and with_dgrad) else None
                d.kc = self.kc[s.name]
                d.Cout, d.Cin, d.CoutPad, d.CinPad, d.taps = s.cout, s.cin, s.cout_pad, s.cin_pad, s.taps
            dev = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(self.dev)
            self._pack_tables[key] = (descs, dev)
        return self._pack_tables[key]

    def pack(self, with_dgrad: bool = True, counters: Optional[torch.Tensor] = None, seed_add: int = 0):
        """counters: int64 tensor whose elements the same launch advances by one (num_batches_tracked of the network's BatchNorm layers);
        seed_add: advance of the engine's dropout seed word (hpfg_pack_weights_bump)."""
        host, dev = self._pack_table(self.math, with_dgrad)
        if counters is not None:
            assert counters.dtype == torch.int64 and counters.is_contiguous() and counters.device == self.dev
        self._run("pack_weights", lambda: L.check(self.lib.hpfg_pack_weights_bump(
            dev.data_ptr(), host, len(self.packed), L.ptr(counters) if counters is not None else None,
            counters.numel() if counters is not None else 0, L.ptr(self.seed_dev), int(seed_add),
            None, 0, self._stream()), "pack_weights"))

    def _finalize_bwd_all(self, lo: int, hi: int):
        """hpfg_bn_acc_bwd_finalize for BatchNorm layers [lo, hi) of self.bn_layers (encoder layers come first): dgamma / dbeta (+ the k rows)
        from the backward accumulators, which the launch zeroes again."""
        if not self._accb_live or hi <= lo:
            return
        if self._accb_table is None:
            descs = (L.BnAccBwdDesc * len(self.bn_layers))()
            for d, s in zip(descs, self.bn_layers):
                d.acc, d.gamma, d.bn = L.ptr(self.accb_of[s.name]), L.ptr(self.params[f"{s.bn}.weight"]), L.ptr(self.bn[s.name])
                d.dgamma, d.dbeta = L.ptr(self.grads[f"{s.bn}.weight"]), L.ptr(self.grads[f"{s.bn}.bias"])
                d.C, d.count, d.shards = s.cout, float(self.N * s.h * s.w), self.acc_shards[s.name]
            self._accb_table = (descs, torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(self.dev))
        host, dev = self._accb_table
        sz = C.sizeof(L.BnAccBwdDesc)
        sub = (L.BnAccBwdDesc * (hi - lo)).from_buffer(host, lo * sz)
        self._run("bn_bfin_all", lambda: L.check(self.lib.hpfg_bn_acc_bwd_finalize(dev.data_ptr() + lo * sz, sub, hi - lo, self._stream()), "bn_acc_bwd_finalize"))

    def _finalize_all(self, track: bool):
        """hpfg_bn_acc_finalize: every BatchNorm table of this forward (+ running statistics) in one launch, from the sum accumulators."""
        if track not in self._acc_tables:
            descs = (L.BnAccDesc * len(self.bn_layers))()
            for d, s in zip(descs, self.bn_layers):
                d.acc, d.gamma, d.beta = L.ptr(self.acc_of[s.name]), L.ptr(self.params[f"{s.bn}.weight"]), L.ptr(self.params[f"{s.bn}.bias"])
                d.running_mean = L.ptr(self.buffers[f"{s.bn}.running_mean"]) if track else None
                d.running_var = L.ptr(self.buffers[f"{s.bn}.running_var"]) if track else None
                d.bn, d.C, d.count, d.shards = L.ptr(self.bn[s.name]), s.cout, float(self.N * s.h * s.w), self.acc_shards[s.name]
            self._acc_tables[track] = (descs, torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(self.dev))
        host, dev = self._acc_tables[track]
        self._run("bn_fin_all", lambda: L.check(self.lib.hpfg_bn_acc_finalize(dev.data_ptr(), host, len(self.bn_layers), BN_MOMENTUM, BN_EPS, self._stream()),
                                                "bn_acc_finalize"))

    def _finalize_bn(self, s: ConvSpec, nblk: int, track: bool):
        st = self._stream()
        count = float(self.N * s.h * s.w * self.world)
        g, b = self.params[f"{s.bn}.weight"], self.params[f"{s.bn}.bias"]
        rm = self.buffers[f"{s.bn}.running_mean"] if track else None
        rv = self.buffers[f"{s.bn}.running_var"] if track else None
        if self.peer is not None and (self.world > 1 or self.force_sync):
            px = self.peer.peer_desc(self.peer_base + 2 * self._bn_index(s), self.xepoch)
            self._run("bn_fin:" + s.name, lambda: L.check(self.lib.hpfg_bn_fwd_finalize_x(
                L.ptr(self.partials), nblk, C.byref(px), count, L.ptr(g), L.ptr(b), L.ptr(rm), L.ptr(rv), BN_MOMENTUM, BN_EPS, L.ptr(self.bn[s.name]),
                s.cout, st), "bn_fwd_finalize_x"))
        elif self.world > 1 or self.force_sync:
            sums = self.sums[: 2 * s.cout_pad]
            L.check(self.lib.hpfg_reduce_partials(L.ptr(self.partials), nblk, s.cout_pad, L.ptr(sums), st), "reduce_partials")
            self.allreduce(sums)
            L.check(self.lib.hpfg_bn_fwd_finalize(None, 0, L.ptr(sums), count, L.ptr(g), L.ptr(b), L.ptr(rm), L.ptr(rv), BN_MOMENTUM, BN_EPS,
                                                  L.ptr(self.bn[s.name]), s.cout, st), "bn_fwd_finalize")
        else:
            self._run("bn_fin:" + s.name, lambda: L.check(self.lib.hpfg_bn_fwd_finalize(
                L.ptr(self.partials), nblk, None, count, L.ptr(g), L.ptr(b), L.ptr(rm), L.ptr(rv), BN_MOMENTUM, BN_EPS, L.ptr(self.bn[s.name]), s.cout, st),
                "bn_fwd_finalize"))

    def _fwd_begin(self, x: torch.Tensor, train: bool, dropout: Optional[bool], seed_step: Optional[int], needs_grad: bool) -> torch.Tensor:
        assert x.is_cuda and x.dtype == torch.float32 and tuple(x.shape) == (self.N, self.in_ch, self.H, self.W), (x.shape, x.dtype, x.device)
        self.x = x
        self.train_mode = train
        self._act_live = set()
        self._stage_inputs = bool(train and needs_grad and self.act_side and self.math == L.MATH_BF16X3)
        self.dropout_on = train if dropout is None else dropout
        # seed_step: None = leave the seed word alone; SEED_BUMP = advance it on the device (a step being captured into a hipGraph: every
        # replay then draws new masks); otherwise the host's per-forward counter value
        if seed_step is not None and seed_step != SEED_BUMP:
            self.seed_dev.fill_(int(seed_step) & 0x7FFFFFFF)
        counters, self.bump_counters = self.bump_counters, None
        self._acc_live = bool(train and self.bn_acc_on and self.math == L.MATH_BF16X3 and self.peer is None and not (self.world > 1 or self.force_sync))
        if self._acc_live:
            if self._acc_dirty:          # a forward that never reached its finalize launch (an exception): start from zero (hpfg_bn_acc_finalize zeroes otherwise)
                self.acc_all.zero_()
            self._acc_dirty = True
        self._pack_args = dict(with_dgrad=bool(train and needs_grad), counters=counters, seed_add=1 if seed_step == SEED_BUMP else 0)
        if not (self.pack_overlap and train):
            self.pack(**self._pack_args)
            self._pack_args = None
        if self.marks is not None:
            self.marks.calib(self._stream())
        if train and self.peer is not None and (self.world > 1 or self.force_sync):
            self.peer.bump(self.xepoch, self._stream())      # the epoch of this forward's (and its backward's) mailbox exchanges
        return torch.empty(self.N, self.H, self.W, self.ncls, dtype=torch.float32, device=self.dev)

    def _conv_args(self, s: ConvSpec, out: torch.Tensor, want_stats: bool) -> L.ConvArgs:
        ca = L.ConvArgs()
        ca.a0, ca.a1 = self.input_acts(s.name)
        ca.math = self.math
        ca.wpk = L.ptr(self.wpk16_f[s.name]) if self.math == L.MATH_BF16X3 else L.ptr(self.wpk_f[s.name])
        ca.bias, ca.out = L.ptr(self.bias_pad[s.name]), L.ptr(out)
        ca.stat_partials = L.ptr(self.partials) if want_stats else None
        ca.out_pstride, ca.Cout, ca.CoutPad = s.cout, s.cout, s.cout_pad
        ca.N, ca.H, ca.W, ca.taps = self.N, s.h, s.w, s.taps
        return ca

    def forward(self, x: torch.Tensor, train: bool = True, dropout: Optional[bool] = None, track_running: bool = True,
                seed_step: Optional[int] = None, needs_grad: bool = True) -> torch.Tensor:
        """x: [N,C,H,W] fp32 on the device (any strides).  Returns logits as an [N,H,W,ncls] tensor (fresh allocation)."""
        logits = self._fwd_begin(x, train, dropout, seed_step, needs_grad)
        for i, s in enumerate(self.order):
            if i == 0 and self._pack_args is not None:
                # one fork / join against the stream this forward runs on (the step's origin stream: UNet sets pack_overlap only there -- a fork
                # of a forked stream inside a capture faults in hipStreamEndCapture on ROCm 7.2): [pack] beside [first conv]
                main = torch.cuda.current_stream(self.dev)
                if self._side is None:
                    self._side = torch.cuda.Stream(device=self.dev)
                self._side.wait_stream(main)
                with torch.cuda.stream(self._side):
                    self.pack(**self._pack_args)
                self._pack_args = None
                self._fwd_layer(s, logits, train, track_running)
                main.wait_stream(self._side)
                continue
            self._fwd_layer(s, logits, train, track_running)
            if self.after_layer is not None and i == self.after_layer[0]:
                self.after_layer[1]()
        if self._acc_live:
            self._finalize_all(track_running)
            self._acc_dirty = False
        self.bwd_ready = bool(train and needs_grad)
        return logits

    def _fwd_layer(self, s: ConvSpec, logits: torch.Tensor, train: bool, track_running: bool):
        """One conv of the schedule (+ its BatchNorm table) on the current stream."""
        st = self._stream()
        out = logits if s.name == "decoder.out_conv" else self.z[s.name]
        want_stats = bool(s.bn) and train
        acc = self._acc_live and want_stats      # the sums go into the layer accumulator; the consumers (and _finalize_all) take them from there
        nblk = self.lib.hpfg_conv_stat_blocks(self.N, s.h, s.w)
        if s.idx == 0:
            a0, _ = self.input_acts(s.name)
            if acc:
                self._run("fwd:" + s.name, lambda: L.check(self.lib.hpfg_conv3x3_first_fwd_acc(
                    C.byref(a0), L.ptr(self.params[f"{s.name}.weight"]), L.ptr(self.params[f"{s.name}.bias"]), L.ptr(out),
                    None, L.ptr(self.acc_of[s.name]), self.acc_shards[s.name], self.N, s.h, s.w, s.cin,
                    s.cout, st), "conv3x3_first_fwd_acc"))
            else:
                self._run("fwd:" + s.name, lambda: L.check(self.lib.hpfg_conv3x3_first_fwd(
                    C.byref(a0), L.ptr(self.params[f"{s.name}.weight"]), L.ptr(self.params[f"{s.name}.bias"]), L.ptr(out),
                    L.ptr(self.partials) if want_stats else None, self.N, s.h, s.w, s.cin, s.cout, st), "conv3x3_first_fwd"))
            nblk = self.lib.hpfg_conv_first_rows(self.N, s.h, s.w)
        else:
            ca = self._conv_args(s, out, want_stats and not acc)
            if acc:
                ca.stat_acc, ca.stat_shards = L.ptr(self.acc_of[s.name]), self.acc_shards[s.name]
            if self._stage_inputs and self._side_layer(s) and s.cin % 8 == 0 and ca.a0.mode != L.ACT_PLAIN:
                buf = self.actbuf.get(s.name)
                if buf is None:
                    buf = self.actbuf[s.name] = torch.empty(self.N, s.h, s.w, s.cin, dtype=torch.float32, device=self.dev)
                ca.stage_out = L.ptr(buf)
                self._act_live.add(s.name)
            self._run("fwd:" + s.name, lambda: L.check(self.lib.hpfg_conv_fwd(C.byref(ca), st), f"conv_fwd[{s.name}]"))
            if want_stats and not acc:
                nblk = self.lib.hpfg_conv_stat_rows(C.byref(ca))
        if s.bn:
            if train and acc:
                pass
            elif train:
                self._finalize_bn(s, nblk, track_running)
            else:
                L.check(self.lib.hpfg_bn_eval_table(L.ptr(self.params[f"{s.bn}.weight"]), L.ptr(self.params[f"{s.bn}.bias"]),
                                                    L.ptr(self.buffers[f"{s.bn}.running_mean"]), L.ptr(self.buffers[f"{s.bn}.running_var"]), BN_EPS,
                                                    L.ptr(self.bn[s.name]), s.cout, st), "bn_eval_table")

    def materialize(self, name: str, mode=L.ACT_BNACT) -> torch.Tensor:
        """Activated output of conv `name` as a real [N,h,w,C] tensor (projection-neck input, tests)."""
        s = self.specs[name]
        a = self._act_bn(name, mode)
        h, w = (s.h // 2, s.w // 2) if mode == L.ACT_BNACT_POOL else (s.h, s.w)
        out = torch.empty(self.N, h, w, s.cout, dtype=torch.float32, device=self.dev)
        L.check(self.lib.hpfg_act_materialize(C.byref(a), None, self.N, h, w, L.ptr(out), self._stream()), "act_materialize")
        return out

    def materialize_input(self, name: str) -> torch.Tensor:
        """Virtual input of conv `name` ([N,h,w,Cin]) -- what the conv kernel stages into LDS."""
        s = self.specs[name]
        a0, a1 = self.input_acts(name)
        out = torch.empty(self.N, s.h, s.w, s.cin, dtype=torch.float32, device=self.dev)
        L.check(self.lib.hpfg_act_materialize(C.byref(a0), C.byref(a1), self.N, s.h, s.w, L.ptr(out), self._stream()), "act_materialize")
        return out

    # ---------------------------------------------------------------------------------------------------------
    def _alloc_bwd(self):
        key = (self.math, self.fused_bwd, self.upb_fuse)      # what the slab layout depends on
        if self._bwd_alloc and self._bwd_alloc_key == key:
            return
        self._bwd_alloc_key = key
        f32 = dict(dtype=torch.float32, device=self.dev)
        N = self.N
        self.dA: Dict[str, torch.Tensor] = {}
        self.dA_ps: Dict[str, int] = {}
        # gradient w.r.t. each BN'd conv's activated output.  Skip features alias the first half of the decoder's dCat.
        self.dcat: Dict[int, torch.Tensor] = {}
        self.dup: Dict[int, torch.Tensor] = {}
        for k in range(1, 5):
            s = self.specs[f"decoder.up{k}.conv.conv_conv.0"]
            skip = enc_prefix(4 - k) + ".4"
            if s.cin // 2 * 4 < 128:              # a half of the concat gradient is less than a 128-byte line per pixel (up4: 16 channels): interleaved,
                self.dcat[k] = torch.empty(N, s.h, s.w, s.cin // 2, **f32)     # every reader of one half would fetch both -> two buffers (HpfgConvArgs.out2)
                self.dup[k] = torch.empty(N, s.h, s.w, s.cin // 2, **f32)
                self.dA[skip], self.dA_ps[skip] = self.dcat[k], s.cin // 2
                continue
            self.dcat[k] = torch.empty(N, s.h, s.w, s.cin, **f32)
            self.dA[skip] = self.dcat[k]          # channels [0, C2) with pixel stride 2*C2
            self.dA_ps[skip] = s.cin
        for s in self.order:
            if s.bn and s.name not in self.dA:
                self.dA[s.name] = torch.empty(N, s.h, s.w, s.cout, **f32)
                self.dA_ps[s.name] = s.cout
        self.dU = {k: torch.empty(N, self.specs[f"decoder.up{k}.conv1x1"].h, self.specs[f"decoder.up{k}.conv1x1"].w,
                                  self.specs[f"decoder.up{k}.conv1x1"].cout, **f32) for k in range(1, 5)}
        self.dP = {lvl: torch.empty(N, self.specs[enc_prefix(lvl) + ".0"].h, self.specs[enc_prefix(lvl) + ".0"].w,
                                    self.specs[enc_prefix(lvl) + ".0"].cin, **f32) for lvl in range(1, 5)}
        # one slab region per layer; all of them are summed by ONE launch at the end of backward()
        sizes = [self.lib.hpfg_wgrad_slab_floats(N, s.h, s.w, s.cin_pad, s.cout_pad, s.taps) for s in self.order]
        self.fused_grid = {}
        if self.fused_bwd and self.math == L.MATH_BF16X3:
            for i, s in enumerate(self.order):
                if s.taps != 9:
                    continue
                fa = L.FusedBwdArgs()
                fa.xa0, fa.xa1 = self.input_acts(s.name)
                fa.d.a0.mode = L.ACT_DZ if s.bn else L.ACT_PLAIN
                fa.d.out = None if s.idx == 0 else L.ptr(self.partials)      # (no dX for the first layer: weight gradient only; any non-null otherwise)
                fa.d.N, fa.d.H, fa.d.W, fa.d.taps, fa.d.math = N, s.h, s.w, 9, self.math
                fa.Cin, fa.CinPad, fa.Cout, fa.CoutPad = s.cin, s.cin_pad, s.cout, s.cout_pad
                grid = self.lib.hpfg_fused_bwd_grid(C.byref(fa))
                if grid > 0:
                    self.fused_grid[s.name] = grid
                    sizes[i] = max(sizes[i], grid * 9 * s.cin_pad * s.cout_pad)
        self.slab_all = torch.empty(sum(sizes), **f32)
        self.slab_of, off = {}, 0
        # bias gradients of the convs without BatchNorm (1x1 convs, out_conv): their per-block channel sums are summed by the same
        # launch, as pseudo layers {taps 1, Cin 1} (one kernel less per bias)
        self.bias_layers = [s for s in self.order if not s.bn]
        self.csum_rows = {s.name: ((self._upb_rows(s) if self._upb_on(s) else self.lib.hpfg_upsample2x_bwd_blocks(N, s.h, s.w, s.cout)) if s.taps == 1
                                   else self.lib.hpfg_channel_sum_blocks(N * s.h * s.w, s.cout)) for s in self.bias_layers}
        self.csum_part = {s.name: torch.empty(self.csum_rows[s.name] * s.cout, **f32) for s in self.bias_layers}
        descs = (L.SlabDesc * (len(self.order) + len(self.bias_layers)))()
        for d, s, sz in zip(descs, self.order, sizes):
            self.slab_of[s.name] = self.slab_all[off:off + sz]
            off += sz
            d.slab, d.dw_oihw = L.ptr(self.slab_of[s.name]), L.ptr(self.grads[f"{s.name}.weight"])
            d.S = self.fused_grid.get(s.name) or self.lib.hpfg_wgrad_splits(N, s.h, s.w, s.cin_pad, s.cout_pad, s.taps)
            d.taps, d.Cin, d.CinPad, d.Cout, d.CoutPad = s.taps, s.cin, s.cin_pad, s.cout, s.cout_pad
        for j, s in enumerate(self.bias_layers):
            d = descs[len(self.order) + j]
            d.slab, d.dw_oihw = L.ptr(self.csum_part[s.name]), L.ptr(self.grads[f"{s.name}.bias"])
            d.S = self.csum_rows[s.name]
            d.taps, d.Cin, d.CinPad, d.Cout, d.CoutPad = 1, 1, 1, s.cout, s.cout
        self._slab_host = descs
        self._slab_dev = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(self.dev)
        # descriptor order = encoder convs | decoder convs | bias pseudo layers (all decoder): two contiguous ranges, so the decoder's
        # weight gradients can be finished (and handed to the data-parallel all-reduce) while the encoder half still back-propagates
        self._n_enc_desc = sum(1 for s in self.order if s.name.startswith("encoder."))
        # encoder levels 0 and 1 (in_conv, down1) come first: their slabs are written last (by the main stream's final kernels)
        self._n_thin_enc_desc = sum(1 for s in self.order if s.name.startswith("encoder.in_conv") or s.name.startswith("encoder.down1"))
        self._bwd_alloc = True

    def _bn_backward(self, s: ConvSpec, pooled_grad: Optional[torch.Tensor] = None):
        """BatchNorm/LeakyReLU/Dropout backward statistics of layer s -> k1,k2,k3 rows + dgamma/dbeta.
        pooled_grad: gradient w.r.t. MaxPool2d(2)(this layer's activation) [N,h/2,w/2,C]; it is scattered into dA by the same pass."""
        st = self._stream()
        g = self._act_dz(s.name, self.dA[s.name], self.dA_ps[s.name])
        fused = self._fused_rows.pop(s.name, None)
        if self._accb_live:          # sums -> the layer's backward accumulator; no finalize launch: the consumers of `g` derive k1 .. k3
            acc, sh = L.ptr(self.accb_of[s.name]), self.acc_shards[s.name]
            if fused is not None:
                assert pooled_grad is None      # (the dgrad epilogue that produced dA added them)
            elif pooled_grad is not None:
                self._run("bn_red:" + s.name, lambda: L.check(self.lib.hpfg_bn_bwd_reduce_pool_acc(
                    C.byref(g), L.ptr(pooled_grad), s.cout, self.N, s.h // 2, s.w // 2, acc, sh, st), f"bn_bwd_reduce_pool_acc[{s.name}]"))
            else:
                self._run("bn_red:" + s.name, lambda: L.check(self.lib.hpfg_bn_bwd_reduce_acc(C.byref(g), self.N, s.h, s.w, acc, sh, st),
                                                               f"bn_bwd_reduce_acc[{s.name}]"))
            return g
        if fused is not None:          # the dgrad that produced dA already left the sums in self.partials
            assert pooled_grad is None
            nblk = fused
        elif pooled_grad is not None:
            nblk = self.lib.hpfg_bn_bwd_pool_blocks(self.N, s.h // 2, s.w // 2, s.cout)
            self._run("bn_red:" + s.name, lambda: L.check(self.lib.hpfg_bn_bwd_reduce_pool(
                C.byref(g), L.ptr(pooled_grad), s.cout, self.N, s.h // 2, s.w // 2, L.ptr(self.partials), st), f"bn_bwd_reduce_pool[{s.name}]"))
        else:
            nblk = self.lib.hpfg_bn_bwd_blocks(self.N, s.h, s.w, s.cout)
            self._run("bn_red:" + s.name, lambda: L.check(self.lib.hpfg_bn_bwd_reduce(C.byref(g), self.N, s.h, s.w, L.ptr(self.partials), st),
                                                           f"bn_bwd_reduce[{s.name}]"))
        count = float(self.N * s.h * s.w * self.world)
        gam = self.params[f"{s.bn}.weight"]
        dg, db = self.grads[f"{s.bn}.weight"], self.grads[f"{s.bn}.bias"]
        if self.peer is not None and (self.world > 1 or self.force_sync):
            px = self.peer.peer_desc(self.peer_base + 2 * self._bn_index(s) + 1, self.xepoch)
            self._run("bn_bfin:" + s.name, lambda: L.check(self.lib.hpfg_bn_bwd_finalize_x(
                L.ptr(self.partials), nblk, C.byref(px), count, L.ptr(gam), L.ptr(self.bn[s.name]), L.ptr(dg), L.ptr(db), s.cout, 1.0 / self.world, st),
                "bn_bwd_finalize_x"))
        elif self.world > 1 or self.force_sync:
            sums = self.sums[: 2 * s.cout]
            L.check(self.lib.hpfg_reduce_partials(L.ptr(self.partials), nblk, s.cout, L.ptr(sums), st), "reduce_partials")
            self.allreduce(sums)
            L.check(self.lib.hpfg_bn_bwd_finalize(None, 0, L.ptr(sums), count, L.ptr(gam), L.ptr(self.bn[s.name]), L.ptr(dg), L.ptr(db), s.cout,
                                                  1.0 / self.world, st), "bn_bwd_finalize")      # global sums on every rank: the SUM all-reduce of the gradients restores them
        else:
            self._run("bn_bfin:" + s.name, lambda: L.check(self.lib.hpfg_bn_bwd_finalize(
                L.ptr(self.partials), nblk, None, count, L.ptr(gam), L.ptr(self.bn[s.name]), L.ptr(dg), L.ptr(db), s.cout, 1.0, st), "bn_bwd_finalize"))
        return g

    def _wgrad_dgrad(self, s: ConvSpec, g: L.Act, dgrad_out: torch.Tensor, stats_for: Optional[str] = None, out2: Optional[torch.Tensor] = None,
                     pool_of: Optional[str] = None):
        """Both gradients of layer s from the same dZ source (they only read it, so their order is free)."""
        if s.name in self.fused_grid:
            return self._fused_bwd(s, g, dgrad_out, stats_for, out2)      # (pool_of: the fused kernel's variant measured mixed, profiles/r04_schedule_experiments.txt)
        dz = None
        if self.dz_side and self.math == L.MATH_BF16X3 and self._side_layer(s) and g.mode == L.ACT_DZ and s.cout % 8 == 0:
            dz = self.dzbuf.get(s.name)
            if dz is None:
                dz = self.dzbuf[s.name] = torch.empty(self.N, s.h, s.w, s.cout, dtype=torch.float32, device=self.dev)
        if self._deferred is not None:      # decoder half: the weight gradient is queued for the side stream (see backward())
            self._dgrad(s, g, dgrad_out, stats_for, out2, stage_out=dz, pool_of=pool_of)
            self._deferred.append((s, g if dz is None else self._act_split(dz, s.cout, s.h, s.w)))
        else:
            if dz is not None:
                self._dgrad(s, g, dgrad_out, stats_for, out2, stage_out=dz, pool_of=pool_of)
                self._wgrad(s, self._act_split(dz, s.cout, s.h, s.w))
            else:
                self._wgrad(s, g)
                self._dgrad(s, g, dgrad_out, stats_for, out2, pool_of=pool_of)

    def _upb_on(self, su: ConvSpec) -> bool:
        """Does the dgrad of this decoder 1x1 conv gather the upsample backward itself?  (bf16x3 kernels, channel groups of 8)"""
        # (not the 14-pixel level, C1 = 256: the gather is repeated in each of its four output-channel slices, in a launch of few workgroups --
        # up1 12.0 -> 23.3 us at 8 images; up4 31.7 -> 27.0, up3 19.9 -> 19.3 us; up2, two slices: +4.8 us at 8 images, -3.5 at 16.
        # Same-box A/B of the threshold: profiles/r05_upb_fuse.txt)
        return bool(self.upb_fuse and self.math == L.MATH_BF16X3 and su.taps == 1 and su.cout % 8 == 0 and su.cin <= self.upb_max_c1)

    def _upb_rows(self, su: ConvSpec) -> int:
        """Workgroups of that dgrad launch = rows of the bias-gradient sums it leaves (16 x 16-pixel tiles at the aligned sizes, 8 x 8 otherwise)."""
        t = 16 if (su.h % 16 == 0 and su.w % 16 == 0) else 8
        return self.N * ((su.h + t - 1) // t) * ((su.w + t - 1) // t)

    @staticmethod
    def _side_layer(s: ConvSpec) -> bool:
        """3x3 layers below the 16-pixel-aligned resolutions: separate dgrad + wgrad on the persistent conv kernel (the aligned ones run the
        fused / thin kernels, which stage input and dZ once for both gradients already)."""
        return s.taps == 9 and bool(s.h % 16 or s.w % 16)

    def _fused_bwd(self, s: ConvSpec, g: L.Act, out: torch.Tensor, stats_for: Optional[str], out2: Optional[torch.Tensor]):
        """hpfg_fused_bwd: dX into `out` (/ `out2`), the weight-gradient slabs of layer s and -- with stats_for -- the BatchNorm-backward
        sums of the layer below, from one read of (dA, z) and one read of the layer input."""
        fa = L.FusedBwdArgs()
        fa.xa0, fa.xa1 = self.input_acts(s.name)
        fa.slab = L.ptr(self.slab_of[s.name])
        fa.Cin, fa.CinPad, fa.Cout, fa.CoutPad = s.cin, s.cin_pad, s.cout, s.cout_pad
        ca = fa.d
        ca.a0, ca.math, ca.out = g, self.math, L.ptr(out)
        ca.wpk = L.ptr(self.wpk16_d[s.name]) if out is not None else None      # (out None: the first layer, weight gradient only)
        ca.out_pstride, ca.Cout, ca.CoutPad = s.cin, s.cin, s.cin_pad
        if out2 is not None:
            ca.out2, ca.out_split, ca.out_pstride, ca.out2_pstride = L.ptr(out2), s.cin // 2, s.cin // 2, s.cin // 2
        ca.N, ca.H, ca.W, ca.taps = self.N, s.h, s.w, 9
        rows = self.fused_grid[s.name]
        if stats_for is not None and self.fuse_bwd_stats and s.cin == s.cin_pad and out2 is None:
            if rows * 2 * s.cin > self.partials.numel():
                raise RuntimeError(f"fused_bwd[{s.name}]: {rows} rows of backward sums do not fit the partials workspace")
            ca.bwd_stats, ca.bwd_of, ca.stat_partials = 1, self._act_dz(stats_for, out, s.cin), L.ptr(self.partials)
            if self._accb_live:
                ca.stat_partials, ca.stat_acc, ca.stat_shards = None, L.ptr(self.accb_of[stats_for]), self.acc_shards[stats_for]
            self._fused_rows[stats_for] = rows
        self._last_fused[s.name] = fa      # (bench.py re-launches it alone)
        self._run("fused_bwd:" + s.name, lambda: L.check(self.lib.hpfg_fused_bwd(C.byref(fa), self._stream()), f"fused_bwd[{s.name}]"))

    def _wgrad(self, s: ConvSpec, g: L.Act, on_side: bool = False):
        """Weight gradient of layer s.  It is off the critical chain of backward (nothing downstream consumes it before the final
        slab reduction), so it is issued on a second HIP stream forked behind the kernels recorded so far and joined at the end."""
        # on_side: one of the deferred batches, on the side stream the caller (backward / flush) forked
        stream = self._side.cuda_stream if on_side else torch.cuda.current_stream(self.dev).cuda_stream
        wa = L.WgradArgs()
        if s.name in self._act_live:      # the forward conv stored the input it staged
            wa.a0, wa.a1 = self._act_split(self.actbuf[s.name], s.cin, s.h, s.w), L.Act()
        else:
            wa.a0, wa.a1 = self.input_acts(s.name)
        wa.g = g
        wa.slab, wa.dw_oihw, wa.defer_reduce = L.ptr(self.slab_of[s.name]), L.ptr(self.grads[f"{s.name}.weight"]), 1
        wa.Cin, wa.CinPad, wa.Cout, wa.CoutPad = s.cin, s.cin_pad, s.cout, s.cout_pad
        wa.N, wa.H, wa.W, wa.taps = self.N, s.h, s.w, s.taps
        wa.S = self.lib.hpfg_wgrad_splits(self.N, s.h, s.w, s.cin_pad, s.cout_pad, s.taps)
        wa.math = self.math
        self._run("wgrad:" + s.name, lambda: L.check(self.lib.hpfg_wgrad(C.byref(wa), stream), f"wgrad[{s.name}]"), stream)

    def _dgrad(self, s: ConvSpec, g: L.Act, out: torch.Tensor, stats_for: Optional[str] = None, out2: Optional[torch.Tensor] = None,
               stage_out: Optional[torch.Tensor] = None, pool_of: Optional[str] = None, side_sums: Optional[torch.Tensor] = None):
        """out [N,h,w,cin] = conv-transpose of dZ with this layer's weights.
        stats_for: name of the BatchNorm layer whose activated output `out` is the COMPLETE gradient of (this conv is its only
        consumer): the bf16x3 kernel's epilogue then also leaves that layer's backward sums in self.partials, and the following
        _bn_backward(stats_for) skips its own streaming pass over (dA, z)."""
        ca = L.ConvArgs()
        ca.a0, ca.a1 = g, L.Act()
        ca.math = self.math
        ca.wpk = L.ptr(self.wpk16_d[s.name]) if self.math == L.MATH_BF16X3 else L.ptr(self.wpk_d[s.name])
        ca.bias, ca.out, ca.stat_partials = None, L.ptr(out), None
        ca.out_pstride, ca.Cout, ca.CoutPad = s.cin, s.cin, s.cin_pad
        if out2 is not None:          # [d(skip) | d(upsampled)] into two buffers
            ca.out2, ca.out_split, ca.out_pstride, ca.out2_pstride = L.ptr(out2), s.cin // 2, s.cin // 2, s.cin // 2
        ca.N, ca.H, ca.W, ca.taps = self.N, s.h, s.w, s.taps
        if stage_out is not None:
            ca.stage_out = L.ptr(stage_out)
        if side_sums is not None:
            ca.side_sums = L.ptr(side_sums)
            rows = self.lib.hpfg_conv_stat_rows(C.byref(ca))
            if rows != self.csum_rows[s.name]:
                raise RuntimeError(f"dgrad[{s.name}]: {rows} workgroups, {self.csum_rows[s.name]} rows of bias-gradient sums allocated")
        # (not for the 32-channel slices of 16x16-pixel tiles: that instantiation is out of registers and the extra epilogue spills)
        spills = s.taps == 9 and s.cin_pad % 32 == 0 and s.h % 16 == 0 and s.w % 16 == 0
        if stats_for is not None and self.math == L.MATH_BF16X3 and self.fuse_bwd_stats and s.cin == s.cin_pad and (not spills or self.fuse_bwd_stats == 2):
            ca.bwd_stats, ca.bwd_of, ca.stat_partials = 1, self._act_dz(stats_for, out, s.cin), L.ptr(self.partials)
            rows = self.lib.hpfg_conv_stat_rows(C.byref(ca))
            if rows <= 0 or rows * 2 * s.cin > self.partials.numel():
                raise RuntimeError(f"dgrad[{s.name}]: {rows} rows of backward sums do not fit the partials workspace")
            if self._accb_live:
                ca.stat_partials, ca.stat_acc, ca.stat_shards = None, L.ptr(self.accb_of[stats_for]), self.acc_shards[stats_for]
            self._fused_rows[stats_for] = rows
        elif pool_of is not None and self.math == L.MATH_BF16X3 and self.fuse_bwd_stats and s.cin == s.cin_pad and self._side_layer(s) and out2 is None:
            # `out` would be dP, the gradient w.r.t. MaxPool2d(2) of layer pool_of's activation: the epilogue scatters it into that layer's
            # gradient (arg-max of the 2 x 2 window) and takes the BatchNorm-backward sums of the completed gradient -- hpfg_bn_bwd_reduce_pool's
            # pass, without its launch on the chain
            ca.bwd_stats, ca.bwd_of, ca.stat_partials = 2, self._act_dz(pool_of, self.dA[pool_of], self.dA_ps[pool_of]), L.ptr(self.partials)
            rows = self.lib.hpfg_conv_stat_rows(C.byref(ca))
            if rows <= 0 or rows * 2 * s.cin > self.partials.numel():
                raise RuntimeError(f"dgrad[{s.name}]: {rows} rows of backward sums do not fit the partials workspace")
            if self._accb_live:
                ca.stat_partials, ca.stat_acc, ca.stat_shards = None, L.ptr(self.accb_of[pool_of]), self.acc_shards[pool_of]
            self._fused_rows[pool_of] = rows
            self._pool_done.add(pool_of)
        self._run("dgrad:" + s.name, lambda: L.check(self.lib.hpfg_conv_fwd(C.byref(ca), self._stream()), f"dgrad[{s.name}]"))

    def _slab_reduce(self, lo: int, hi: int, stream=None):
        """Sum the weight-gradient slabs of descriptors [lo, hi) into the gradient buffer (one launch)."""
        if hi <= lo:
            return
        sz = C.sizeof(L.SlabDesc)
        host = (L.SlabDesc * (hi - lo)).from_buffer(self._slab_host, lo * sz)
        st_ = stream if stream is not None else self._stream()
        self._run("slab_reduce", lambda: L.check(self.lib.hpfg_slab_reduce_multi(self._slab_dev.data_ptr() + lo * sz, host, hi - lo, st_), "slab_reduce_multi"), st_)

    def backward(self, dlogits: torch.Tensor, dfeat4: Optional[torch.Tensor] = None, bucket_cb=None):
        """dlogits: [N,H,W,ncls] contiguous.  dfeat4: optional gradient w.r.t. the activated bottleneck [N,h,w,256].
        bucket_cb: optional callable(i); called with 0 once every DECODER parameter gradient is final (queued on the current stream:
        decoder slabs reduced by a launch of their own) and with 1 after the encoder's -- data parallel: the decoder bucket is
        all-reduced on a side stream while the encoder half of backward still computes (hpfg_amd.parallel.GradBuckets).
        Writes every parameter gradient of encoder/decoder into self.grads (overwrite).  Biases of convs that feed a
        train-mode BatchNorm have an exactly zero gradient (the batch mean removes them); their slots are never written and
        rely on the zero-initialised gradient buffer."""
        assert self.bwd_ready, "backward() needs a preceding train-mode forward()"
        assert dlogits.is_contiguous() and tuple(dlogits.shape) == (self.N, self.H, self.W, self.ncls)
        self._alloc_bwd()
        st = self._stream()
        N = self.N
        sp = self.specs
        self._accb_live = bool(self.bn_acc_on and self.math == L.MATH_BF16X3 and self.peer is None and not (self.world > 1 or self.force_sync))
        if self._accb_live:
            if self._accb_dirty:          # a backward pass that did not reach its finalize launch (an exception): start from zero
                self.accb_all.zero_()
            self._accb_dirty = True
        # ONE fork for the decoder's separate weight gradients (the channel-rich layers and the 1x1 convs; the thin layers' are fused with their
        # dgrad): they are queued while the decoder half back-propagates and run on the side stream beside the encoder half -- everything they
        # read (dA, z, the BatchNorm tables of their layers) stays in place until the next forward.  A fork / join per layer cost more than it
        # returned (round 1; removed); this is one of each.  Not with the data-parallel buckets: the decoder's gradients must be final at the
        # bucket boundary.
        self._deferred = [] if (self.defer_wgrad and bucket_cb is None) else None
        self._pool_done = set()
        # ---- out_conv
        s = sp["decoder.out_conv"]
        g = self._act_plain(dlogits, self.ncls, s.h, s.w)
        if self.marks is not None:
            self.marks.calib(st)          # (what a bracket costs by itself, measured where backward starts)

        def csum(stream):          # out_conv's bias gradient: per-block channel sums of dlogits (rows of the final slab reduction)
            self._run("csum:" + s.name, lambda: L.check(self.lib.hpfg_channel_sum_partials(
                L.ptr(dlogits), self.ncls, N * s.h * s.w, self.ncls, L.ptr(self.csum_part[s.name]), stream), "channel_sum_partials"), stream)

        if self._deferred is None:
            csum(st)
        # (else: nothing on the chain of backward reads those rows -- they go out with the decoder's queued weight gradients on the side stream,
        # instead of 11 us at the head of the critical chain)
        self._wgrad_dgrad(s, g, self.dA["decoder.up4.conv.conv_conv.4"], "decoder.up4.conv.conv_conv.4")
        self._csum_done = False

        def flush(lo, hi):
            """queued weight gradients -> side stream, followed there by the slab reduction of descriptors [lo, hi) (all their producers -- these
            launches and fused kernels already queued on the main stream -- are ordered before it by the fork)"""
            if self._deferred is not None:
                main = torch.cuda.current_stream(self.dev)
                if self._side is None:
                    self._side = torch.cuda.Stream(device=self.dev)
                self._side.wait_stream(main)
                if not self._csum_done:          # out_conv's bias sums ride along with the first batch
                    csum(self._side.cuda_stream)
                    self._csum_done = True
                split = hi == self._n_enc_desc and lo + 2 < hi and len(self._deferred) == hi - lo
                if split:      # the deeper levels first, their slabs reduced beside the last level's weight gradients: the serial piece at the stream's
                    # end -- it finished AFTER the main chain -- shrinks from 58 to ~20 us (with the reduction below: mt -0.8 %)
                    for s_, g_ in self._deferred[:-2]:
                        self._wgrad(s_, g_, on_side=True)
                    self._slab_reduce(lo + 2, hi, self._side.cuda_stream)
                    for s_, g_ in self._deferred[-2:]:
                        self._wgrad(s_, g_, on_side=True)
                    self._slab_reduce(lo, lo + 2, self._side.cuda_stream)
                else:
                    for s_, g_ in self._deferred:
                        self._wgrad(s_, g_, on_side=True)
                    self._slab_reduce(lo, hi, self._side.cuda_stream)
                self._side_used = True

        # ---- decoder blocks, last to first
        for k in range(4, 0, -1):
            p = f"decoder.up{k}.conv.conv_conv"
            s2, s1, su = sp[f"{p}.4"], sp[f"{p}.0"], sp[f"decoder.up{k}.conv1x1"]
            g2 = self._bn_backward(s2)
            self._wgrad_dgrad(s2, g2, self.dA[s1.name], s1.name)
            g1 = self._bn_backward(s1)
            c2 = su.cout
            if k in self.dup:
                self._wgrad_dgrad(s1, g1, self.dcat[k], out2=self.dup[k])      # dSkip and dUp in buffers of their own
                dup, dup_ps = self.dup[k], c2
            else:
                self._wgrad_dgrad(s1, g1, self.dcat[k])            # [dSkip | dUp]
                dup, dup_ps = self.dcat[k].view(-1)[c2:], 2 * c2   # channel offset c2, pixel stride 2*c2
            prev = enc_prefix(4) + ".4" if k == 1 else f"decoder.up{k - 1}.conv.conv_conv.4"
            # the 1x1 conv is the only consumer of the block output below (the bottleneck also feeds the dense head of UNet_Plus)
            stats_prev = prev if (k > 1 or dfeat4 is None) else None
            gu = self._act_plain(self.dU[k], c2, su.h, su.w)
            if self._upb_on(su):
                # the dgrad gathers the upsample backward while it stages (and leaves dU + the bias-gradient rows for the weight gradient / the
                # slab reduction): one launch of the chain instead of two, dU written once and read once less
                gsrc = L.Act()
                gsrc.z, gsrc.mode, gsrc.C, gsrc.Hs, gsrc.Ws, gsrc.pstride = L.ptr(dup), L.ACT_UPBWD, c2, su.h, su.w, dup_ps
                self._dgrad(su, gsrc, self.dA[prev], stats_prev, stage_out=self.dU[k], side_sums=self.csum_part[su.name])
                if self._deferred is not None:
                    self._deferred.append((su, gu))
                else:
                    self._wgrad(su, gu)
            else:
                self._run("upbwd:" + su.name, lambda: L.check(self.lib.hpfg_upsample2x_bwd_sums(
                    L.ptr(dup), dup_ps, L.ptr(self.dU[k]), N, su.h, su.w, c2, L.ptr(self.csum_part[su.name]), st),
                    "upsample2x_bwd"))                               # + per-workgroup channel sums of dU: the 1x1 conv's bias gradient
                self._wgrad_dgrad(su, gu, self.dA[prev], stats_prev)
        if dfeat4 is not None:
            self.dA[enc_prefix(4) + ".4"].add_(dfeat4)
        defer = self._deferred is not None

        flush(self._n_enc_desc, len(self._slab_host))
        self._deferred = [] if defer else None      # second batch: the channel-rich encoder layers, beside the thin layers' fused kernels
        if bucket_cb is not None:
            if self._side_used:
                torch.cuda.current_stream(self.dev).wait_stream(self._side)
                self._side_used = False
            self._slab_reduce(self._n_enc_desc, len(self._slab_host))
            self._finalize_bwd_all(self._n_enc_bn, len(self.bn_layers))      # the decoder's dgamma / dbeta are part of bucket 0
            bucket_cb(0)
        # ---- encoder blocks, deepest first
        for lvl in range(4, -1, -1):
            p = enc_prefix(lvl)
            s2, s1 = sp[f"{p}.4"], sp[f"{p}.0"]
            # a block output below the bottleneck also fed the max-pool of the next level: its dP is folded in by the reduction pass
            g2 = self._bn_backward(s2, self.dP[lvl + 1] if (lvl < 4 and s2.name not in self._pool_done) else None)
            self._wgrad_dgrad(s2, g2, self.dA[s1.name], s1.name)
            g1 = self._bn_backward(s1)
            if lvl == 0:
                if s1.name in self.fused_grid:
                    self._fused_bwd(s1, g1, None, None, None)
                else:
                    self._wgrad(s1, g1)
            if lvl > 0:      # (pool_fuse: the max-pool backward into the block output below rides in this dgrad's epilogue)
                self._wgrad_dgrad(s1, g1, self.dP[lvl], pool_of=(enc_prefix(lvl - 1) + ".4") if self.pool_fuse else None)
            if lvl == 2 and self._deferred is not None:
                flush(self._n_thin_enc_desc, self._n_enc_desc)
                self._deferred = None
        tail_first = defer and bucket_cb is None
        if tail_first:
            # the thin encoder layers' slabs come from fused kernels on THIS stream: reduce them before the join, beside the side stream's last
            # launches, instead of behind the join and the finalize launch
            self._slab_reduce(0, self._n_thin_enc_desc)
        if self._side_used:
            torch.cuda.current_stream(self.dev).wait_stream(self._side)
            self._side_used = False
        # every dZ consumer of the pass (the queued weight gradients on the side stream included) has been ordered before this point
        self._finalize_bwd_all(0, self._n_enc_bn if bucket_cb is not None else len(self.bn_layers))
        self._accb_dirty = False
        if bucket_cb is not None:
            self._slab_reduce(0, self._n_enc_desc)
            bucket_cb(1)
        elif not defer:
            self._slab_reduce(0, len(self._slab_host))
        # (defer: the thin layers' slabs were reduced in front of the join above, everything else on the side stream)
        self.bwd_ready = False
