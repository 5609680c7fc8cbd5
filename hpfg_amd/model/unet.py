"""U-Net / U-Net+ modules whose forward and backward run on the gfx950 HIP library.

Drop-in for the reference's ``model/unet.py`` classes ``UNet`` and ``UNet_Plus`` (unet.py:155-206): same constructor
signature, same ``state_dict()`` keys and shapes (so reference checkpoints load), same ``.encoder`` / ``.decoder`` attributes
(``main.py:72-76`` iterates their parameters), ``UNet_Plus.val(x)`` and the ``(logits, (g, d), (g, d))`` return value.
The submodules are parameter containers only: the arithmetic is done by ``hpfg_amd.engine.UNetEngine`` through one
``torch.autograd.Function`` per network, not by nn.Conv2d / nn.BatchNorm2d forward calls.

All parameters live in one flat fp32 buffer (and all gradients in another), so that EMA, SGD and the data-parallel gradient
all-reduce are one kernel / one collective each.  There is no CPU path: a non-GPU input raises.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

import os

from .. import _lib as L
from .. import engine as E
from .. import heads

WIDTHS = E.WIDTHS

_INSTANCES = [0]


def reset_dropout_streams(count: int = 0) -> None:
    """Restart the numbering of the per-instance dropout streams: two runs that construct their networks in the same order after
    the same ``torch.manual_seed`` then draw identical masks (run-to-run reproducibility inside one process)."""
    _INSTANCES[0] = int(count)


def _fresh_dropout_seed() -> int:
    """A dropout stream of its own for every network instance (student / teacher / model1 / model2 draw independent masks, as the
    reference's separate nn.Dropout calls do: 2017_03_NIPS_Mean-Teacher_ACDC.py:95-101 feeds both nets the same x, the masks are
    the perturbation).  Derived from a process-wide instance counter, NOT from torch's generator: constructing a model must
    consume the RNG exactly like the reference's constructor so that seeded initial weights stay bit-equal."""
    _INSTANCES[0] += 1
    h = (0x1234567 + _INSTANCES[0] * 0x9E3779B1 + (torch.initial_seed() & 0xFFFFFFFF) * 0x85EBCA6B) & 0xFFFFFFFF      # reads, never advances, torch's generator
    h ^= h >> 15
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    return h & 0x7FFFFFFF


def _block(cin: int, cout: int) -> nn.Module:
    """Parameter container with the key layout of the reference's ConvBlock (unet.py:15-25): conv_conv.{0,1,4,5}."""
    m = nn.Module()
    m.conv_conv = nn.ModuleDict(OrderedDict([
        ("0", nn.Conv2d(cin, cout, 3, padding=1)), ("1", nn.BatchNorm2d(cout)),
        ("4", nn.Conv2d(cout, cout, 3, padding=1)), ("5", nn.BatchNorm2d(cout))]))
    return m


def _down(cin: int, cout: int) -> nn.Module:
    m = nn.Module()
    m.maxpool_conv = nn.ModuleDict(OrderedDict([("1", _block(cin, cout))]))
    return m


def _up(c1: int, c2: int) -> nn.Module:
    m = nn.Module()
    m.conv1x1 = nn.Conv2d(c1, c2, 1)
    m.conv = _block(2 * c2, c2)
    return m


def _neck(cin: int, hid: int, out: int = 128) -> nn.Module:
    """projection_conv parameters (unet.py:125-138): mlp.{0,2} Linear, mlp_conv.{0,2} 1x1 conv."""
    m = nn.Module()
    m.mlp = nn.ModuleDict(OrderedDict([("0", nn.Linear(cin, hid)), ("2", nn.Linear(hid, out))]))
    m.mlp_conv = nn.ModuleDict(OrderedDict([("0", nn.Conv2d(cin, hid, 1)), ("2", nn.Conv2d(hid, out, 1))]))
    return m


class _UNetFn(torch.autograd.Function):
    """Whole-network autograd node: forward = engine.forward, backward = engine.backward into the flat gradient buffer."""

    @staticmethod
    def forward(ctx, anchor: torch.Tensor, x: torch.Tensor, net: "UNet", want_feat: bool, needs_grad: bool):
        eng = net._acquire_engine(x)
        if getattr(net, "_bn_bump_pending", False):
            eng.bump_counters, net._bn_bump_pending = net._flat_long, False
        # (an eval-mode forward draws no masks: it must not consume a seed either, or an evaluation between two training iterations would shift
        # the masks of an eager loop against those of the captured one, whose seed words advance inside the graph)
        logits = eng.forward(x, train=net.training, track_running=True, seed_step=net._next_seed() if net.training else None, needs_grad=needs_grad)
        ctx.net, ctx.eng, ctx.want_feat = net, eng, want_feat
        out = logits.permute(0, 3, 1, 2)
        if want_feat:
            feat = eng.materialize(E.enc_prefix(4) + ".4").permute(0, 3, 1, 2)
            return out, feat
        return out, logits.new_empty(0)

    @staticmethod
    def backward(ctx, dlogits, dfeat):
        net, eng = ctx.net, ctx.eng
        dl = dlogits.permute(0, 2, 3, 1)
        if not dl.is_contiguous():
            dl = dl.contiguous()
        df = None
        if ctx.want_feat and dfeat is not None:
            df = dfeat.permute(0, 2, 3, 1).contiguous()
        dp = net.dp
        cb = None
        # Peer-window buckets fork dp._side from the CURRENT stream.  With two trainable networks the second one back-propagates on a forked
        # stream already, and an event wait between two non-origin streams of a capture makes hipStreamEndCapture fault (ROCm 7.2,
        # tools/nested_fork_probe.py): only the single trainable network of a step (it runs on the origin stream) takes the bucketed path;
        # the others are exchanged once after both backward passes have joined (_StepBase._reduce_grads).
        nested = getattr(dp, "p2p_grads", False) and not getattr(net, "_alone", False)
        if dp is not None and dp.active and getattr(dp, "overlap", False) and getattr(eng, "direct", False) and not nested:
            # data parallel: hand each finished slice of the flat gradient buffer to the all-reduce while backward continues
            cb = lambda i: dp.launch_bucket(net.grad_bucket(i))
            net._buckets_launched = True
        eng.backward(dl, df, cb)
        net._accumulate_grads(eng)
        return None, None, None, None, None


class UNet(nn.Module):
    def __init__(self, in_channels: int = 1, num_classes: int = 4):
        super().__init__()
        self.in_channels, self.num_classes = in_channels, num_classes
        enc = nn.Module()
        enc.in_conv = _block(in_channels, WIDTHS[0])
        for k in range(1, 5):
            setattr(enc, f"down{k}", _down(WIDTHS[k - 1], WIDTHS[k]))
        dec = nn.Module()
        for k in range(1, 5):
            setattr(dec, f"up{k}", _up(WIDTHS[5 - k], WIDTHS[4 - k]))
        dec.out_conv = nn.Conv2d(WIDTHS[0], num_classes, 3, padding=1)
        self.encoder, self.decoder = enc, dec
        self._init_runtime()

    # ---- flat storage -----------------------------------------------------------------------------------------
    def _init_runtime(self):
        self._flat: Optional[torch.Tensor] = None
        self._flat_grad: Optional[torch.Tensor] = None
        self._flat_buf: Optional[torch.Tensor] = None
        self._flat_long: Optional[torch.Tensor] = None
        self._engines: Dict[Tuple, List[E.UNetEngine]] = {}
        self._anchor: Optional[torch.Tensor] = None
        self._seed_counter = 0
        self.dropout_seed = _fresh_dropout_seed()
        self.dp = None   # hpfg_amd.parallel.DataParallelContext or None
        self.math = os.environ.get("HPFG_MATH", "bf16x3")   # "bf16x3": split-bf16 MFMA, fp32 accumulate (default); "f32": exact fp32 MFMA
        self.external_dropout_masks = None   # optional {conv name: uint8 NHWC keep-mask}: replay masks drawn elsewhere (tests)
        self._flatten()

    def _float_buffers(self):
        return [(n, b) for n, b in self.named_buffers() if b.is_floating_point()]

    def _flatten(self):
        """(Re)pack parameters, their gradients and the float buffers into three flat tensors; tensors become views."""
        ps = list(self.named_parameters())
        if not ps:
            return
        dev = ps[0][1].device
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1).float() for _, p in ps]).to(dev)
            flat_grad = torch.zeros_like(flat)
            off = 0
            self._offsets: Dict[str, Tuple[int, int]] = {}
            for n, p in ps:
                k = p.numel()
                p.data = flat[off:off + k].view(p.shape)
                p.grad = None
                self._offsets[n] = (off, k)
                off += k
            bs = self._float_buffers()
            fb = torch.cat([b.detach().reshape(-1).float() for _, b in bs]).to(dev)
            off = 0
            for n, b in bs:
                k = b.numel()
                b.data = fb[off:off + k].view(b.shape)
                off += k
            ls = [(n, b) for n, b in self.named_buffers() if b.dtype == torch.long]
            fl = torch.stack([b.detach().reshape(()) for _, b in ls]).to(dev) if ls else torch.zeros(0, dtype=torch.long, device=dev)
            for i, (n, b) in enumerate(ls):
                b.data = fl[i]
        self._flat, self._flat_grad, self._flat_buf, self._flat_long = flat, flat_grad, fb, fl
        self._engines = {}
        self._anchor = torch.zeros(1, device=dev, requires_grad=True)
        self._backbone_numel = sum(p.numel() for n, p in ps if n.startswith("encoder.") or n.startswith("decoder."))
        self._encoder_numel = sum(p.numel() for n, p in ps if n.startswith("encoder."))

    def _is_flat(self) -> bool:
        ps = list(self.parameters())
        if self._flat is None or not ps:
            return False
        first, last = ps[0], ps[-1]
        return (first.data_ptr() == self._flat.data_ptr() and
                last.data_ptr() == self._flat.data_ptr() + (self._flat.numel() - last.numel()) * 4)

    def _ensure_flat(self):
        if not self._is_flat():
            self._flatten()

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._flatten()
        return out

    def __deepcopy__(self, memo):
        cls = self.__class__
        new = cls.__new__(cls)
        nn.Module.__init__(new)
        import copy
        for k, v in self.__dict__.items():
            if k in ("_flat", "_flat_grad", "_flat_buf", "_flat_long", "_engines", "_anchor", "dp"):
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        new.dp = self.dp
        new.dropout_seed = _fresh_dropout_seed()      # a copy (EMA teacher) draws its own masks
        new._flat = new._flat_grad = new._flat_buf = new._flat_long = None
        new._engines, new._anchor = {}, None
        new._flatten()
        return new

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self._ensure_flat()
        return out

    @property
    def flat_params(self) -> torch.Tensor:
        self._ensure_flat()
        return self._flat

    @property
    def flat_grads(self) -> torch.Tensor:
        self._ensure_flat()
        return self._flat_grad

    def grad_bucket(self, i: int) -> torch.Tensor:
        """Data-parallel gradient buckets in the order backward finishes them: 0 = decoder (+ projection necks, whose gradients
        torch autograd has written before the U-Net backward starts), 1 = encoder.  Contiguous slices of the flat buffer."""
        return self._flat_grad[self._encoder_numel:] if i == 0 else self._flat_grad[:self._encoder_numel]

    def backbone_numel(self) -> int:
        """Number of leading flat elements that belong to encoder+decoder (main.py:68-76 updates only those)."""
        return self._backbone_numel

    def attach_grad_views(self):
        """Make every p.grad a view of the flat gradient buffer (so torch optimizers see the HIP-computed gradients)."""
        for n, p in self.named_parameters():
            if p.requires_grad:
                off, k = self._offsets[n]
                g = self._flat_grad[off:off + k].view(p.shape)
                if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                    p.grad = g

    def zero_flat_grad(self):
        """One memset -- or none in ``direct_grads`` mode: every backward then OVERWRITES the encoder/decoder gradients in place
        (slots backward never writes, e.g. biases in front of a train-mode BatchNorm, stay at their initial zero), which is what
        a zero_grad() + one backward() per step amounts to.  The fused step classes switch it on; loops that accumulate several
        backward passes into .grad must leave it off."""
        if getattr(self, "direct_grads", False):      # (the projection necks' backward overwrites its slots too: heads._Linear)
            return
        self._flat_grad.zero_()

    # ---- engines ----------------------------------------------------------------------------------------------
    def _next_seed(self):
        """Per-forward dropout seed word.  While a step is being captured into a hipGraph the forward advances the engine's device seed
        word itself (E.SEED_BUMP, inside its first launch), so every replay draws new masks."""
        if getattr(self, "_graph_seed_mode", False):
            self._graph_fwds = getattr(self, "_graph_fwds", 0) + 1      # forwards of this network inside the step being captured
            return E.SEED_BUMP
        self._seed_counter += 1
        return self._seed_counter

    def bump_graph_seed(self):
        """In front of a replay: a forward captured in graph-seed mode advances its engine's seed word itself (E.SEED_BUMP); the host counter
        follows, so that an EAGER step after replays (a ragged last batch in the driver loops) continues the seed sequence instead of
        re-drawing masks the replays already used."""
        self._seed_counter += getattr(self, "_fwds_per_replay", 0)

    def _acquire_engine(self, x: torch.Tensor) -> E.UNetEngine:
        if not x.is_cuda:
            raise RuntimeError("hpfg_amd.UNet runs on the HIP library only: move the input to the GPU (no CPU fallback)")
        self._ensure_flat()
        if x.device != self._flat.device:
            raise RuntimeError(f"input on {x.device} but parameters on {self._flat.device}")
        key = (tuple(x.shape), x.device.index)
        pool = self._engines.setdefault(key, [])
        direct = bool(getattr(self, "direct_grads", False))
        eng = next((e for e in pool if not e.bwd_ready and getattr(e, "direct", False) == direct), None)
        if eng is None and len(pool) >= 4:
            eng = next((e for e in pool if getattr(e, "direct", False) == direct), None)   # graphs never back-propagated are dropped
        if eng is None:
            named = dict(self.named_parameters())
            params = {n: p.data for n, p in named.items()}
            bufs = {n: b.data for n, b in self.named_buffers()}
            # direct mode: the engine's gradient views ARE the flat gradient buffer (no staging copy, no accumulate pass)
            gtmp = self._flat_grad if direct else torch.zeros_like(self._flat)
            grads = {n: gtmp[o:o + k].view(named[n].shape) for n, (o, k) in self._offsets.items()}
            eng = E.UNetEngine(params, bufs, grads, self.in_channels, self.num_classes, x.shape[0], x.shape[2], x.shape[3], x.device)
            eng.gtmp = gtmp
            eng.direct = direct
            pool.append(eng)
        rank = self.dp.rank if self.dp is not None else 0
        eng.base_seed = (self.dropout_seed + 0x632BE5AB * rank) & 0x7FFFFFFF      # every data-parallel rank draws its own masks too
        eng.math = L.MATH_BF16X3 if self.math == "bf16x3" else L.MATH_F32
        eng.defer_wgrad = bool(getattr(self, "defer_wgrad", False))
        eng.pack_overlap = bool(getattr(self, "_alone", False))
        eng.after_layer = getattr(self, "_after_layer", None)
        ext = self.external_dropout_masks
        if isinstance(ext, (list, tuple)):      # tests: one mask set per forward, consumed in order (several forwards per step)
            k = getattr(self, "_ext_mask_idx", 0)
            self._ext_mask_idx = k + 1
            ext = ext[k % len(ext)] if len(ext) else None
        eng.ext_masks = ext or {}
        if self.dp is not None and getattr(self.dp, "sync_bn", True):
            eng.world, eng.allreduce = self.dp.world_size, self.dp.allreduce_sum
            eng.force_sync = bool(getattr(self.dp, "force_sync", False))
            if getattr(self.dp, "p2p", False) and eng.peer is None:      # BatchNorm sums cross the ranks inside the finalize kernels (peer mailboxes)
                eng.peer, eng.peer_base = self.dp, self.dp.alloc_slots(2 * 18)
                eng.xepoch = torch.zeros(1, dtype=torch.int32, device=x.device)
        return eng

    def _accumulate_grads(self, eng: E.UNetEngine):
        if getattr(eng, "direct", False):      # backward wrote straight into the flat gradient buffer
            self.attach_grad_views()
            return
        n = self._backbone_numel
        first = next(iter(self.parameters()))
        if first.grad is None:          # zero_grad(set_to_none=True) semantics: None means zero
            self._flat_grad[:n].copy_(eng.gtmp[:n])
        else:
            self._flat_grad[:n].add_(eng.gtmp[:n])
        self.attach_grad_views()

    def _prepare_head_grads(self):
        """Neck parameters get their gradients from torch autograd; make those accumulate into the flat buffer."""
        n = self._backbone_numel
        heads_ = [p for nme, p in self.named_parameters() if not (nme.startswith("encoder.") or nme.startswith("decoder."))]
        if heads_ and heads_[0].requires_grad and (heads_[0].grad is None or heads_[0].grad.data_ptr() != self._flat_grad.data_ptr() + 4 * n):
            self._flat_grad[n:].zero_()
            for nme, p in self.named_parameters():
                if not (nme.startswith("encoder.") or nme.startswith("decoder.")):
                    off, k = self._offsets[nme]
                    p.grad = self._flat_grad[off:off + k].view(p.shape)

    def _bump_bn_counters(self):
        """num_batches_tracked += 1 of all 18 BatchNorm layers (one flat int64 buffer): done by the forward's first launch, the weight packing
        (UNetEngine.bump_counters -> hpfg_pack_weights_bump), not by a kernel of its own."""
        self._bn_bump_pending = True

    def _run(self, x: torch.Tensor, want_feat: bool):
        self._ensure_flat()
        if self.training:
            self._bump_bn_counters()
        trainable = next(iter(self.parameters())).requires_grad
        anchor = self._anchor if trainable else self._anchor.detach()
        return _UNetFn.apply(anchor, x.float(), self, want_feat, bool(trainable and torch.is_grad_enabled()))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._run(x, False)[0]


class UNet_Plus(UNet):
    def __init__(self, in_channels: int = 1, num_classes: int = 4):
        nn.Module.__init__(self)
        self.in_channels, self.num_classes = in_channels, num_classes
        tmp = UNet.__new__(UNet)
        UNet.__init__(tmp, in_channels, num_classes)       # consumes the RNG exactly like encoder+decoder construction
        self.encoder, self.decoder = tmp.encoder, tmp.decoder
        self.dense_projection_high = _neck(WIDTHS[-1], 2048)
        self.dense_projection_head = _neck(num_classes, 1024)
        self._init_runtime()

    def val(self, x):
        return self._run(x, False)[0]

    def forward(self, x):
        self._prepare_head_grads()
        logits, feat = self._run(x, True)
        if getattr(self, "skip_necks", False):      # set by a step that discards them (HPFG's first student, main.py:152): nothing to compute, no gradient
            return logits, None, None
        direct = bool(getattr(self, "direct_grads", False)) and torch.is_grad_enabled()
        high = heads.projection_neck(self.dense_projection_high, feat, direct=direct)
        head = heads.projection_neck(self.dense_projection_head, logits, direct=direct)
        return logits, high, head
