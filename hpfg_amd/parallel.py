"""Data parallelism for the hot path: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The reference is single-process / single-device (main.py:44 says so); sharding is new functionality whose contract is
"R ranks on per-rank batches == one process on the concatenated global batch".  Three exchanges make that exact:
  1. BatchNorm statistics: every BN layer's [sum z, sum z^2] (forward) and [sum g, sum g*xhat] (backward), 2*C fp64 values,
     all-reduced between the partial-sum kernel and the finalize kernel (engine.py) -- train-mode teacher included;
  2. loss sums: the CE/Dice/MSE partial sums (32 floats) all-reduced before the scalar finalize (utils/loss.py);
  3. gradients: SUM all-reduce (the loss is already normalised by the global counts) of the flat fp32 gradient buffer of every
     trainable model (7.26 MB U-Net / 14.65 MB U-Net+) in TWO buckets overlapped with backward: the decoder's slice (+ the
     projection necks) is all-reduced on a side HIP stream as soon as the decoder half of engine.backward has finished it, while
     the encoder half still computes; the encoder's slice follows at the end (`launch_bucket` / `join_buckets`).  xGMI is
     point-to-point (7 links x ~153 GB/s): two ~3.6 MB messages keep every link busy without paying the ring latency more than
     twice per model and step.
EMA, SGD and the LR/ramp-up scalars stay per rank (parameters are bit-identical after the reduced step).

The ~90 small exchanges of (1) and (2) per step are latency-bound as collectives (tens of microseconds of launch + ring latency each,
and no collective may sit inside a captured hipGraph: RCCL's watchdog thread polls its events during a capture).  With
`enable_peer_exchange()` they are not collectives at all: the kernels that finalize a BatchNorm layer / reduce the loss sums write
their per-channel sums into every peer's MAILBOX (fine-grained device memory mapped through hipIpc: xGMI stores between the GPUs of a
node) and add the peers' values from their own mailbox in rank order (csrc/peer.h) -- no host code between the kernels, so the whole
forward + loss + backward stays one hipGraph and only the gradient all-reduce (3) is an RCCL call, between two graphs.

The gradient exchange (3) has a second form that needs no RCCL call at all (`enable_peer_grads()`): the same IPC-mapped windows carry the
flat gradient.  xGMI is point to point, so instead of a ring every rank owns one slice: each rank PUSHES its copy of slice p into rank p's
window (7 links busy at once), the owner adds the R contributions in rank order and pushes the sum into every rank's window, and each rank
copies the result back (csrc/peer.hip: hpfg_peer_allreduce_f32 -- three launches behind epoch flags).  No host code, so under data parallel
the whole step -- forward, loss, backward, exchange, SGD + EMA -- is ONE hipGraph, as on one GPU.  `enable_peer_grads()` tests the path on
the actual devices and reports whether every rank passed; bench.py falls back to the RCCL exchange between two graphs if not.

`sync_bn=False` selects the usual DistributedDataParallel semantics instead: BatchNorm statistics and the loss are per rank
(every rank sees exactly what the single-GPU reference run sees: its own 8+8 batch), and the only exchange is the gradient
all-reduce, averaged over ranks.  bench.py's default for N > 1 is the global-batch mode above (`sync_bn=True`, sums through the peer
mailboxes, gradient buckets through the peer windows); `--local-bn` makes this mode the headline, and either way the other one is timed on
the same ranks and printed as `other_bn_mode`.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


class DataParallelContext:
    LOSS_SLOTS = 4          # fused-loss calls per step that can exchange their sums (HPFG / CPS / S4CVNet use two)

    def next_loss_slot(self):
        """(slot, epoch word) of the next seg_loss call of the current step."""
        k = self._loss_call
        if k >= self.LOSS_SLOTS:
            raise RuntimeError(f"more than {self.LOSS_SLOTS} exchanged losses in one step (StepScalars.push() begins a step)")
        self._loss_call = k + 1
        return self.loss_slot + k, self.loss_epoch[k:k + 1]

    def __init__(self, group=None, device: Optional[torch.device] = None):
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = device
        self.force_sync = False     # tests: issue the collectives (and the sync code path of the engines) even with one rank
        self.sync_bn = True         # False: per-rank BatchNorm statistics and loss, gradients averaged (DDP semantics)
        self.overlap = True         # gradient buckets all-reduced on a side stream from inside backward (False: one blocking exchange after it)
        self.bucket_hook = None     # GraphedStep sets it while capturing: a bucket boundary then ends one hipGraph and starts the next
        self._side = None
        self._pending = False
        # peer mailbox exchange (enable_peer_exchange): BatchNorm / loss sums cross the ranks inside the kernels
        self.p2p = False
        self._mbox = None            # this rank's mailbox (raw device pointer), _peers[r] = rank r's mailbox as mapped here
        self._peers = []
        self._slots_used = 0
        self.peer_cap = 512          # payload values per rank in a slot: [2][C <= 256]
        self.peer_slots = 0
        self.peer_err = None         # device int32 word: set by a kernel whose poll for a peer's value expired
        self.loss_slot = -1          # first of LOSS_SLOTS mailbox slots: one per seg_loss call of a step (call k uses slot k and epoch word k, so
        self.loss_epoch = None       # two losses of one step never share a slot, whichever streams they are issued on)
        self._loss_call = 0          # index of the next seg_loss call inside the current step (reset by StepScalars.push / GraphedStep)
        # peer gradient exchange (enable_peer_grads): the flat gradient crosses the ranks through IPC windows, inside the captured step
        self.p2p_grads = False
        self._gwin = None            # this rank's window, _gpeers[r] = rank r's window as mapped here
        self._gpeers = []
        self.grad_floats = 0
        self.grad_epoch = None

    @property
    def active(self) -> bool:
        return self.world_size > 1 or self.force_sync

    def launch_bucket(self, t: torch.Tensor):
        """All-reduce (SUM) the gradient slice `t`, whose producers are queued on the current stream, without blocking that stream:
        the collective runs on a side stream that waits for the producers; `join_buckets()` makes the consumers wait for it."""
        if not self.active:
            return
        if self.p2p_grads and t.is_cuda:
            # peer windows: the exchange is three launches on the side stream, forked from and joined into the step's stream -- capturable, so
            # the bucket runs beside the encoder half of backward INSIDE the step's one hipGraph (buckets follow each other on the side stream:
            # one window, one epoch sequence)
            main = torch.cuda.current_stream(t.device)
            if self._side is None:
                self._side = torch.cuda.Stream(device=t.device)
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                self.peer_allreduce_sum(t)
            t.record_stream(self._side)
            self._pending = True
            return
        if self.bucket_hook is not None:
            self.bucket_hook(t)
            return
        if not t.is_cuda:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            return
        main = torch.cuda.current_stream(t.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=t.device)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        t.record_stream(self._side)
        self._pending = True

    def join_buckets(self):
        if self._pending:
            torch.cuda.current_stream(self._side.device).wait_stream(self._side)
            self._pending = False

    def allreduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        if self.world_size > 1 or self.force_sync:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def barrier(self):
        if self.device is not None and self.device.type == "cuda" and dist.get_backend(self.group) == "nccl":
            dist.barrier(group=self.group, device_ids=[self.device.index])
        else:
            dist.barrier(group=self.group)

    def max_float(self, v: float) -> float:
        t = torch.tensor([v], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    # ---- peer mailbox exchange ----------------------------------------------------------------------------------------------------------
    def enable_peer_exchange(self, n_slots: int = 256):
        """Allocate this rank's mailbox, exchange the IPC handles over the process group and map every peer's mailbox.  Collective: all
        ranks call it, once, before the first forward.  With one rank it only switches the code path on (the kernels skip the exchange)."""
        import ctypes as C
        from . import _lib as L
        if self.p2p:
            return
        if self.world_size > 8:
            raise RuntimeError("peer mailbox exchange serves one node (<= 8 ranks)")
        if self.device is None or self.device.type != "cuda":
            raise RuntimeError("peer mailbox exchange needs CUDA (HIP) devices")
        lib = L.load()
        self.peer_slots = n_slots
        self._slot_bytes = lib.hpfg_peer_slot_bytes(self.world_size, self.peer_cap)
        self.peer_err = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.loss_epoch = torch.zeros(self.LOSS_SLOTS, dtype=torch.int32, device=self.device)
        self._peers = [None] * self.world_size
        if self.world_size > 1:
            torch.cuda.set_device(self.device)
            ptr = C.c_void_p()
            L.check(lib.hpfg_peer_alloc(self._slot_bytes * n_slots, C.byref(ptr)), "peer_alloc")
            self._mbox = ptr.value
            h = C.create_string_buffer(64)
            L.check(lib.hpfg_peer_handle(self._mbox, h), "peer_handle")
            handles = [None] * self.world_size
            dist.all_gather_object(handles, bytes(h.raw), group=self.group)
            for r, hr in enumerate(handles):
                if r == self.rank:
                    self._peers[r] = self._mbox
                    continue
                q = C.c_void_p()
                L.check(lib.hpfg_peer_open(C.create_string_buffer(hr, 64), C.byref(q)), f"peer_open[{r}]")
                self._peers[r] = q.value
            dist.barrier(group=self.group)          # every mailbox is mapped everywhere before any kernel stores into one
        self.p2p = True
        self.loss_slot = self.alloc_slots(self.LOSS_SLOTS)

    def enable_peer_grads(self, max_floats: int) -> bool:
        """Allocate and map the gradient windows (buffers of up to `max_floats` fp32 values), run one all-reduce through them and return
        whether EVERY rank got the right sums.  Collective: all ranks call it, once, with the same size.  On False nothing is enabled (the
        caller keeps the RCCL exchange); a failure to map a peer's memory counts as False, not as an error."""
        import ctypes as C
        from . import _lib as L
        if self.p2p_grads:
            return True
        if self.world_size > 8 or self.device is None or self.device.type != "cuda":
            return False
        lib = L.load()
        torch.cuda.set_device(self.device)
        if self.peer_err is None:
            self.peer_err = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.grad_epoch = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.grad_floats = int(max_floats)
        if self.world_size == 1:
            self.p2p_grads = True
            return True
        ok = 1
        self.peer_report = {"windows_mapped": False, "self_test": None}          # what THIS rank saw (bench.py prints every rank's: exchange_path)
        self._gpeers = [None] * self.world_size
        try:
            ptr = C.c_void_p()
            L.check(lib.hpfg_peer_alloc(lib.hpfg_peer_buf_bytes(self.world_size, self.grad_floats), C.byref(ptr)), "peer_alloc(grads)")
            self._gwin = ptr.value
            h = C.create_string_buffer(64)
            L.check(lib.hpfg_peer_handle(self._gwin, h), "peer_handle(grads)")
            mine = bytes(h.raw)
        except Exception:
            ok, mine = 0, b""
        handles = [None] * self.world_size
        dist.all_gather_object(handles, mine, group=self.group)          # (also the point where a rank that could not allocate tells the others)
        if ok and all(len(hr) == 64 for hr in handles):
            for r, hr in enumerate(handles):
                if r == self.rank:
                    self._gpeers[r] = self._gwin
                    continue
                q = C.c_void_p()
                if lib.hpfg_peer_open(C.create_string_buffer(hr, 64), C.byref(q)) != 0:
                    ok = 0
                    break
                self._gpeers[r] = q.value
        else:
            ok = 0
        self.peer_report["windows_mapped"] = bool(ok)
        flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)          # (everything is mapped everywhere, or nobody starts)
        if int(flag.item()) == 1:
            self.p2p_grads = True
            good = False
            try:          # (a rank that fails here still reaches the agreement below)
                n = min(self.grad_floats, 262147)
                t = (torch.arange(n, device=self.device, dtype=torch.float32) % 97.0) * float(self.rank + 1)
                want = (torch.arange(n, device=self.device, dtype=torch.float32) % 97.0) * float(self.world_size * (self.world_size + 1) // 2)
                for _ in range(2):          # twice: the second use of the flags is the one every later step repeats
                    got = self.peer_allreduce_sum(t.clone())
                torch.cuda.synchronize(self.device)
                good = bool(torch.equal(got, want)) and int(self.peer_err.item()) == 0
            except Exception as e:
                import sys
                print(f"[hpfg_amd.parallel] rank {self.rank}: peer-window self-test raised {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            self.peer_report["self_test"] = bool(good)
            flag = torch.tensor([1 if good else 0], dtype=torch.int32, device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            self.p2p_grads = int(flag.item()) == 1
        if not self.p2p_grads:
            self.peer_err.zero_()
            self._close_grad_windows()
        return self.p2p_grads

    def peer_allreduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        """SUM over ranks of the contiguous fp32 tensor t, in place, through the peer windows: kernels on the current stream only."""
        import ctypes as C
        from . import _lib as L
        if self.world_size == 1:
            return t
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() <= self.grad_floats):
            raise RuntimeError(f"peer_allreduce_sum: needs a contiguous fp32 device tensor of at most {self.grad_floats} values (got {tuple(t.shape)}, {t.dtype})")
        lib = L.load()
        pb = L.PeerBuf()
        pb.world, pb.rank, pb.n = self.world_size, self.rank, t.numel()
        pb.slice = lib.hpfg_peer_buf_slice(self.world_size, t.numel())
        pb.stride = lib.hpfg_peer_buf_slice(self.world_size, self.grad_floats)      # layout from the CAPACITY: independent of this call's size
        pb.epoch, pb.err = self.grad_epoch.data_ptr(), self.peer_err.data_ptr()
        for r in range(self.world_size):
            pb.win[r] = self._gpeers[r]
        L.check(lib.hpfg_peer_allreduce_f32(C.byref(pb), t.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream), "peer_allreduce_f32")
        return t

    def _close_grad_windows(self):
        from . import _lib as L
        lib = L.load()
        for r, q in enumerate(self._gpeers):
            if r != self.rank and q:
                lib.hpfg_peer_close(q)
        if self._gwin is not None:
            lib.hpfg_peer_free(self._gwin)
        self._gwin, self._gpeers = None, []

    def alloc_slots(self, n: int) -> int:
        """First of n consecutive mailbox slots (the same numbers on every rank: engines are built in the same order everywhere)."""
        base = self._slots_used
        self._slots_used += n
        if self._slots_used > self.peer_slots:
            raise RuntimeError(f"peer mailbox: out of slots ({self._slots_used} > {self.peer_slots})")
        return base

    def peer_desc(self, slot: int, epoch: torch.Tensor):
        """HpfgPeerX for one exchange: `epoch` = the device int32 word that counts the uses of `slot` (bumped BEFORE the exchanging kernel)."""
        from . import _lib as L
        px = L.PeerX()
        px.world, px.rank, px.slot, px.cap, px.slot_bytes = self.world_size, self.rank, slot, self.peer_cap, self._slot_bytes
        px.epoch, px.err = epoch.data_ptr(), self.peer_err.data_ptr()
        for r in range(self.world_size):
            px.mbox[r] = self._peers[r]
        return px

    def bump(self, epoch: torch.Tensor, stream: int):
        from . import _lib as L
        L.check(L.load().hpfg_word_add(epoch.data_ptr(), 1, stream), "word_add")

    def exchange_report(self, rccl_requested: bool = False) -> list:
        """Every rank's view of how its sums cross the ranks, gathered on all ranks (collective; synchronises): one dict per rank with the
        gradient path that actually runs (peer windows or the RCCL fallback, and why: not mapped / self-test failed / requested), the
        BatchNorm + loss path, and the peer error word (non-zero: a bounded poll expired and a kernel carried on with partial sums)."""
        rep = dict(getattr(self, "peer_report", None) or {"windows_mapped": None, "self_test": None})
        if self.p2p_grads:
            grad = "peer-window"
        elif rccl_requested:
            grad = "rccl (requested)"
        elif rep.get("windows_mapped") is False:
            grad = "rccl-fallback (a peer window could not be allocated / mapped)"
        elif rep.get("self_test") is False:
            grad = "rccl-fallback (peer-window self-test returned a wrong sum on this rank)"
        elif rep.get("windows_mapped") is None:
            grad = "rccl"
        else:
            grad = "rccl-fallback (another rank failed its mapping or self-test)"
        if not getattr(self, "sync_bn", True):
            bn = "none (per-rank BatchNorm and loss)"
        else:
            bn = "peer-mailbox" if self.p2p else "rccl"
        err = int(self.peer_err.item()) if self.peer_err is not None else 0
        mine = {"rank": self.rank, "grad_path": grad, "bn_loss_path": bn, "windows_mapped": rep.get("windows_mapped"), "self_test": rep.get("self_test"),
                "peer_err": err}
        out = [None] * self.world_size
        dist.all_gather_object(out, mine, group=self.group)
        return out

    def check_peer_errors(self):
        """Raise if a kernel gave up waiting for a peer's value (synchronises)."""
        if (self.p2p or self.p2p_grads) and self.peer_err is not None and int(self.peer_err.item()) != 0:
            raise RuntimeError("peer mailbox exchange: a poll for a peer's value expired (a rank is missing or far behind)")

    def shutdown(self):
        if self.p2p_grads and self.world_size > 1 and self._gwin is not None:
            torch.cuda.synchronize(self.device)
            if dist.is_initialized():
                dist.barrier(group=self.group)      # nobody unmaps a window a peer's kernel may still store into
            self._close_grad_windows()
            self.p2p_grads = False
        if self.p2p and self.world_size > 1 and self._mbox is not None:
            from . import _lib as L
            lib = L.load()
            torch.cuda.synchronize(self.device)
            if dist.is_initialized():
                dist.barrier(group=self.group)      # nobody unmaps a mailbox a peer's kernel may still store into
            for r, q in enumerate(self._peers):
                if r != self.rank and q:
                    lib.hpfg_peer_close(q)
            lib.hpfg_peer_free(self._mbox)
            self._mbox, self._peers, self.p2p = None, [], False
        if dist.is_initialized():
            dist.destroy_process_group()


def init_from_env(device: torch.device, backend: Optional[str] = None) -> DataParallelContext:
    """RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT come from torch.distributed.run."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = backend or ("nccl" if device.type == "cuda" else "gloo")
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl":
            kw["device_id"] = device
        dist.init_process_group(backend=backend, **kw)
    return DataParallelContext(None, device)


def shard_batch(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Rank r's contiguous share of a global batch (parity runs split one global batch R ways)."""
    n = t.shape[0]
    assert n % world == 0, f"global batch {n} not divisible by world size {world}"
    k = n // world
    return t[rank * k:(rank + 1) * k]
