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

`sync_bn=False` selects the usual DistributedDataParallel semantics instead: BatchNorm statistics and the loss are per rank
(every rank sees exactly what the single-GPU reference run sees: its own 8+8 batch), and the only exchange is the gradient
all-reduce, averaged over ranks.  That removes the ~90 latency-bound 2*C-element collectives per step, which cannot be captured
into a hipGraph here (the RCCL watchdog rejects stream capture) and bound the multi-GPU step; bench.py uses it for N > 1.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


class DataParallelContext:
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

    @property
    def active(self) -> bool:
        return self.world_size > 1 or self.force_sync

    def launch_bucket(self, t: torch.Tensor):
        """All-reduce (SUM) the gradient slice `t`, whose producers are queued on the current stream, without blocking that stream:
        the collective runs on a side stream that waits for the producers; `join_buckets()` makes the consumers wait for it."""
        if not self.active:
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

    def shutdown(self):
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
