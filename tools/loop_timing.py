"""Throughput of the DROP-IN LOOPS next to bench.py's replayed step (the reference's hot loop IS the loop,
2017_03_NIPS_Mean-Teacher_ACDC.py:82-113 / sup_ACDC.py:83-93).

For `Mean_Teacher(...)` and `Supervise(...)` of hpfg_amd/train.py: ITERS iterations with (a) the device-resident loader
(`build_loader("device_synthetic")`: slices in HBM, one augmentation kernel per batch) and (b) a host loader that hands over CPU tensors
like the reference's DataLoader (pinned batches of the same shape, so the copy into the step's static buffers is the only extra work), and
(c) the bare captured step replayed on resident inputs, the way bench.py times it, on the same box.  Prints ms per iteration of each.

usage: python tools/loop_timing.py [iters] > profiles/r05_loop_timing.txt
"""
import os
import sys
import time
from copy import deepcopy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hpfg_amd.datasets import build_loader  # noqa: E402
from hpfg_amd.datasets.synthetic import synth_batch  # noqa: E402
from hpfg_amd.model import build_model, reset_dropout_streams  # noqa: E402
from hpfg_amd.train import GraphedStep, Mean_Teacher, MeanTeacherStep, Supervise, SupervisedStep, batch_pair  # noqa: E402
from hpfg_amd.utils import loadyaml  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
DEV = torch.device("cuda:0")


class HostLoader:
    """A DataLoader-shaped source of pinned CPU batches (image float32 [B,1,H,W], mask uint8 [B,H,W]): what the reference's loaders yield."""

    def __init__(self, seed, batch, n_batches, size=224):
        self.batches = []
        for k in range(8):          # eight distinct batches, cycled
            x, y = synth_batch(seed + k, batch, size, size, 1, 4, 32)
            self.batches.append((x.pin_memory(), y.pin_memory()))
        self.n = n_batches
        self.dataset = list(range(batch * n_batches))

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            yield self.batches[i % len(self.batches)]


class Limited:
    """Exactly n batches of `loader` per epoch (restarting it as often as needed), with a length that keeps the driver loops to ONE epoch."""

    def __init__(self, loader, n, warm=50):
        self.loader, self.n, self.warm, self.t_warm = loader, n, warm, None
        self.dataset = getattr(loader, "dataset", None)

    def __len__(self):
        return 10 ** 9

    def __iter__(self):
        k = 0
        while k < self.n:
            for b in self.loader:
                if k == self.warm:          # the steady state starts here: iterations 1 (eager) and 2 (capture) are long past
                    torch.cuda.synchronize()
                    self.t_warm = time.perf_counter()
                yield b
                k += 1
                if k >= self.n:
                    return


def args_for(cfg_name, iters, lab, unlab):
    a = loadyaml(os.path.join(ROOT, "config", cfg_name))
    a.device = DEV
    a.batch_size, a.unlabel_batch_size = lab, unlab
    a.total_itrs = 30000              # (the loaders end the run: Limited)
    a.num_labeled, a.num_unlabeled = 64, 256
    a.log_every = 50
    return a


def models(a, teacher):
    reset_dropout_streams()
    torch.manual_seed(a.seed)
    m = build_model(a).to(DEV)
    m.train()
    if not teacher:
        return m, None
    e = deepcopy(m)
    for p in e.parameters():
        p.requires_grad = False
    e.train()
    return m, e


def timed(fn, lim=None):
    """(ms per iteration over the whole run, ms per iteration from iteration `lim.warm` on)"""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = fn()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    steady = (t1 - lim.t_warm) / (n - lim.warm) * 1e3 if lim is not None and lim.t_warm is not None else None
    return (t1 - t0) / n * 1e3, steady


def run_loop(kind, loader_kind, iters):
    if kind == "mt":
        a = args_for("mean_teacher_unet_30k_224x224_ACDC.yaml", iters, 8, 8)
        m, e = models(a, True)
        if loader_kind == "device":
            a.datasets = "device_synthetic"
            lab, unl, _ = build_loader(a)
        else:
            lab, unl = HostLoader(10, 8, 64), HostLoader(500, 8, 64)
        lim = Limited(unl, iters)
        return timed(lambda: len(Mean_Teacher(m, e, lab, lim, None, a)), lim)
    a = args_for("unet_30k_224x224_ACDC.yaml", iters, 8, 0)
    m, _ = models(a, False)
    if loader_kind == "device":
        a.datasets = "device_synthetic"
        a.unlabel_batch_size = 8
        lab, _, _ = build_loader(a)
    else:
        lab = HostLoader(10, 8, 64)
    lim = Limited(lab, iters)
    return timed(lambda: len(Supervise(m, lim, None, a)), lim)


def run_bare(kind, iters):
    """The captured step replayed on inputs resident in HBM (bench.py's timed region)."""
    if kind == "mt":
        a = args_for("mean_teacher_unet_30k_224x224_ACDC.yaml", iters, 8, 8)
        m, e = models(a, True)
        st = MeanTeacherStep(m, e, a, None)
        xl, yl = synth_batch(1234, 8, 224, 224, 1, 4, 32)
        xu, _ = synth_batch(91234, 8, 224, 224, 1, 4, 32)
        xl, xu = batch_pair(xl.to(DEV), xu.to(DEV))
        inputs = [xl, yl.to(DEV), xu]
    else:
        a = args_for("unet_30k_224x224_ACDC.yaml", iters, 8, 0)
        m, _ = models(a, False)
        st = SupervisedStep(m, a, None)
        x, y = synth_batch(1234, 8, 224, 224, 1, 4, 32)
        inputs = [x.to(DEV), y.to(DEV)]
    g = GraphedStep(st, inputs, warmup=3, alias_inputs=True)
    for k in range(10):
        g.step(inputs, 4 + k)

    def go():
        for k in range(iters):
            g.step(inputs, 20 + k)
        return iters
    return timed(go)[0]


if __name__ == "__main__":
    print(f"# tools/loop_timing.py, one MI355X, {ITERS} iterations per run.  A loop's run contains iteration 1 (eager: allocates the engines' workspaces) and\n"
          f"# iteration 2 (captures the step into a hipGraph), ~0.2 s together; 'steady' = iterations 51 .. {ITERS} of the same run.  ms per iteration:")
    for kind, name in (("mt", "Mean_Teacher (8 + 8 x 224^2)"), ("sup", "Supervise (8 x 224^2)")):
        bare = run_bare(kind, ITERS)
        d, h = run_loop(kind, "device", ITERS), run_loop(kind, "host", ITERS)
        bare2 = run_bare(kind, ITERS)
        b = 0.5 * (bare + bare2)
        print(f"{name}: captured step on resident inputs (bench.py's timed region) {bare:.4f} (before the loops), {bare2:.4f} (after)\n"
              f"    loop, device-resident loader (build_loader('device_synthetic')): whole run {d[0]:.4f}, steady {d[1]:.4f} ({100 * (d[1] / b - 1):+.1f} % vs the bare step)\n"
              f"    loop, host loader (pinned CPU batches, one copy per input into the step's static buffers): whole run {h[0]:.4f}, steady {h[1]:.4f} "
              f"({100 * (h[1] / b - 1):+.1f} %)", flush=True)
