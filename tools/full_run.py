"""A full-length run of the drop-in Mean_Teacher loop (the reference's 30 000 iterations, 2017_03_NIPS_Mean-Teacher_ACDC.py:63-162) on the device-resident
synthetic loader, with the periodic evaluations of both networks: wall time, loss trace, best Dice (diagnostics / evidence).
usage: python tools/full_run.py [total_itrs] > profiles/r05_full_run.txt"""
import os
import sys
import time
from copy import deepcopy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hpfg_amd.datasets import build_loader  # noqa: E402
from hpfg_amd.model import build_model  # noqa: E402
from hpfg_amd.train import Mean_Teacher  # noqa: E402
from hpfg_amd.utils import loadyaml  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = torch.device("cuda:0")


class Log:
    def __init__(self):
        self.lines = []

    def info(self, m):
        self.lines.append(str(m))


a = loadyaml(os.path.join(ROOT, "config", "mean_teacher_unet_30k_224x224_ACDC.yaml"))
a.device = DEV
a.total_itrs = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
a.step_size = max(1, a.total_itrs // 20)
a.datasets = "device_synthetic"
a.num_labeled, a.num_unlabeled = 64, 256
a.log_every = 500
a.logger = Log()
torch.manual_seed(a.seed)
m = build_model(a).to(DEV)
e = deepcopy(m)
for p in e.parameters():
    p.requires_grad = False
lab, unl, _ = build_loader(a)
b = loadyaml(os.path.join(ROOT, "config", "mean_teacher_unet_30k_224x224_ACDC.yaml"))          # held-out volumes: the "synthetic" key's test split
b.datasets, b.device, b.synthetic_labeled, b.synthetic_unlabeled, b.synthetic_test_volumes = "synthetic", DEV, 8, 8, 4
b.test_crop_size = getattr(b, "test_crop_size", (224, 224))
_, _, test = build_loader(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
log = Mean_Teacher(m, e, lab, unl, test, a)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n = log.numel()
print(f"# Mean_Teacher(...) drop-in loop, {n} iterations of 8 + 8 x 224^2 (device-resident synthetic loader, {len(lab.dataset) if hasattr(lab, 'dataset') else '?'} labelled slices), "
      f"evaluation of student and teacher every {a.step_size} iterations: {dt:.1f} s wall = {dt / n * 1e3:.3f} ms per iteration all included")
idx = [0, 1, 2, 9, 99, 999] + list(range(a.step_size - 1, n, a.step_size)) + [n - 1]
print("loss trace (iteration: loss):", "  ".join(f"{i + 1}: {float(log[i]):.4f}" for i in sorted(set(i for i in idx if i < n))))
print("all losses finite:", bool(torch.isfinite(log).all()), " last < first:", bool(log[-1] < log[0]))
for ln in a.logger.lines[-6:]:
    print("eval:", ln)
assert all(torch.isfinite(p).all() for p in m.parameters()) and all(torch.isfinite(p).all() for p in e.parameters())
print("parameters finite: True")
