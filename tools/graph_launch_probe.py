"""Does the host run ahead of the GPU when a captured training step is replayed?  (diagnostics)

Times each GraphedStep.step() call on the host (no synchronisation inside the loop) and the whole loop with a final synchronise.
If hipGraphLaunch returned as soon as the work is queued, the per-call host time would be far below the GPU time per step.
"""
import os
import sys
import time
from copy import deepcopy

import torch

sys.path.insert(0, ".")
from hpfg_amd.datasets.synthetic import synth_batch  # noqa: E402
from hpfg_amd.model import build_model  # noqa: E402
from hpfg_amd.train import GraphedStep, MeanTeacherStep  # noqa: E402
from hpfg_amd.utils import loadyaml  # noqa: E402

dev = torch.device("cuda:0")
args = loadyaml("config/mean_teacher_unet_30k_224x224_ACDC.yaml")
args.batch_size = args.unlabel_batch_size = 8
torch.manual_seed(1)
model = build_model(args).to(dev)
ema = deepcopy(model)
for p in ema.parameters():
    p.requires_grad = False
model.train()
ema.train()
step = MeanTeacherStep(model, ema, args, None)
xl, yl = synth_batch(1, 8, 224, 224, 1, 4, 32)
xu, _ = synth_batch(2, 8, 224, 224, 1, 4, 32)
xl, yl, xu = xl.to(dev), yl.to(dev), xu.to(dev)
r = GraphedStep(step, [xl, yl, xu], warmup=3, alias_inputs=True)
for i in range(10):
    r.step([xl, yl, xu], 4 + i)
torch.cuda.synchronize()
K = int(os.environ.get("K", "30"))
host = []
t0 = time.perf_counter()
for i in range(K):
    a = time.perf_counter()
    r.step([xl, yl, xu], 20 + i)
    host.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host time per step() call: mean {sum(host) / K * 1e6:.0f} us, min {min(host) * 1e6:.0f}, max {max(host) * 1e6:.0f}; "
      f"loop returned after {(t1 - t0) / K * 1e6:.0f} us/step, GPU done after {(t2 - t0) / K * 1e6:.0f} us/step")
print("first calls (us):", [round(h * 1e6) for h in host[:8]])
# the raw replay alone
g = r.graph
torch.cuda.synchronize()
host = []
for i in range(K):
    a = time.perf_counter()
    g.replay()
    host.append(time.perf_counter() - a)
torch.cuda.synchronize()
print("graph.replay() alone (us):", [round(h * 1e6) for h in host[:10]])
