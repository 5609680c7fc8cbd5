"""Soak run of the captured Mean-Teacher step (GPU only, diagnostics): N replays on synthetic data; checks that the loss stays finite and falls,
that num_batches_tracked of both networks equals the number of forwards and that the engines' dropout seed words advanced once per replay."""
import os
import sys
from copy import deepcopy

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
W = 3
r = GraphedStep(step, [xl, yl, xu], warmup=W, alias_inputs=True)
n = int(os.environ.get("STEPS", "400"))
losses = []
for i in range(n):
    out = r.step([xl, yl, xu], W + 1 + i)
    if i % 50 == 0 or i == n - 1:
        losses.append(float(out["loss"]))
torch.cuda.synchronize()
print("losses", [round(v, 4) for v in losses])
assert all(v == v and abs(v) < 1e4 for v in losses) and losses[-1] < losses[0]
for name, m in (("student", model), ("teacher", ema)):
    nbt = {int(b) for k, b in m.named_buffers() if k.endswith("num_batches_tracked")}
    seeds = [int(e.seed_dev) for pool in m._engines.values() for e in pool]
    print(name, "num_batches_tracked", nbt, "seed words", seeds)
    assert nbt == {W + n} and seeds == [W + n]
assert all(torch.isfinite(p).all() for p in model.parameters()) and all(torch.isfinite(p).all() for p in ema.parameters())
print("soak OK")
