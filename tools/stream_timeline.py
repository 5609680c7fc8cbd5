"""Phase timeline of the captured Mean-Teacher step WITHOUT a profiler attached (diagnostics).

HPFG_STEP_MARKS=1 makes the step enqueue one-thread timestamp kernels (s_memrealtime, 100 MHz) at its phase boundaries; they are
captured into the hipGraph like any other node.  Marks: 0 step start, 1 student forward start, 2/3 teacher forward start/end
(second stream), 4 student forward end, 5 loss start (after the join), 6 backward end, 7 update end.
Run:  python tools/stream_timeline.py
"""
import os
import sys
from copy import deepcopy

import torch

os.environ.setdefault("HPFG_STEP_MARKS", "1")
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
rows = []
for i in range(40):
    r.step([xl, yl, xu], 4 + i)
    if i >= 30:
        torch.cuda.synchronize()          # marks are overwritten by the next replay: read them step by step at the end of the run
        rows.append(step.marks.cpu().numpy().copy())
names = ["start", "student fwd start", "teacher fwd start", "teacher fwd end", "student fwd end", "loss start", "bwd end", "update end"]
for m in rows[-3:]:
    t0 = m[0]
    print("  ".join(f"{n} {(int(m[k]) - int(t0)) / 100.0:.1f}us" for k, n in enumerate(names) if m[k]))
