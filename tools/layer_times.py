"""Per-launch times of one captured step (device time stamps, the probe of bench.py) listed layer by layer with the achieved rates.
usage: python tools/layer_times.py [workload]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from hpfg_amd.engine import MarkLog  # noqa: E402
from hpfg_amd.train import GraphedStep  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "mt"
wl = bench.Workload(name, argparse.Namespace(lab=None, unlab=None), torch.device("cuda:0"), os.environ.get("HPFG_MATH", "bf16x3"), None, 0)
engines = wl.unet_engines() or None
for _ in range(2):
    wl.step.step(*wl.inputs, 1)
engines = wl.unet_engines()
logs = {id(e): MarkLog(wl.dev) for e in engines}
for e in engines:
    e.marks = logs[id(e)]


def reset():
    for lg in logs.values():
        lg.n, lg.spans = 0, []


g = GraphedStep(wl.step, list(wl.inputs), warmup=1, alias_inputs=True, before_capture=reset)
acc = {}
for r in range(8):
    g.step(list(wl.inputs), 5 + r)
    torch.cuda.synchronize()
    if r >= 3:
        for ei, e in enumerate(engines):
            for i, (tag, us) in enumerate(logs[id(e)].read_us()):
                acc.setdefault((ei, i, tag), []).append(us)
calib = float(np.median([np.mean(v) for (_, _, t), v in acc.items() if t == "calib"]))
print(f"calibration bracket {calib:.2f} us; engines: {len(engines)}")
for (ei, i, tag), v in acc.items():
    if tag == "calib":
        continue
    us = float(np.mean(v)) - calib
    fam, b, fl, _ = bench.layer_costs(engines[ei], tag)
    extra = f"  {b / us / 1e3:7.1f} GB/s  {3 * fl / us / 1e6:7.1f} TF(x3)" if b > 0 and us > 0 else ""
    print(f"eng{ei} {tag:52s} {us:7.1f} us{extra}")
