"""HPFG / CPS steps captured into a hipGraph with the separate weight gradients of BOTH students queued onto side streams
(HPFG_DEFER_ALL=1): the configuration whose capture faulted in round 2 (a student back-propagating on a forked stream was never joined
back).  Prints the losses of a few replays and the step time with and without the deferral.   usage: python tools/hpfg_defer_probe.py"""
import faulthandler
import os
import sys
import time

faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse  # noqa: E402

import torch  # noqa: E402


def run(workload, defer):
    os.environ["HPFG_DEFER_ALL"] = "1" if defer else "0"
    import bench
    a = argparse.Namespace(lab=None, unlab=None)
    wl = bench.Workload(workload, a, torch.device("cuda:0"), "bf16x3", None, 0)
    from hpfg_amd.train import GraphedStep
    g = GraphedStep(wl.step, list(wl.inputs), warmup=2, alias_inputs=True)
    losses = []
    for i in range(5):
        r = g.step(list(wl.inputs), 3 + i)
        losses.append(float(r["loss"]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        g.step(list(wl.inputs), 10 + i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"{workload} defer={defer}: {ms:.3f} ms/step, losses {['%.5f' % v for v in losses]}", flush=True)
    return losses


for w in sys.argv[1:] or ["cps", "hpfg"]:
    a = run(w, False)
    b = run(w, True)
    print(w, "max |loss difference| deferred vs not:", max(abs(x - y) for x, y in zip(a, b)), flush=True)
