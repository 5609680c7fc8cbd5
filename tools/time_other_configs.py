"""Step times of the other BASELINE.json workloads on one GPU (not bench lines: parity-test configurations, timed for DESIGN.md).

cfg0 supervised U-Net 8 x 224^2; cfg2 HPFG (two U-Net+ students + EMA teacher, 16 + 16 x 224^2); cfg3 CPS (two U-Nets, 32 + 32 x 96^2 RGB); cfg4 CTCT (U-Net + SegFormer-B0, 8 + 24 x 224^2).
Eager launches and, where the step captures, the hipGraph replay.
"""
import os
import sys
import time
from copy import deepcopy

import numpy as np
import torch

sys.path.insert(0, ".")
from hpfg_amd.datasets.synthetic import synth_batch  # noqa: E402
from hpfg_amd.model import build_model  # noqa: E402
from hpfg_amd.train import CPSStep, GraphedStep, HPFGStep, SupervisedStep  # noqa: E402
from hpfg_amd.utils import loadyaml  # noqa: E402

DEV = torch.device("cuda:0")


def cfg(name):
    a = loadyaml(os.path.join("config", name))
    a.device = DEV
    return a


def timeit(fn, n=30, warm=5):
    for i in range(warm):
        fn(i + 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(warm + i + 1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def report(tag, imgs, step, inputs, extra=()):
    ms = timeit(lambda k: step.step(*inputs, *extra, k))
    line = f"{tag}: eager {ms:.3f} ms/step = {imgs / ms * 1e3:.0f} img/s"
    try:
        g = GraphedStep(step, list(inputs) + list(extra), warmup=2, alias_inputs=True)
        msg = timeit(lambda k: g.step(list(inputs) + list(extra), k))
        line += f"; hipGraph {msg:.3f} ms/step = {imgs / msg * 1e3:.0f} img/s"
    except Exception as e:       # a step that does not capture still runs eager
        line += f"; hipGraph capture not available ({type(e).__name__})"
    print(line, flush=True)


a = cfg("unet_30k_224x224_ACDC.yaml")
torch.manual_seed(a.seed)
m = build_model(a).to(DEV)
m.train()
x, y = synth_batch(1, 8, 224, 224, 1, 4, 32)
report("cfg0 supervised U-Net, 8 x 224^2", 8, SupervisedStep(m, a), (x.to(DEV), y.to(DEV)))
del m
a = cfg("hpfg_unet_plus_30k_224x224_ACDC.yaml")
torch.manual_seed(a.seed)
m1, m2 = build_model(a.model1).to(DEV), build_model(a.model2).to(DEV)
ema = deepcopy(m2)
for p in ema.parameters():
    p.requires_grad = False
m1.train(), m2.train(), ema.train()
st = HPFGStep(m1, m2, ema, a)
xl, yl = synth_batch(5, 16, 224, 224, 1, 4, 32)
xl1, yl1 = synth_batch(6, 16, 224, 224, 1, 4, 32)
xu, _ = synth_batch(7, 16, 224, 224, 1, 4, 32)
cm = st.make_cutmix_mask(16, (224, 224), rng=np.random.RandomState(1))
report("cfg2 HPFG U-Net+ x2 + teacher, 16 + 16 x 224^2", 32, st, tuple(t.to(DEV) for t in (xl, yl, xl1, yl1, xu, cm)))
del st, m1, m2, ema
torch.cuda.empty_cache()
a = cfg("cps_unet_30k_96x96_LIDC.yaml")
torch.manual_seed(a.seed)
m1, m2 = build_model(a.model1).to(DEV), build_model(a.model2).to(DEV)
m1.train(), m2.train()
xl, yl = synth_batch(8, 32, 96, 96, 3, 2, 12)
xu, _ = synth_batch(9, 32, 96, 96, 3, 2, 12)
report("cfg3 CPS U-Net x2, 32 + 32 x 96^2 RGB", 64, CPSStep(m1, m2, a), tuple(t.to(DEV) for t in (xl, yl, xu)))
del m1, m2
torch.cuda.empty_cache()
from hpfg_amd.train import CTCTStep  # noqa: E402
a = cfg("ctct_unet_segformer_30k_224x224_ACDC.yaml")
torch.manual_seed(a.seed)
m1, m2 = build_model(a.model1).to(DEV), build_model(a.model2).to(DEV)
m1.train(), m2.train()
xl, yl = synth_batch(10, 8, 224, 224, 1, 4, 32)
xu, _ = synth_batch(11, 24, 224, 224, 1, 4, 32)
report("cfg4 CTCT U-Net + SegFormer-B0, 8 + 24 x 224^2", 32, CTCTStep(m1, m2, a), tuple(t.to(DEV) for t in (xl, yl, xu)))
m2.eval()
xe = torch.randn(32, 1, 224, 224, device=DEV)
def fwd(k):
    with torch.no_grad():
        m2(xe)
print(f"SegFormer-B0 alone, eval forward of 32 x 224^2: {timeit(fwd):.3f} ms", flush=True)
