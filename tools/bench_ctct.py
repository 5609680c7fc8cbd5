"""CTCT step (U-Net + SegFormer-B0, BASELINE.json configs[4]: 8 + 24 images of 224x224 per step) timed on one GPU (diagnostics)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpfg_amd.datasets.synthetic import synth_batch  # noqa: E402
from hpfg_amd.model import build_model  # noqa: E402
from hpfg_amd.train import CTCTStep  # noqa: E402
from hpfg_amd.utils import loadyaml  # noqa: E402

DEV = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
a = loadyaml(os.path.join(ROOT, "config", "ctct_unet_segformer_30k_224x224_ACDC.yaml"))
torch.manual_seed(a.seed)
m1, m2 = build_model(a.model1).to(DEV), build_model(a.model2).to(DEV)
m1.train(), m2.train()
xl, yl = synth_batch(10, 8, 224, 224, 1, 4, 32)
xu, _ = synth_batch(11, 24, 224, 224, 1, 4, 32)
xl, yl, xu = xl.to(DEV), yl.to(DEV), xu.to(DEV)
st = CTCTStep(m1, m2, a)
n = int(os.environ.get("STEPS", "20"))
for i in range(5):
    st.step(xl, yl, xu, i + 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    st.step(xl, yl, xu, 6 + i)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print(f"CTCT 8+24 x 224^2: {ms:.3f} ms/step = {32 / ms * 1e3:.0f} img/s")
res = {"workload": "CTCT U-Net + SegFormer-B0 cross teaching (BASELINE configs[4]): 8 labelled + 24 unlabelled 224x224 per step, both networks forward + backward "
                   "on all 32 images, FusedSGD + FusedAdamW", "math": m1.math, "eager_ms_per_step": round(ms, 3), "eager_images_per_s": round(32 / ms * 1e3, 1)}
if os.environ.get("GRAPH", "1") == "1":
    from hpfg_amd.train import GraphedStep  # noqa: E402
    try:
        gs = GraphedStep(st, [xl, yl, xu], warmup=2, alias_inputs=True)
        for i in range(5):
            gs.step([xl, yl, xu], 100 + i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            gs.step([xl, yl, xu], 200 + i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"CTCT 8+24 x 224^2, hipGraph: {ms:.3f} ms/step = {32 / ms * 1e3:.0f} img/s")
        res.update(hipgraph_ms_per_step=round(ms, 3), hipgraph_images_per_s=round(32 / ms * 1e3, 1))
    except Exception as e:
        print("hipGraph capture of the CTCT step failed:", type(e).__name__, str(e)[:300])
import json  # noqa: E402
print(json.dumps(res))
