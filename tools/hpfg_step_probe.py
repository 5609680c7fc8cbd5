"""A few eager HPFG steps (BASELINE configs[2] shape: two U-Net+ students + EMA teacher, 16 + 16 images of 224 x 224) for a rocprofv3 kernel trace:
evidence that no rocBLAS / MIOpen kernel is left on that step (projection necks and Dense_Loss run on csrc/gemm.hip + csrc/heads.hip)."""
import os
import sys
import time
from copy import deepcopy

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpfg_amd.datasets.synthetic import synth_batch  # noqa: E402
from hpfg_amd.model import build_model  # noqa: E402
from hpfg_amd.train import HPFGStep  # noqa: E402
from hpfg_amd.utils import loadyaml  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = torch.device("cuda:0")
a = loadyaml(os.path.join(ROOT, "config", "hpfg_unet_plus_30k_224x224_ACDC.yaml"))
torch.manual_seed(a.seed)
m1, m2 = build_model(a.model1).to(DEV), build_model(a.model2).to(DEV)
ema = deepcopy(m2)
for p in ema.parameters():
    p.requires_grad = False
m1.train(), m2.train()
st = HPFGStep(m1, m2, ema, a)
nl, nu = a.batch_size, a.unlabel_batch_size
xl, yl = synth_batch(1, nl, 224, 224, 1, 4, 32)
xl1, yl1 = synth_batch(2, nl, 224, 224, 1, 4, 32)
xu, _ = synth_batch(3, nu, 224, 224, 1, 4, 32)
rep = nu // nl
cm = st.make_cutmix_mask(nu, (224, 224), device=DEV)
args = (xl.to(DEV), yl.to(DEV), xl1.repeat(rep, 1, 1, 1).to(DEV), yl1.repeat(rep, 1, 1).to(DEV), xu.to(DEV), cm.to(DEV))
n = int(os.environ.get("STEPS", "6"))
for k in range(3):
    st.step(*args, 1500 + k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(n):
    st.step(*args, 1600 + k)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print(f"HPFG {nl}+{nu} x 224^2 eager: {ms:.3f} ms/step = {(nl + nu) / ms * 1e3:.0f} img/s")
