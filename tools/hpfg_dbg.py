import faulthandler, os, sys
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from copy import deepcopy
import numpy as np, torch
from hpfg_amd.datasets.synthetic import synth_batch
from hpfg_amd.model import build_model
from hpfg_amd.train import HPFGStep
from hpfg_amd.utils import loadyaml
DEV = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
a = loadyaml(os.path.join(ROOT, "config", "hpfg_unet_plus_30k_224x224_ACDC.yaml"))
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
inp = tuple(t.to(DEV) for t in (xl, yl, xl1, yl1, xu, cm))
for k in range(3):
    r = st.step(*inp, k + 1)
    torch.cuda.synchronize()
    print("step", k, float(r["loss"]), flush=True)
import time
from hpfg_amd.train import GraphedStep
for k in range(35):
    st.step(*inp, k + 4)
torch.cuda.synchronize()
print("eager loop ok", flush=True)
g = GraphedStep(st, list(inp), warmup=2, alias_inputs=True)
print("captured", flush=True)
for k in range(35):
    g.step(list(inp), k + 50)
torch.cuda.synchronize()
print("graph loop ok", flush=True)
