"""Soak of the in-graph data-parallel exchanges with two ranks on ONE GPU (gloo process group; hipIpc windows / mailboxes are the real ones):
N replays of the captured Mean-Teacher step with the peer-window gradient all-reduce (and, with SYNC_BN=1, the BatchNorm / loss mailboxes),
then: no poll expired, both ranks hold bit-identical parameters, and the loss fell.  Diagnostics, GPU only.

    HPFG_BENCH_ONE_DEVICE=1 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/dp2_soak.py [steps]
"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, ".")
from hpfg_amd import parallel  # noqa: E402
from hpfg_amd.datasets.synthetic import synth_batch  # noqa: E402
from hpfg_amd.model import UNet  # noqa: E402
from hpfg_amd.train import GraphedStep, MeanTeacherStep  # noqa: E402
from tests.dp_rank_worker import _frozen, opt_args  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dp = parallel.init_from_env(dev, backend="gloo")
dp.sync_bn = os.environ.get("SYNC_BN", "0") == "1"
dp.overlap = False
if dp.sync_bn:
    dp.enable_peer_exchange()
torch.manual_seed(5)
m = UNet(1, 4).to(dev)
m.math = "bf16x3"
ema = _frozen(m)
m.train()
assert dp.enable_peer_grads(int(m.flat_grads.numel())), "peer windows unavailable"
st = MeanTeacherStep(m, ema, opt_args(lr=0.01), dp)
xl, yl = synth_batch(11 + dp.rank, 4, 96, 96, 1, 4, 8)
xu, _ = synth_batch(21 + dp.rank, 4, 96, 96, 1, 4, 8)
inputs = [xl.to(dev), yl.to(dev), xu.to(dev)]
runner = GraphedStep(st, inputs, warmup=2, alias_inputs=True)
assert not runner.split, "the step should be one graph"
dp.barrier()
first = last = None
for i in range(steps):
    out = runner.step(inputs, 3 + i)
    if i == 0:
        first = float(out["loss"])
last = float(out["loss"])
torch.cuda.synchronize()
dp.check_peer_errors()
mine = torch.cat([m.flat_params.detach().cpu(), ema.flat_params.detach().cpu()])
both = [None, None]
dist.all_gather_object(both, mine)
same = bool(torch.equal(both[0], both[1]))
finite = bool(torch.isfinite(mine).all())
if dp.rank == 0:
    print(f"soak: {steps} replays, sync_bn={dp.sync_bn}: loss {first:.4f} -> {last:.4f}, ranks bit-identical: {same}, finite: {finite}, epoch word {int(dp.grad_epoch.item())}")
assert same and finite and last < first
dp.shutdown()
