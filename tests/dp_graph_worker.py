"""Child process of tests/test_gpu_dp_path.py::test_sync_path_captures_into_a_graph: a 1-rank nccl group, BatchNorm / loss / gradient
collectives of the Mean-Teacher step captured INTO the hipGraph (the optional HPFG_DP_GRAPH=1 mode).  Prints the losses as JSON.
Its own process because RCCL's watchdog thread polls events while the capture is open: when it loses that race the HIP runtime aborts
the process (seen once in many runs), which must not take the test session with it."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from hpfg_amd import parallel  # noqa: E402
from tests.test_gpu_dp_path import _run  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    ctx = parallel.init_from_env(dev)
    ctx.force_sync = True
    losses, _ = _run(ctx, graphed=True, steps=2)
    print("LOSSES " + json.dumps(losses), flush=True)
    ctx.shutdown()
