"""GPU, TWO ranks on one device (gloo; fresh child processes): the data-parallel contract through the REAL engine --
"R ranks on shards == one process on the global batch" in the --sync-bn mode: BatchNorm sums of every layer (student and train-mode
teacher, forward and backward), the loss sums and the gradient buckets (all-reduced from inside backward on a side stream) cross the
ranks; losses, student and teacher parameters and running statistics after two Mean-Teacher steps equal the single-process run at 1e-5."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from tests import dp_rank_worker as W
from tests.helpers import maxerr

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("overlap", [True, False])
def test_two_ranks_equal_the_global_batch(tmp_path, overlap):
    out = str(tmp_path / "res")
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", HPFG_TEST_OVERLAP="1" if overlap else "0")
        procs.append(subprocess.Popen([sys.executable, "-m", "tests.dp_rank_worker", out], cwd=ROOT, env=env))
    try:
        for p in procs:
            assert p.wait(timeout=300) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    ref_l, ref_p, ref_e, ref_rv = W.run(torch.device("cuda:0"), None, 0, 1)
    r0 = torch.load(f"{out}.rank0")
    r1 = torch.load(f"{out}.rank1")
    for (l, p, e, rv) in (r0, r1):
        assert maxerr(l, ref_l) < 1e-5, (l, ref_l)                       # loss parts are normalised by the GLOBAL counts on every rank
        assert maxerr(p, ref_p) < 1e-5 and maxerr(e, ref_e) < 1e-5      # parameters identical on all ranks and equal to the global run
        assert maxerr(rv, ref_rv) < 1e-5
    assert torch.equal(r0[1], r1[1])
