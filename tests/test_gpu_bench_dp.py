"""GPU: `bench.py --gpus 2` end to end, both ranks on this box's one GPU (gloo process group; the engines, kernels and hipIpc peer windows are the
real ones) -- the N > 1 code path the driver times on a multi-GPU node: launcher environment, peer-window self-test, one-graph step with the
in-graph gradient exchange, barrier / max-over-ranks timing, one JSON line from rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("flags,expect,sync", [((), "peer-window all-reduce", True), (("--rccl", "--no-overlap", "--local-bn"), "all-reduce between two hipGraphs", False)])
def test_bench_two_ranks_on_one_gpu(flags, expect, sync):
    """No flags = what the driver's scaling run executes: BatchNorm / loss sums through the peer mailboxes (global-batch mode), bucketed
    peer-window gradient exchange overlapped with backward, one hipGraph; the other BatchNorm mode is timed by the same job."""
    env = dict(os.environ, HPFG_BENCH_ONE_DEVICE="1", HPFG_DP_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--blocks", "2", "--no-probe", *flags]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["hipgraph"] is True and out["config"]["sync_bn"] is sync
    assert expect in out["config"]["parallelism"], out["config"]["parallelism"]
    assert len(out["blocks"]["ms_per_step"]) == 2 and out["blocks"]["min_ms"] <= out["blocks"]["median_ms"] <= out["blocks"]["max_ms"]
    other = out["other_bn_mode"]          # the same job times the other BatchNorm mode as well
    assert other["sync_bn"] is (not sync) and other["value"] > 0
    assert out["cpu_baseline"] is None          # reported on rank 0 at N = 1 only
    # every rank's own view of the exchange: which path its gradients and BatchNorm / loss sums took, and its peer error word
    xp = out["exchange_path"]
    assert [e["rank"] for e in xp] == [0, 1] and all(e["peer_err"] == 0 for e in xp)
    if sync:
        assert all(e["grad_path"] == "peer-window" and e["windows_mapped"] and e["self_test"] and e["bn_loss_path"] == "peer-mailbox" for e in xp), xp
    else:
        assert all(e["grad_path"] == "rccl (requested)" and e["bn_loss_path"].startswith("none") for e in xp), xp
