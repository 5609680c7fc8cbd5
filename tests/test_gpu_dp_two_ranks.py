"""GPU, TWO ranks on one device (gloo between fresh child processes; the box has one GPU): the data-parallel contract through the REAL engine.

  * global-batch mode (sync_bn = True): "R ranks on shards == one process on the global batch" -- BatchNorm sums of every layer (students
    and train-mode teacher, forward and backward), the loss sums, the gathered Dense_Loss features and the gradient exchange (bucketed from
    inside backward, or one all-reduce after it) cross the ranks; losses, parameters and running statistics after two steps equal the
    single-process run at 1e-5.  Mean-Teacher (one trainable network), CPS (two networks through one _reduce_grads) and HPFG (two U-Net+
    students + teacher on three streams, CutMix, Dense_Loss: BASELINE configs[2] is "DDP over 2/4/8").
    The same with the PEER MAILBOX exchange (p2p: the BatchNorm / loss sums cross the ranks inside the finalize kernels over hipIpc-mapped
    memory, no collective between kernels), eager and -- only possible that way -- captured into the chain of hipGraphs.
  * the gradient exchange through peer windows (hpfg_peer_allreduce_f32: push / reduce / gather kernels behind epoch flags, no RCCL call):
    ragged sizes against the plain sum, the per-rank mode with it == the mean of the shard runs, and the whole step as ONE hipGraph.
  * per-rank BatchNorm (sync_bn = False, what `bench.py --gpus N` times): after ONE step from identical weights the parameters equal the MEAN
    of the two single-process runs on the shards (SGD's first step is linear in the gradient), and the chain of hipGraphs around the eager
    exchange (both overlap settings) equals the eager run.
"""
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
DEV = torch.device("cuda:0")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _two_ranks(tmp_path, tag, world=2, **env_kw):
    out = str(tmp_path / tag)
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", **{k: str(v) for k, v in env_kw.items()})
        procs.append(subprocess.Popen([sys.executable, "-X", "faulthandler", "-m", "tests.dp_rank_worker", out], cwd=ROOT, env=env))
    try:
        for p in procs:
            assert p.wait(timeout=300) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return tuple(torch.load(f"{out}.rank{r}") for r in range(world))


@pytest.mark.parametrize("step,overlap,p2p", [("mt", True, 0), ("mt", False, 0), ("cps", True, 0), ("cps", False, 0), ("hpfg", True, 0), ("hpfg", False, 0),
                                              ("mt", True, 1), ("cps", False, 1), ("hpfg", True, 1)])
def test_two_ranks_equal_the_global_batch(tmp_path, step, overlap, p2p):
    r0, r1 = _two_ranks(tmp_path, f"{step}{int(overlap)}{p2p}", HPFG_TEST_STEP=step, HPFG_TEST_OVERLAP=int(overlap), HPFG_TEST_SYNC_BN=1, HPFG_TEST_P2P=p2p)
    ref = W.run(DEV, None, 0, 1, step=step)
    for got in (r0, r1):
        assert maxerr(got[0], ref[0]) < 1e-5, (got[0], ref[0])          # loss parts are normalised by the GLOBAL counts on every rank
        for a, b in zip(got[1:], ref[1:]):
            assert maxerr(a, b) < 1e-5                                   # parameters (both students, teacher) / running statistics
    assert torch.equal(r0[1], r1[1]) and torch.equal(r0[2], r1[2])


@pytest.mark.parametrize("step", ["mt", "cps", "hpfg"])
def test_per_rank_batchnorm_mode_averages_the_shard_gradients(tmp_path, step):
    """sync_bn = False: every rank is the single-GPU run on its own shard up to the gradient, which is averaged.  One SGD step from the
    same weights is linear in the gradient, so the parameters must equal the mean of the two single-process shard runs; the BatchNorm
    running statistics stay each rank's own."""
    r0, r1 = _two_ranks(tmp_path, f"ddp_{step}", HPFG_TEST_STEP=step, HPFG_TEST_OVERLAP=0, HPFG_TEST_SYNC_BN=0, HPFG_TEST_STEPS=1)
    s0 = W.run(DEV, None, 0, 1, steps=1, step=step, shard=(0, 2))
    s1 = W.run(DEV, None, 0, 1, steps=1, step=step, shard=(1, 2))
    assert torch.equal(r0[1], r1[1])                                     # the ranks stay in lockstep
    assert maxerr(r0[1], 0.5 * (s0[1] + s1[1])) < 2e-6
    if step != "mt":
        assert torch.equal(r0[2], r1[2])
        if step == "cps":
            assert maxerr(r0[2], 0.5 * (s0[2] + s1[2])) < 2e-6          # (HPFG: model2 is also pulled towards model1 by the backbone EMA -- still linear)
        else:
            assert maxerr(r0[2], 0.5 * (s0[2] + s1[2])) < 2e-6
    assert maxerr(r0[0], s0[0]) < 1e-6 and maxerr(r1[0], s1[0]) < 1e-6   # per-rank losses ARE the shard runs' losses


@pytest.mark.parametrize("step,overlap", [("mt", True), ("mt", False), ("cps", False), ("hpfg", False)])
def test_per_rank_mode_graph_chain_equals_eager(tmp_path, step, overlap):
    """What bench.py runs for N > 1: [forward + loss + backward] | eager gradient exchange | [SGD + EMA] as hipGraphs (with the bucket
    chain for the one-network step when overlap is on) -- same parameters as the eager two-rank run of the same two iterations."""
    common = dict(HPFG_TEST_STEP=step, HPFG_TEST_OVERLAP=int(overlap), HPFG_TEST_SYNC_BN=0, HPFG_TEST_STEPS=2)
    g0, g1 = _two_ranks(tmp_path, f"g_{step}{int(overlap)}", HPFG_TEST_GRAPH=1, **common)
    e0, e1 = _two_ranks(tmp_path, f"e_{step}{int(overlap)}", HPFG_TEST_GRAPH=0, HPFG_TEST_FIXED=1, **common)
    assert torch.equal(g0[1], g1[1])
    for a, b in zip(g0[1:], e0[1:]):
        assert maxerr(a, b) < 1e-6
    assert maxerr(g0[0][-1], e0[0][-1]) < 1e-6


@pytest.mark.parametrize("step,overlap", [("mt", True), ("cps", False)])
def test_global_batch_mode_with_peer_mailboxes_captures_into_graphs(tmp_path, step, overlap):
    """sync_bn = True with the peer mailbox exchange: no host code between the kernels of forward + loss + backward, so the step runs as
    hipGraphs around the eager gradient all-reduce -- and equals the eager two-rank run of the same two iterations."""
    common = dict(HPFG_TEST_STEP=step, HPFG_TEST_OVERLAP=int(overlap), HPFG_TEST_SYNC_BN=1, HPFG_TEST_P2P=1, HPFG_TEST_STEPS=2)
    g0, g1 = _two_ranks(tmp_path, f"pg_{step}", HPFG_TEST_GRAPH=1, **common)
    e0, e1 = _two_ranks(tmp_path, f"pe_{step}", HPFG_TEST_GRAPH=0, HPFG_TEST_FIXED=1, **common)
    assert torch.equal(g0[1], g1[1])
    for a, b in zip(g0[1:], e0[1:]):
        assert maxerr(a, b) < 1e-6
    assert maxerr(g0[0][-1], e0[0][-1]) < 1e-6


@pytest.mark.parametrize("world", [2, 4])
def test_peer_window_allreduce_ragged_sizes(tmp_path, world):
    """(four ranks -- four processes on this one GPU -- exercise the slice / flag indexing beyond a pair; sums are formed in rank order)"""
    res = _two_ranks(tmp_path, f"ar{world}", world=world, HPFG_TEST_STEP="allreduce")
    k = 0
    for rep in range(2):
        for n in W.ALLREDUCE_SIZES:
            want = torch.zeros(n)
            for r in range(world):
                want = want + W.allreduce_case(r, n)
            for got in res:
                assert torch.equal(got[k], want), (rep, n)
            k += 1
    for n in W.ALTERNATING:          # sizes alternate, ranks alternately delayed: the window layout does not depend on the call's size
        want = torch.zeros(n)
        for r in range(world):
            want = want + W.allreduce_case(r, n)
        for got in res:
            assert torch.equal(got[k], want), ("alternating", k, n)
        k += 1


@pytest.mark.parametrize("step", ["mt", "cps", "hpfg"])
def test_peer_gradient_exchange_averages_the_shard_gradients(tmp_path, step):
    """sync_bn = False with the gradient all-reduce through the peer windows: same contract as the collective (mean of the shard runs)."""
    r0, r1 = _two_ranks(tmp_path, f"pg_{step}", HPFG_TEST_STEP=step, HPFG_TEST_OVERLAP=0, HPFG_TEST_SYNC_BN=0, HPFG_TEST_STEPS=1, HPFG_TEST_P2P_GRADS=1)
    s0 = W.run(DEV, None, 0, 1, steps=1, step=step, shard=(0, 2))
    s1 = W.run(DEV, None, 0, 1, steps=1, step=step, shard=(1, 2))
    assert torch.equal(r0[1], r1[1])
    assert maxerr(r0[1], 0.5 * (s0[1] + s1[1])) < 2e-6
    if step != "mt":
        assert torch.equal(r0[2], r1[2]) and maxerr(r0[2], 0.5 * (s0[2] + s1[2])) < 2e-6
    assert maxerr(r0[0], s0[0]) < 1e-6 and maxerr(r1[0], s1[0]) < 1e-6


@pytest.mark.parametrize("step,sync_bn,overlap", [("mt", 0, 0), ("cps", 0, 0), ("mt", 1, 0), ("mt", 0, 1), ("mt", 1, 1), ("cps", 1, 1), ("hpfg", 1, 1)])
def test_peer_gradient_exchange_makes_the_step_one_graph(tmp_path, step, sync_bn, overlap):
    """No host-launched collective left (sync_bn = 1: BatchNorm / loss sums through the mailboxes as well): forward + loss + backward + exchange +
    update captured as ONE hipGraph -- and equal to the eager two-rank run of the same two iterations with the host-launched all-reduce."""
    common = dict(HPFG_TEST_STEP=step, HPFG_TEST_SYNC_BN=sync_bn, HPFG_TEST_P2P=sync_bn, HPFG_TEST_STEPS=2)
    # (mt, 1, 1) is what `bench.py --gpus N` runs by default: BatchNorm / loss sums through the mailboxes + bucketed peer-window gradients,
    # one graph.  Two trainable networks (cps, hpfg) with overlap = 1: the second network back-propagates on a forked stream, so its buckets
    # must NOT fork again inside a capture (hipStreamEndCapture faults on such an edge, ROCm 7.2) -- one exchange after both have joined;
    # two exchanges of DIFFERENT sizes back to back (U-Net+ 3.66 M floats after U-Net ... : the window layout comes from the capacity).
    # hpfg + sync_bn: Dense_Loss's feature gather goes through the peer windows too (heads._GatherRows), so cfg3's global-batch mode captures.
    # overlap = 1: the decoder bucket is exchanged on a side stream beside the encoder half of backward, still inside the one graph
    g0, g1 = _two_ranks(tmp_path, f"og_{step}{sync_bn}{overlap}", HPFG_TEST_GRAPH=1, HPFG_TEST_P2P_GRADS=1, HPFG_TEST_OVERLAP=overlap, **common)
    e0, e1 = _two_ranks(tmp_path, f"oe_{step}{sync_bn}{overlap}", HPFG_TEST_GRAPH=0, HPFG_TEST_FIXED=1, HPFG_TEST_OVERLAP=0, **common)
    assert torch.equal(g0[1], g1[1])
    for a, b in zip(g0[1:], e0[1:]):
        assert maxerr(a, b) < 1e-6
    assert maxerr(g0[0][-1], e0[0][-1]) < 1e-6
