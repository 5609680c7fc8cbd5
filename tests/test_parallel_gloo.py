"""CPU, world_size 2 over gloo: the data-parallel contract of hpfg_amd.parallel -- "R ranks on shards == one process on the global
batch".  The sharded statistics are computed with the CPU oracle on each rank's shard (tests may use the oracle), exchanged with the
PRODUCT's DataParallelContext, and compared with the single-process oracle on the concatenated batch:
  1. BatchNorm: all-reduced [sum z, sum z^2] give the global-batch mean / biased variance;
  2. loss: all-reduced CE/Dice/MSE partial sums give the global-batch Med_Sup_Loss + consistency MSE;
  3. gradients: SUM all-reduce of the per-rank gradients of (global-normalised) losses equals the global-batch gradient."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hpfg_amd import parallel
from oracle import losses_ref


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dp = parallel.init_from_env(torch.device("cpu"), backend="gloo")
    try:
        g = torch.Generator().manual_seed(0)
        z = torch.randn(4, 8, 6, 6, generator=g) * 2 + 0.3            # global batch of raw conv outputs
        logits = torch.randn(4, 4, 6, 6, generator=g)
        tlog = torch.randn(4, 4, 6, 6, generator=g)
        lab = torch.randint(0, 4, (4, 6, 6), generator=g)
        zs, ls, ts, ys = (parallel.shard_batch(t, rank, world) for t in (z, logits, tlog, lab))
        # 1. BN sums
        sums = torch.stack([zs.double().sum((0, 2, 3)), (zs.double() ** 2).sum((0, 2, 3))])
        dp.allreduce_sum(sums)
        cnt = z.numel() / z.shape[1]
        mean, var = sums[0] / cnt, sums[1] / cnt - (sums[0] / cnt) ** 2
        ok_bn = torch.allclose(mean, z.double().mean((0, 2, 3)), atol=1e-9) and torch.allclose(var, z.double().var((0, 2, 3), unbiased=False), atol=1e-9)
        # 2./3. loss sums and gradient of the globally normalised loss w.r.t. the local logits
        x = ls.clone().requires_grad_(True)
        p = torch.softmax(x, 1)
        i, zz, yy = losses_ref.dice_sums(p, ys)
        nll = torch.nn.functional.cross_entropy(x, ys, reduction="sum")
        mse = ((p - torch.softmax(ts, 1)) ** 2).sum()
        loc = torch.cat([i, zz, yy, nll.view(1), mse.view(1), torch.tensor([float(ys.numel())])]).detach().clone()
        dp.allreduce_sum(loc)
        I, Z, Y = loc[0:4], loc[4:8], loc[8:12]
        # d(loss)/d(local sums) evaluated at the GLOBAL sums (what seg_loss_bwd does with the all-reduced sums)
        dice_g = (1 - (2 * (i + (I - i.detach())) + 1e-5) / ((zz + (Z - zz.detach())) + (yy + (Y - yy.detach())) + 1e-5)).mean()
        loss = 0.5 * (nll + (loc[12] - nll.detach())) / loc[14] + 0.5 * dice_g + 0.3 * (mse + (loc[13] - mse.detach())) / (logits.numel())
        loss.backward()
        grads = [torch.zeros_like(logits[: logits.shape[0] // world]) for _ in range(world)]
        dist.all_gather(grads, x.grad)
        xg = logits.clone().requires_grad_(True)
        ref = losses_ref.med_sup_loss(xg, lab) + 0.3 * losses_ref.mse_consistency(torch.softmax(xg, 1), torch.softmax(tlog, 1))
        ref.backward()
        ok_loss = abs(float(loss) - float(ref)) < 1e-6
        ok_grad = torch.allclose(torch.cat(grads), xg.grad, atol=1e-7)
        # parameter-gradient reduction: SUM of per-rank gradients of a shared weight
        w = torch.full((3,), 0.5, requires_grad=True)
        (w * x.grad.detach().sum()).sum().backward()
        gw = w.grad.clone()
        dp.allreduce_sum(gw)
        ok_sum = torch.allclose(gw, torch.full((3,), float(xg.grad.sum())), atol=1e-6)
        tmax = dp.max_float(float(rank))
        dp.barrier()
        q.put((rank, ok_bn, ok_loss, ok_grad, ok_sum, tmax))
    finally:
        dp.shutdown()


def test_two_rank_gloo_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_bn, ok_loss, ok_grad, ok_sum, tmax in res:
        assert ok_bn and ok_loss and ok_grad and ok_sum, (rank, ok_bn, ok_loss, ok_grad, ok_sum)
        assert tmax == 1.0


def test_shard_batch_contract():
    t = torch.arange(12).view(6, 2)
    assert torch.equal(parallel.shard_batch(t, 1, 3), t[2:4])
    with pytest.raises(AssertionError):
        parallel.shard_batch(t, 0, 4)
