"""One data-parallel rank of tests/test_gpu_dp_two_ranks.py (started as a child process: RANK / WORLD_SIZE / MASTER_* in the environment).
Runs steps of the REAL engine on this rank's shard of a fixed global batch and saves losses + final parameters.

    HPFG_TEST_STEP     mt | cps | hpfg        which step law (Mean-Teacher: one student + teacher; CPS: two students sharing the gradient
                                              exchange; HPFG: two U-Net+ students + teacher on three streams, CutMix, Dense_Loss)
    HPFG_TEST_SYNC_BN  1 | 0                  global-batch mode (all-reduced BatchNorm statistics, loss sums, gathered contrast features) or
                                              per-rank BatchNorm with averaged gradients (what `bench.py --gpus N` times)
    HPFG_TEST_OVERLAP  1 | 0                  gradient buckets all-reduced from inside backward on a side stream, or one exchange after it
    HPFG_TEST_GRAPH    1 | 0                  the step as a chain of hipGraphs around the eager gradient exchange, as bench.py runs it
    HPFG_TEST_P2P      1 | 0                  (sync_bn = 1) BatchNorm / loss sums exchanged by the kernels through peer mailboxes (hipIpc) instead of
                                              host-launched collectives
    HPFG_TEST_P2P_GRADS 1 | 0                 the gradient all-reduce through the peer windows (kernels on the step's stream; with GRAPH = 1 the whole step
                                              is then ONE hipGraph) instead of a host-launched collective
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from hpfg_amd import engine as E  # noqa: E402
from hpfg_amd import parallel  # noqa: E402
from hpfg_amd.datasets.synthetic import synth_batch  # noqa: E402
from hpfg_amd.model import UNet, UNet_Plus, reset_dropout_streams  # noqa: E402
from hpfg_amd.train import CPSStep, GraphedStep, HPFGStep, MeanTeacherStep  # noqa: E402
from hpfg_amd.utils import AttrDict  # noqa: E402

N_LAB = N_UNL = 4
SIZE = 64


def opt_args(**kw):
    a = dict(opt="sgd", lr=0.01, momentum=0.9, weight_decay=1e-4, sched="medical", total_itrs=30000, step_size=200, warmup_epochs=0,
             warmup_lr=1e-4, min_lr=1e-6, consistency=0.1, consistency_rampup=200.0, ema_decay=0.99)
    a.update(kw)
    return AttrDict(a)


def global_batch():
    xl, yl = synth_batch(11, N_LAB, SIZE, SIZE, 1, 4, 8)
    xu, _ = synth_batch(12, N_UNL, SIZE, SIZE, 1, 4, 8)
    return xl, yl, xu


def image_masks(step, who, n_images):
    """Per-IMAGE dropout keep masks of the five encoder dropout sites (explicit masks: the device RNG is indexed by the position inside
    the local batch, so shards could not reproduce a global batch's draws).  Returns {conv name: uint8 [n_images, h, w, C]}."""
    g = torch.Generator().manual_seed(1000 * step + who)
    out = {}
    for lvl in range(5):
        c, h = E.WIDTHS[lvl], SIZE >> lvl
        out[E.enc_prefix(lvl) + ".0"] = (torch.rand(n_images, h, h, c, generator=g) >= E.ENC_DROPOUT[lvl]).to(torch.uint8)
    return out


def take(masks, idx, dev):
    return {k: v[idx].contiguous().to(dev) for k, v in masks.items()}


def _frozen(m):
    from copy import deepcopy
    e = deepcopy(m)
    for p in e.parameters():
        p.requires_grad = False
    e.train()
    return e


def run(dev, dp, rank, world, steps=2, overlap=True, step="mt", sync_bn=True, graph=False, shard=None, fixed=False, p2p=False, p2p_grads=False):
    """shard: (rank, world) of the data this process sees when it runs WITHOUT a process group (the per-shard reference runs of the
    sync_bn = 0 test); with dp the shard is the rank's.  fixed: the same dropout masks and the step's own consistency law in every
    iteration -- what a captured graph replays (graph=True implies it: one eager warm-up step, the capture, one replay = 2 iterations)."""
    fixed = fixed or graph
    if graph:
        assert steps == 2, "graph mode runs exactly the warm-up step and one replay"
    torch.manual_seed(5)
    reset_dropout_streams()
    if dp is not None:
        dp.sync_bn, dp.overlap = sync_bn, overlap
        if p2p:
            dp.enable_peer_exchange()
        if p2p_grads:
            assert dp.enable_peer_grads(4 * 1024 * 1024), "the peer gradient exchange failed its self-test"
    xl, yl, xu = global_batch()
    srank, sworld = (rank, world) if shard is None else shard
    kl, ku = N_LAB // sworld, N_UNL // sworld
    il = list(range(srank * kl, (srank + 1) * kl))
    iu = list(range(srank * ku, (srank + 1) * ku))
    idx = il + [N_LAB + i for i in iu]          # this rank's images in the order of ITS batch [labelled shard ; unlabelled shard]
    n_all = N_LAB + N_UNL
    if step == "mt":
        m = UNet(1, 4).to(dev)
        m.math = "f32"
        ema = _frozen(m)
        m.train()
        st = MeanTeacherStep(m, ema, opt_args(), dp)
        nets, inputs = [(m, 0), (ema, 1)], (xl[il].to(dev), yl[il].to(dev), xu[iu].to(dev))
        kw = {} if fixed else dict(cons_w=0.05)
        outs = lambda r: r["parts"].cpu()
        saved = lambda: (m.flat_params.cpu(), ema.flat_params.cpu(), dict(ema.named_buffers())["encoder.down2.maxpool_conv.1.conv_conv.1.running_var"].cpu())
    elif step == "cps":
        m1, m2 = UNet(1, 4).to(dev), UNet(1, 4).to(dev)
        m1.math = m2.math = "f32"
        m1.train(), m2.train()
        a = opt_args()
        a.model1, a.model2 = opt_args(), opt_args()
        st = CPSStep(m1, m2, a, dp)
        nets, inputs = [(m1, 0), (m2, 1)], (xl[il].to(dev), yl[il].to(dev), xu[iu].to(dev))
        kw = {} if fixed else dict(cons_w=0.05)
        outs = lambda r: torch.cat([r["parts1"].cpu(), r["parts2"].cpu()])
        saved = lambda: (m1.flat_params.cpu(), m2.flat_params.cpu(), dict(m2.named_buffers())["encoder.down2.maxpool_conv.1.conv_conv.1.running_var"].cpu())
    else:
        m1, m2 = UNet_Plus(1, 4).to(dev), UNet_Plus(1, 4).to(dev)
        m1.math = m2.math = "f32"
        ema = _frozen(m2)
        m1.train(), m2.train()
        a = opt_args(batch_size=kl, unlabel_batch_size=ku)
        a.model1, a.model2 = opt_args(weight_decay=5e-4), opt_args(weight_decay=5e-4)
        st = HPFGStep(m1, m2, ema, a, dp)
        xl1, yl1 = synth_batch(13, N_LAB, SIZE, SIZE, 1, 4, 8)
        cm = st.make_cutmix_mask(N_UNL, (SIZE, SIZE), rng=np.random.RandomState(3))          # one mask per GLOBAL unlabelled image
        rep = ku // kl
        inputs = (xl[il].to(dev), yl[il].to(dev), xl1[il].repeat(rep, 1, 1, 1).to(dev), yl1[il].repeat(rep, 1, 1).to(dev), xu[iu].to(dev), cm[iu].to(dev))
        nets, kw = [(m1, 0), (m2, 1), (ema, 2)], {}
        outs = lambda r: torch.cat([r["parts1"].cpu(), r["parts2"].cpu(), r["contrast"].reshape(1).cpu()])
        saved = lambda: (m1.flat_params.cpu(), m2.flat_params.cpu(), ema.flat_params.cpu())
    first = 1
    if step == "hpfg":
        first = 1000          # past the `cur_itrs < 1000` gate of the consistency term (main.py:186-188)
    losses = []
    if graph:
        for net, who in nets:
            net.external_dropout_masks = take(image_masks(first, who, n_all), idx, dev)
        runner = GraphedStep(st, list(inputs), warmup=1, alias_inputs=True)          # (the warm-up is a real iteration: host_scalars(1))
        losses.append(outs(runner.step(list(inputs), 2)).clone())
    else:
        for k in range(first, first + steps):
            for net, who in nets:
                net.external_dropout_masks = take(image_masks(first if fixed else k, who, n_all), idx, dev)
            kk = (k - first + 1) if fixed else k          # the graphed run counts its iterations 1, 2
            losses.append(outs(st.step(*inputs, kk, **kw)).clone())
    torch.cuda.synchronize()
    if dp is not None:
        dp.check_peer_errors()
    return (torch.stack(losses),) + tuple(saved())


ALLREDUCE_SIZES = (1, 3, 4, 7, 1021, 4096, 65537, 1000003)
ALTERNATING = (630001, 1000003, 4099, 1000003, 630001, 300007) * 4      # (decoder bucket, encoder bucket, ... of different models: ADVICE r3)


def allreduce_case(rank, n):
    g = torch.Generator().manual_seed(100 * n + rank)
    return torch.randn(n, generator=g)


def run_allreduce(dev, dp):
    """The peer-window all-reduce alone: ragged sizes (tails, slices shorter than a vector, empty slices), several uses of the same window."""
    assert dp.enable_peer_grads(1000003)
    outs = []
    for rep in range(2):
        for n in ALLREDUCE_SIZES:
            t = allreduce_case(dp.rank, n).to(dev)
            outs.append(dp.peer_allreduce_sum(t).cpu())
    # consecutive calls of DIFFERENT sizes with the ranks alternately held back on the device (a spin kernel in front of the call): the
    # window layout must not depend on the size -- a rank that has finished call k pushes call k + 1 while its peer still reads call k
    for k, n in enumerate(ALTERNATING):
        if k % dp.world_size == dp.rank:
            torch.cuda._sleep(400_000)
        t = allreduce_case(dp.rank, n).to(dev)
        outs.append(dp.peer_allreduce_sum(t).cpu() if k % 5 == 4 else dp.peer_allreduce_sum(t))
    outs = [o.cpu() for o in outs]
    torch.cuda.synchronize()
    dp.check_peer_errors()
    return tuple(outs)


def main():
    out = sys.argv[1]
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dp = parallel.init_from_env(dev, backend="gloo")
    if os.environ.get("HPFG_TEST_STEP") == "allreduce":
        try:
            torch.save(run_allreduce(dev, dp), f"{out}.rank{dp.rank}")
        finally:
            dp.shutdown()
        return
    try:
        res = run(dev, dp, dp.rank, dp.world_size, steps=int(os.environ.get("HPFG_TEST_STEPS", "2")), overlap=os.environ.get("HPFG_TEST_OVERLAP", "1") == "1",
                  step=os.environ.get("HPFG_TEST_STEP", "mt"), sync_bn=os.environ.get("HPFG_TEST_SYNC_BN", "1") == "1",
                  graph=os.environ.get("HPFG_TEST_GRAPH", "0") == "1", fixed=os.environ.get("HPFG_TEST_FIXED", "0") == "1",
                  p2p=os.environ.get("HPFG_TEST_P2P", "0") == "1", p2p_grads=os.environ.get("HPFG_TEST_P2P_GRADS", "0") == "1")
        torch.save(res, f"{out}.rank{dp.rank}")
    finally:
        dp.shutdown()


if __name__ == "__main__":
    main()
