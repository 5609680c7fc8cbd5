"""GPU, one rank: the data-parallel code path (BN partial sums -> reduce -> RCCL all-reduce -> finalize; loss-sum and flat-gradient
all-reduce) run on a 1-rank `nccl` process group must reproduce the single-GPU path exactly, eagerly and captured into a hipGraph."""
import os
from copy import deepcopy

import pytest
import torch
import torch.distributed as dist

from hpfg_amd import parallel
from hpfg_amd.datasets.synthetic import synth_batch
from hpfg_amd.model import UNet, reset_dropout_streams
from hpfg_amd.train import GraphedStep, MeanTeacherStep
from hpfg_amd.utils import AttrDict
from tests.helpers import maxerr

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _args():
    return AttrDict(opt="sgd", lr=0.01, momentum=0.9, weight_decay=1e-4, sched="medical", total_itrs=30000, step_size=200, warmup_epochs=0,
                    warmup_lr=1e-4, min_lr=1e-6, consistency=0.1, consistency_rampup=200.0, ema_decay=0.99)


def _run(dp, graphed=False, steps=3):
    torch.manual_seed(7)
    reset_dropout_streams()
    m = UNet(1, 4).to(DEV)
    ema = deepcopy(m)
    for p in ema.parameters():
        p.requires_grad = False
    m.train()
    ema.train()
    st = MeanTeacherStep(m, ema, _args(), dp)
    xl, yl = synth_batch(1, 2, 64, 64, 1, 4, 8)
    xu, _ = synth_batch(2, 2, 64, 64, 1, 4, 8)
    xl, yl, xu = xl.to(DEV), yl.to(DEV), xu.to(DEV)
    if graphed:
        g = GraphedStep(st, [xl, yl, xu], warmup=1)
        outs = [float(g.step([xl, yl, xu], k, cons_w=0.05)["loss"]) for k in range(2, steps + 2)]
        return outs, m.flat_params.clone()
    outs = [float(st.step(xl, yl, xu, k, cons_w=0.05)["loss"]) for k in range(1, steps + 1)]
    return outs, m.flat_params.clone()


@pytest.fixture(scope="module")
def dp():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    ctx = parallel.init_from_env(DEV)
    ctx.force_sync = True
    yield ctx
    ctx.shutdown()


def test_sync_path_equals_local_path(dp):
    l0, p0 = _run(None)
    l1, p1 = _run(dp)
    assert max(abs(a - b) for a, b in zip(l0, l1)) < 1e-5, (l0, l1)
    assert maxerr(p0.cpu(), p1.cpu()) < 1e-5


def test_local_bn_mode_is_the_single_gpu_path_plus_one_gradient_allreduce(dp):
    """sync_bn=False (DDP semantics, what bench.py runs for N > 1): per-rank BatchNorm and loss, only the flat gradient crosses
    ranks and the SGD kernel averages it -> on one rank the trajectory is bit-identical to the path without a process group."""
    l0, p0 = _run(None)
    dp.sync_bn = False
    try:
        l1, p1 = _run(dp)
    finally:
        dp.sync_bn = True
    assert l0 == l1 and torch.equal(p0, p1)


def test_local_bn_mode_split_graph(dp):
    """What bench.py runs for N > 1: forward+backward and the update captured as two hipGraphs around an eager RCCL all-reduce."""
    dp.sync_bn = False
    try:
        l_eager, p_eager = _run(dp, steps=3)
        torch.manual_seed(7)
        m = UNet(1, 4).to(DEV)
        ema = deepcopy(m)
        for p in ema.parameters():
            p.requires_grad = False
        m.train()
        ema.train()
        st = MeanTeacherStep(m, ema, _args(), dp)
        xl, yl = synth_batch(1, 2, 64, 64, 1, 4, 8)
        xu, _ = synth_batch(2, 2, 64, 64, 1, 4, 8)
        xl, yl, xu = xl.to(DEV), yl.to(DEV), xu.to(DEV)
        g = GraphedStep(st, [xl, yl, xu], warmup=1, alias_inputs=True)
        assert g.split
        losses = [float(g.step([xl, yl, xu], k, cons_w=0.05)["loss"]) for k in range(2, 5)]
        assert all(x == x and abs(x) < 10 for x in losses)
    finally:
        dp.sync_bn = True


def test_bucketed_overlap_eager_and_graph_chain(dp):
    """Gradient buckets all-reduced from inside backward on a side stream (decoder bucket while the encoder half computes): the eager
    form and the chain of hipGraphs bench.py replays for N > 1 give the trajectory of the path without a process group, bit for bit
    (one rank: the SUM is the identity, so any difference is a missing dependency between the streams / graphs)."""
    l0, p0 = _run(None, steps=4)
    dp.sync_bn = False
    try:
        assert dp.overlap
        l1, p1 = _run(dp, steps=4)
        assert l0 == l1 and torch.equal(p0, p1)
        torch.manual_seed(7)
        m = UNet(1, 4).to(DEV)
        ema = deepcopy(m)
        for p in ema.parameters():
            p.requires_grad = False
        m.train()
        ema.train()
        st = MeanTeacherStep(m, ema, _args(), dp)
        xl, yl = synth_batch(1, 2, 64, 64, 1, 4, 8)
        xu, _ = synth_batch(2, 2, 64, 64, 1, 4, 8)
        xl, yl, xu = xl.to(DEV), yl.to(DEV), xu.to(DEV)
        g = GraphedStep(st, [xl, yl, xu], warmup=1, alias_inputs=True)
        assert g.split and len(g.graphs) == 3 and len(g.bucket_after) == 2
        assert g.bucket_after[0].numel() + g.bucket_after[1].numel() == m.flat_grads.numel()
        losses = [float(g.step([xl, yl, xu], k, cons_w=0.05)["loss"]) for k in range(2, 5)]
        assert all(x == x and abs(x) < 10 for x in losses)
    finally:
        dp.sync_bn = True


def test_sync_path_refuses_graph_capture(dp):
    """Collectives never sit inside a captured region: with all-reduced BatchNorm statistics the step runs eager, and asking for a hipGraph
    of it is refused with a message (RCCL's watchdog thread polls its events while a capture is open; when that race was lost the HIP runtime
    aborted the process -- the mode that captured them is gone)."""
    dp.sync_bn = True
    torch.manual_seed(7)
    m = UNet(1, 4).to(DEV)
    ema = deepcopy(m)
    for p in ema.parameters():
        p.requires_grad = False
    m.train()
    ema.train()
    st = MeanTeacherStep(m, ema, _args(), dp)
    xl, yl = synth_batch(1, 2, 64, 64, 1, 4, 8)
    xu, _ = synth_batch(2, 2, 64, 64, 1, 4, 8)
    with pytest.raises(RuntimeError, match="collectives"):
        GraphedStep(st, [xl.to(DEV), yl.to(DEV), xu.to(DEV)], warmup=1)


def test_ctct_step_on_the_data_parallel_path(dp):
    """The cross-teaching step with a model that has no HIP engine of its own (SegFormer: gradients exchanged as one flattened buffer,
    FusedAdamW) on a 1-rank group: same losses and weights as without the group, in both BatchNorm modes."""
    from hpfg_amd.model import SegFormer
    from hpfg_amd.train import CTCTStep
    from oracle import segformer_ref as S

    def run(ctx):
        torch.manual_seed(9)
        reset_dropout_streams()
        m1, m2 = UNet(1, 4).to(DEV), SegFormer(image_size=[64, 64], in_channels=1, num_classes=4).to(DEV)
        m1.train()
        m2.train()
        opt = dict(opt="sgd", lr=0.01, momentum=0.9, weight_decay=5e-4, sched="medical", total_itrs=30000, step_size=1500, warmup_epochs=1, warmup_lr=1e-4,
                   min_lr=1e-6)
        a = AttrDict(dict(model1=AttrDict(opt), model2=AttrDict(dict(opt, opt="adamW", lr=0.0008, weight_decay=0.05)), consistency=0.1,
                          consistency_rampup=200.0))
        st = CTCTStep(m1, m2, a, ctx)
        xl, yl = synth_batch(1, 2, 64, 64, 1, 4, 8)
        xu, _ = synth_batch(2, 2, 64, 64, 1, 4, 8)
        losses = []
        for k in range(1, 4):
            torch.manual_seed(100 + k)
            m2.external_draws = S.draw_randomness(4)
            losses.append(float(st.step(xl.to(DEV), yl.to(DEV), xu.to(DEV), k, cons_w=0.05)["loss"]))
        return losses, m2.flat_params.clone(), m1.flat_params.clone()

    l0, s0, u0 = run(None)
    for sync_bn in (True, False):
        dp.sync_bn = sync_bn
        l1, s1, u1 = run(dp)
        assert max(abs(a - b) for a, b in zip(l0, l1)) < 1e-5, (sync_bn, l0, l1)
        assert maxerr(s0, s1) < 1e-5 and maxerr(u0, u1) < 1e-5, sync_bn
    dp.sync_bn = True


def test_captured_ctct_update_consumes_the_reduced_gradients(dp):
    """Round-4 advisor finding: with host-launched collectives the step replays as [forward + backward] | eager exchange | [update]; the captured
    update must consume the flat gradient buffer the exchange REDUCED, not re-pack the local per-parameter gradients over it.  On one rank a
    plain SUM is the identity and cannot tell the two apart, so the all-reduce is made to ADD a constant to every element (a scale would not
    do: AdamW's update m / sqrt(v) is invariant to it): the captured chain must then follow the eager run of the same patched context -- and
    differ from the unpatched one."""
    from hpfg_amd.model import SegFormer
    from hpfg_amd.train import CTCTStep

    def run(graphed, shift):
        torch.manual_seed(9)
        reset_dropout_streams()
        m1, m2 = UNet(1, 4).to(DEV), SegFormer(image_size=[64, 64], in_channels=1, num_classes=4).to(DEV)
        m1.train()
        m2.eval()          # no stochastic depth / dropout draws in the SegFormer: the eager and the replayed run must be comparable element by element
        opt = dict(opt="sgd", lr=0.01, momentum=0.9, weight_decay=5e-4, sched="medical", total_itrs=30000, step_size=1500, warmup_epochs=1, warmup_lr=1e-4,
                   min_lr=1e-6)
        a = AttrDict(dict(model1=AttrDict(opt), model2=AttrDict(dict(opt, opt="adamW", lr=0.0008, weight_decay=0.05)), consistency=0.1,
                          consistency_rampup=200.0))
        real = dp.allreduce_sum
        dp.allreduce_sum = lambda t: real(t).add_(shift)
        try:
            st = CTCTStep(m1, m2, a, dp)
            assert st.optimizer2.external_gather
            xl, yl = synth_batch(1, 2, 64, 64, 1, 4, 8)
            xu, _ = synth_batch(2, 2, 64, 64, 1, 4, 8)
            inputs = [xl.to(DEV), yl.to(DEV), xu.to(DEV)]
            g = None
            for k in range(1, 5):
                if graphed and k >= 2:
                    if g is None:
                        g = GraphedStep(st, inputs, warmup=0, alias_inputs=True)
                        assert g.split
                    g.step(inputs, k, cons_w=0.05)
                else:
                    st.step(*inputs, k, cons_w=0.05)
            torch.cuda.synchronize()
            return m2.flat_params.detach().clone(), m1.flat_params.detach().clone()
        finally:
            dp.allreduce_sum = real

    dp.sync_bn = False
    prev_overlap, dp.overlap = dp.overlap, False          # (every exchange through dp.allreduce_sum, the patched call: the bucketed path calls RCCL directly)
    try:
        s_eager, u_eager = run(False, 1e-3)
        s_graph, u_graph = run(True, 1e-3)
        s_plain, _ = run(False, 0.0)
    finally:
        dp.sync_bn, dp.overlap = True, prev_overlap
    assert maxerr(s_eager, s_graph) < 1e-6 and maxerr(u_eager, u_graph) < 1e-6, (maxerr(s_eager, s_graph), maxerr(u_eager, u_graph))
    assert maxerr(s_eager, s_plain) > 1e-5          # the control: the shifted exchange does change the SegFormer's trajectory
