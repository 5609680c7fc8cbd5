"""GPU: run-time properties of the step machinery that the trajectory tests cannot see (they synchronise after every step and replay
the reference's dropout masks): per-network dropout streams, and per-step host scalars under a host that runs ahead of the GPU."""
from copy import deepcopy

import pytest
import torch

from hpfg_amd import engine as E
from hpfg_amd.datasets.synthetic import synth_batch
from hpfg_amd.model import UNet, reset_dropout_streams
from hpfg_amd.train import GraphedStep, MeanTeacherStep
from hpfg_amd.utils import AttrDict
from tests.helpers import maxerr

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _args():
    return AttrDict(opt="sgd", lr=0.01, momentum=0.9, weight_decay=1e-4, sched="medical", total_itrs=30, step_size=200, warmup_epochs=0,
                    warmup_lr=1e-4, min_lr=1e-6, consistency=0.1, consistency_rampup=200.0, ema_decay=0.99)


def _pair(seed=3):
    torch.manual_seed(seed)
    reset_dropout_streams()
    m = UNet(1, 4).to(DEV)
    ema = deepcopy(m)
    for p in ema.parameters():
        p.requires_grad = False
    m.train()
    ema.train()
    return m, ema


def test_student_and_teacher_draw_independent_dropout_masks():
    """The reference's two networks call nn.Dropout separately on the same input (2017_03_NIPS_Mean-Teacher_ACDC.py:95-101): the masks are
    the perturbation the consistency loss sees.  Every network instance (copies included) owns a dropout stream here too."""
    m, ema = _pair()
    m2 = UNet(1, 4).to(DEV)
    assert len({m.dropout_seed, ema.dropout_seed, m2.dropout_seed}) == 3
    x, _ = synth_batch(5, 2, 64, 64, 1, 4, 8)
    x = x.to(DEV)
    with torch.no_grad():
        m(x)
        ema(x)
    name = E.enc_prefix(0) + ".0"          # dropout p = 0.05 behind the first conv: nothing random upstream of it
    acts = []
    for net in (m, ema):
        eng = next(iter(net._engines.values()))[0]
        eng.dropout_on = True
        acts.append(eng.materialize(name))
    a, b = acts
    dropped_a, dropped_b = a == 0, b == 0
    both = (~dropped_a) & (~dropped_b)
    assert maxerr(a[both].cpu(), b[both].cpu()) < 1e-6           # same weights, same input: equal wherever both keep
    frac = float((dropped_a ^ dropped_b).float().mean())
    assert 0.5 < frac / (2 * 0.05 * 0.95) < 1.5, frac            # independent Bernoulli(0.05) masks disagree on 2p(1-p) of the elements
    # and the stream is reproducible: the same construction order after the same seed gives the same seeds
    m3, ema3 = _pair()
    assert (m3.dropout_seed, ema3.dropout_seed) == (m.dropout_seed, ema.dropout_seed)


def test_graph_replays_without_host_sync_use_their_own_scalars():
    """Six Mean-Teacher steps replayed from one hipGraph with NO host synchronisation in between (the host finishes queueing all of them
    while the GPU is still in the first) must equal six eager steps that synchronise every time: learning rate, EMA alpha
    (1 - 1/(step+1): 0.5, 0.667, 0.75, ...) and the consistency weight of step k reach the kernels of step k, not a later step's."""
    xl, yl = synth_batch(1, 4, 96, 96, 1, 4, 8)
    xu, _ = synth_batch(2, 4, 96, 96, 1, 4, 8)
    xl, yl, xu = xl.to(DEV), yl.to(DEV), xu.to(DEV)
    K = 8
    weights = [0.01 * k for k in range(1, K + 1)]

    m, ema = _pair()
    st = MeanTeacherStep(m, ema, _args())
    float(st.step(xl, yl, xu, 1)["loss"])                                     # == the graph's eager warm-up step
    for k in range(2, K + 1):
        float(st.step(xl, yl, xu, k, cons_w=weights[k - 1])["loss"])          # synchronises
    ref_p, ref_e = m.flat_params.clone(), ema.flat_params.clone()

    m, ema = _pair()
    st = MeanTeacherStep(m, ema, _args())
    g = GraphedStep(st, [xl, yl, xu], warmup=1, alias_inputs=True)
    for k in range(2, K + 1):
        g.step([xl, yl, xu], k, cons_w=weights[k - 1])                        # no synchronisation
    torch.cuda.synchronize()
    assert maxerr(m.flat_params.cpu(), ref_p.cpu()) < 1e-6
    assert maxerr(ema.flat_params.cpu(), ref_e.cpu()) < 1e-6
