"""GPU: eager steps on a step object AFTER it has been captured into a hipGraph and replayed.

Round 3's bench.py died with SIGSEGV inside the HIP runtime in 2 of 5 runs exactly there: eager Mean-Teacher steps (with HIP timing events
recorded around one launch) on the step object whose graph had just been replayed.  The eager probe was deleted; the sequence itself stays
reachable -- the driver loops fall back to an eager step for a batch whose shape differs from the captured one -- so it is pinned here:
capture, replay, then run the SAME step object eagerly, and compare with a run that never captured, bit for bit (dropout seeds included:
UNet.bump_graph_seed keeps the host's seed counter in step with the replays).  What the two forms share and why it is safe is listed in
DESIGN.md section 9 (round 4, item 2)."""
import pytest
import torch

from hpfg_amd.datasets.synthetic import synth_batch
from hpfg_amd.model import UNet, reset_dropout_streams
from hpfg_amd.train import GraphedStep, MeanTeacherStep, batch_pair
from tests.dp_rank_worker import _frozen, opt_args

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _run(graph_iters):
    """Iterations 1 .. 6 of one Mean-Teacher run; those in `graph_iters` are hipGraph replays, the others eager steps of the same object."""
    reset_dropout_streams()
    torch.manual_seed(7)
    m = UNet(1, 4).to(DEV)
    ema = _frozen(m)
    m.train()
    st = MeanTeacherStep(m, ema, opt_args(), None)
    xl, yl = synth_batch(3, 2, 64, 64, 1, 4, 8)
    xu, _ = synth_batch(4, 2, 64, 64, 1, 4, 8)
    xl, xu = batch_pair(xl.to(DEV), xu.to(DEV))
    inputs = [xl, yl.to(DEV), xu]
    losses, runner = [], None
    for it in range(1, 7):
        if it in graph_iters:
            if runner is None:
                runner = GraphedStep(st, inputs, warmup=0, alias_inputs=True)      # captures only
            r = runner.step(inputs, it)
        else:
            r = st.step(*inputs, it)
        losses.append(r["parts"].detach().clone())
    torch.cuda.synchronize()
    return torch.stack(losses).cpu(), m.flat_params.detach().cpu().clone(), ema.flat_params.detach().cpu().clone()


def test_eager_steps_after_capture_and_replay():
    ref = _run(graph_iters=())                    # never captured
    got = _run(graph_iters=(2, 3, 5))             # eager, replay, replay, EAGER, replay, EAGER on one step object
    assert torch.equal(got[0], ref[0]), (got[0][:, 0], ref[0][:, 0])
    assert torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2])
