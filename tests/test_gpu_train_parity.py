"""Long-horizon parity: the HIP Mean-Teacher step trained side by side with the CPU oracle for 25 iterations on fresh synthetic
batches (same initial weights, same dropout masks), then both evaluated on held-out slices -- BASELINE.json's "mean Dice vs CPU
ref" half of the metric: losses, eval logits and mean foreground Dice must agree within 1e-3 (student and EMA teacher)."""
import numpy as np
import pytest
import torch

from hpfg_amd import engine as E
from hpfg_amd.model import UNet
from hpfg_amd.train import MeanTeacherStep
from hpfg_amd.utils import AttrDict
from oracle import bf16x3_ref, laws_ref, losses_ref, steps_ref, unet_ref
from oracle.make_golden import synth_batch
from tests.helpers import maxerr

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
ITERS, HW, NL, NU = 25, 64, 4, 4


def _device_masks(masks):
    """oracle dropout masks (float [n,C,h,w] per encoder level) -> the engine's replay form (uint8 NHWC per conv name)."""
    return {E.enc_prefix(lvl) + ".0": m.to(torch.uint8).permute(0, 2, 3, 1).contiguous().to(DEV) for lvl, m in enumerate(masks)}


ARGS = dict(opt="sgd", lr=0.01, momentum=0.9, weight_decay=1e-4, sched="medical", total_itrs=30000, step_size=200, warmup_epochs=0,
            warmup_lr=1e-4, min_lr=1e-6, consistency=0.1, consistency_rampup=200.0, ema_decay=0.99)
CONS_W = 0.1 * laws_ref.sigmoid_rampup(40, 200.0)


def _batch(k):
    xl, yl = synth_batch(1000 + k, NL, HW, HW)
    xu, _ = synth_batch(2000 + k, NU, HW, HW)
    torch.manual_seed(5000 + k)
    return xl, yl, xu, unet_ref.draw_dropout_masks(NL + NU, HW, HW), unet_ref.draw_dropout_masks(NL + NU, HW, HW)


def _train_oracle(mode):
    """ITERS Mean-Teacher iterations of the CPU oracle in one arithmetic mode of oracle/bf16x3_ref.py; returns (losses, student, teacher)."""
    st = unet_ref.init_state(1337, 1, 4)
    ema_st, bufs, losses = unet_ref.clone_state(st), {}, []
    with bf16x3_ref.math_mode(mode):
        for k in range(1, ITERS + 1):
            xl, yl, xu, ms, mt = _batch(k)
            lr, al = laws_ref.medical_lr(k, 0.01, 30000), laws_ref.ema_alpha(k, 0.99)
            losses.append(steps_ref.mean_teacher_step(st, ema_st, bufs, xl, yl.long(), xu, lr, CONS_W, al, 0.9, 1e-4, ms, mt)["loss"])
    return np.array(losses), st, ema_st


@pytest.fixture(scope="module")
def oracle_run():
    """The CPU oracle trained for ITERS iterations (the reference run), plus the two CONTROL runs that bound the eval logits:
      * "f64acc": the same fp32 products accumulated in fp64 -- the oracle up to summation-order noise, which is all that separates a correct
        exact-fp32 kernel from it; 25 SGD steps on 8 images of 64 x 64 amplify that noise far beyond 1e-3 on the logits;
      * "bf16x3": the oracle with the device's split-bf16 products (its error model) -- what the default math mode is allowed to differ by.
    A math mode's eval logits must stay within 1e-3 + 2x its control's distance from the reference run."""
    losses, st, ema_st = _train_oracle("f32")
    xe, ye = synth_batch(777, 8, HW, HW)
    out = {"losses": losses, "xe": xe, "ye": ye}
    with torch.no_grad():
        out["student"], out["teacher"] = unet_ref.unet_forward(st, xe, False), unet_ref.unet_forward(ema_st, xe, False)
    for math, mode in (("f32", "f64acc"), ("bf16x3", "bf16x3")):
        _, a, b = _train_oracle(mode)
        with torch.no_grad():
            out[f"student_ctl_{math}"] = maxerr(unet_ref.unet_forward(a, xe, False), out["student"])
            out[f"teacher_ctl_{math}"] = maxerr(unet_ref.unet_forward(b, xe, False), out["teacher"])
    return out


@pytest.mark.parametrize("math", ["f32", "bf16x3"])
def test_mean_teacher_25_iterations_then_dice(oracle_run, math):
    from copy import deepcopy
    torch.manual_seed(1337)
    m = UNet(1, 4).to(DEV)
    m.math = math
    ema = deepcopy(m)
    for p in ema.parameters():
        p.requires_grad = False
    m.train()
    ema.train()
    step = MeanTeacherStep(m, ema, AttrDict(dict(ARGS)))
    got_loss = []
    for k in range(1, ITERS + 1):
        xl, yl, xu, ms, mt = _batch(k)
        m.external_dropout_masks, ema.external_dropout_masks = _device_masks(ms), _device_masks(mt)
        got_loss.append(float(step.step(xl.to(DEV), yl.to(DEV), xu.to(DEV), k, cons_w=CONS_W)["loss"]))
    ref_loss, got_loss = oracle_run["losses"], np.array(got_loss)
    assert ref_loss[-5:].mean() < ref_loss[:5].mean() - 0.05            # it trains
    assert np.abs(ref_loss - got_loss).max() < 1e-3, np.abs(ref_loss - got_loss)
    # held-out evaluation, eval-mode BatchNorm (running statistics collected during the 25 steps)
    xe, ye = oracle_run["xe"], oracle_run["ye"]
    m.eval()
    ema.eval()
    m.external_dropout_masks = ema.external_dropout_masks = None
    for net, who in ((m, "student"), (ema, "teacher")):
        with torch.no_grad():
            got = net(xe.to(DEV)).cpu()
        ref, ctl = oracle_run[who], oracle_run[f"{who}_ctl_{math}"]
        assert maxerr(got, ref) < 1e-3 + 2.0 * ctl, (who, maxerr(got, ref), ctl)
        d_got = losses_ref.mean_foreground_dice(got.argmax(1).numpy(), ye.numpy(), 4)
        d_ref = losses_ref.mean_foreground_dice(ref.argmax(1).numpy(), ye.numpy(), 4)
        assert abs(d_got - d_ref) < 1e-3, (who, d_got, d_ref)
