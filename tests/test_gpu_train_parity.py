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


# ---- long horizon: the precision credit of the default math mode must not rest on 25 iterations -------------------
LONG_MT, LONG_HPFG = 500, 200


@pytest.fixture(scope="module")
def oracle_long():
    """500 Mean-Teacher iterations of the fp32 CPU oracle (about 30 s on the GPU box's 16 threads): loss trace + held-out predictions."""
    global ITERS
    torch.set_num_threads(max(1, min(16, len(__import__("os").sched_getaffinity(0)))))
    keep, ITERS = ITERS, LONG_MT
    try:
        losses, st, ema_st = _train_oracle("f32")
    finally:
        ITERS = keep
    xe, ye = synth_batch(777, 16, HW, HW)
    with torch.no_grad():
        return {"losses": losses, "xe": xe, "ye": ye, "student": unet_ref.unet_forward(st, xe, False), "teacher": unet_ref.unet_forward(ema_st, xe, False)}


@pytest.mark.parametrize("math", ["bf16x3", "f32"])
def test_mean_teacher_500_iterations_loss_trace_and_dice(oracle_long, math):
    """BASELINE.json: "mean Dice vs CPU ref ... within 1e-3".  500 iterations from the same weights, batches and dropout masks: the WHOLE loss
    trace stays within a flat 1e-3 of the fp32 oracle's (measured: 1.5e-4 in bf16x3) and the held-out mean foreground Dice of student and EMA
    teacher within 1e-3 (measured: 2e-5; the loss falls 0.96 -> 0.009, Dice 0.998).  The logits themselves are NOT compared at this horizon:
    training is a chaotic map -- the fp64-ACCUMULATING oracle (summation order only) ends 0.2 away from the fp32 oracle on them, as the device
    does (tools/long_parity.py prints that control: profiles/r04_long_parity.txt) -- which is exactly why the metric is Dice."""
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
    got = []
    for k in range(1, LONG_MT + 1):
        xl, yl, xu, ms, mt = _batch(k)
        m.external_dropout_masks, ema.external_dropout_masks = _device_masks(ms), _device_masks(mt)
        got.append(step.step(xl.to(DEV), yl.to(DEV), xu.to(DEV), k, cons_w=CONS_W)["loss"])
    got, ref = torch.stack(got).cpu().numpy(), oracle_long["losses"]
    assert ref[-25:].mean() < 0.05 < ref[:5].mean()                  # it trains to convergence
    assert np.abs(ref - got).max() < 1e-3, (np.abs(ref - got).max(), int(np.abs(ref - got).argmax()))
    m.eval()
    ema.eval()
    m.external_dropout_masks = ema.external_dropout_masks = None
    xe, ye = oracle_long["xe"], oracle_long["ye"]
    for net, who in ((m, "student"), (ema, "teacher")):
        with torch.no_grad():
            o = net(xe.to(DEV)).cpu()
        d_got = losses_ref.mean_foreground_dice(o.argmax(1).numpy(), ye.numpy(), 4)
        d_ref = losses_ref.mean_foreground_dice(oracle_long[who].argmax(1).numpy(), ye.numpy(), 4)
        assert d_ref > 0.9 and abs(d_got - d_ref) < 1e-3, (who, d_got, d_ref)


def test_hpfg_200_iterations_loss_trace_and_dice():
    """The same for the HPFG step (main.py:125-212: two U-Net+ students, EMA teacher, CutMix pseudo-labels, Dense_Loss, backbone EMA) in the
    default math mode: 200 iterations past the `cur_itrs >= 1000` gate of the consistency term, loss trace within a flat 1e-3, held-out Dice
    of both students and the teacher within 1e-3 of the fp32 oracle's."""
    import os
    from copy import deepcopy

    from hpfg_amd.model import UNet_Plus
    from hpfg_amd.train import HPFGStep
    from hpfg_amd.utils import BoxMaskGenerator
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    nl = nu = 4
    first = 1000
    gen = BoxMaskGenerator(prop_range=(0.25, 0.5), n_boxes=4, random_aspect_ratio=True, prop_by_area=True, within_bounds=True, invert=True)

    def batches():
        rng = np.random.RandomState(11)
        for k in range(LONG_HPFG):
            xl, yl = synth_batch(3000 + k, nl, HW, HW)
            xl1, yl1 = synth_batch(4000 + k, nl, HW, HW)
            xu, _ = synth_batch(5000 + k, nu, HW, HW)
            cm = torch.tensor(gen.generate_params(nu, (HW, HW), rng=rng), dtype=torch.float)
            torch.manual_seed(6000 + k)
            yield first + k, xl, yl, xl1, yl1, xu, cm, [unet_ref.draw_dropout_masks(nl + nu, HW, HW) for _ in range(3)]

    # fp32 oracle
    torch.manual_seed(1)
    sa, sb = unet_ref.init_state(None, 1, 4, True), unet_ref.init_state(None, 1, 4, True)
    se, ba, bb, ref = unet_ref.clone_state(sb), {}, {}, []
    for cur, xl, yl, xl1, yl1, xu, cm, (ma, mb, mt) in batches():
        lr = laws_ref.medical_lr(cur, 0.01, 30000)
        ref.append(steps_ref.hpfg_step(sa, sb, se, ba, bb, xl, yl.long(), xl1, yl1.long(), xu, cm, cur, lr, lr, 0.1, 200.0, 0.99, 0.9, 5e-4, ma, mb, mt)["loss"])
    # HIP path
    torch.manual_seed(1)
    m1, m2 = UNet_Plus(1, 4).to(DEV), UNet_Plus(1, 4).to(DEV)
    ema = deepcopy(m2)
    for p in ema.parameters():
        p.requires_grad = False
    m1.train(), m2.train()
    a = AttrDict(dict(ARGS, batch_size=nl, unlabel_batch_size=nu))
    a.model1, a.model2 = AttrDict(dict(ARGS, weight_decay=5e-4)), AttrDict(dict(ARGS, weight_decay=5e-4))
    st = HPFGStep(m1, m2, ema, a)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(first - 1):
            st.lr_scheduler1.step()
            st.lr_scheduler2.step()
    got = []
    for cur, xl, yl, xl1, yl1, xu, cm, (ma, mb, mt) in batches():
        m1.external_dropout_masks, m2.external_dropout_masks, ema.external_dropout_masks = _device_masks(ma), _device_masks(mb), _device_masks(mt)
        got.append(st.step(xl.to(DEV), yl.to(DEV), xl1.to(DEV), yl1.to(DEV), xu.to(DEV), cm.to(DEV), cur)["loss"])
    got, ref = torch.stack(got).cpu().numpy(), np.array(ref)
    assert ref[-10:].mean() < ref[:5].mean() - 0.3
    assert np.abs(ref - got).max() < 1e-3, (np.abs(ref - got).max(), int(np.abs(ref - got).argmax()))
    xe, ye = synth_batch(778, 16, HW, HW)
    for net, state, who in ((m1, sa, "model1"), (m2, sb, "model2"), (ema, se, "teacher")):
        net.eval()
        net.external_dropout_masks = None
        with torch.no_grad():
            o = net(xe.to(DEV))
            o = (o[0] if isinstance(o, (tuple, list)) else o).cpu()
            r = unet_ref.unet_forward(state, xe, False, plus=True)
            r = r[0] if isinstance(r, (tuple, list)) else r
        d_got = losses_ref.mean_foreground_dice(o.argmax(1).numpy(), ye.numpy(), 4)
        d_ref = losses_ref.mean_foreground_dice(r.argmax(1).numpy(), ye.numpy(), 4)
        assert abs(d_got - d_ref) < 1e-3, (who, d_got, d_ref)
