"""GPU parity at BASELINE-like sizes against the REFERENCE's own outputs (tests/golden/trace_mt224.npz, trace_cps96.npz,
trace_hpfg224.npz, grads224.npz; written by oracle/make_golden_r3.py from the reference modules): Mean-Teacher 2 + 2 @ 224^2 (3
iterations), CPS 4 + 4 @ 96^2 RGB (2), HPFG 2 + 2 @ 224^2 (2), and one forward + backward of 4 images @ 224^2.

Bounds.  Losses, logits (every 8th pixel + checksums): the flat 1e-3 of BASELINE.json north_star, in BOTH math modes -- at these sizes
BatchNorm averages over >= 784 samples per channel and the trajectories are not the perturbation amplifiers the 32..64-pixel fixtures
are.  Gradients (relative L2 per tensor, against the fp32 oracle): 1e-3 + 2x the spread of a committed control ensemble on the same tensor --
the oracle re-run in the math mode's OWN arithmetic (for split-bf16: the emulated device products of oracle/bf16x3_ref.py, same hi / lo
operands and the same three partial products) under perturbations at that arithmetic's noise level (tests/trace_replay.py::grad_ensemble).
Measured on the CPU alone (no kernel involved): the split-bf16 arithmetic moves the deepest encoder gradients by up to 1.2e-2 and the median
tensor by 2e-3 at 4 x 224^2 -- LeakyReLU / max-pool decisions within 2^-17 of a tie fall the other way -- while logits and losses stay
within 6e-5 / 2e-7: what BASELINE.json bounds (logits, Dice, loss) holds flat, the gradients carry the error model's spread.
"""
import warnings
from copy import deepcopy

import numpy as np
import pytest
import torch

from hpfg_amd import engine as E
from hpfg_amd.model import UNet, UNet_Plus
from hpfg_amd.train import CPSStep, HPFGStep, MeanTeacherStep
from hpfg_amd.utils import Med_Sup_Loss
from tests import trace_replay as R
from tests.test_gpu_steps import _opt_args

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
TOL = 1e-3


def dev_masks(ms):
    """oracle keep-masks (five NCHW float tensors) -> the engine's external mask dict (uint8 NHWC on the device)."""
    return {E.enc_prefix(lvl) + ".0": m.to(torch.uint8).permute(0, 2, 3, 1).contiguous().to(DEV) for lvl, m in enumerate(ms)}


def _spin(scheds, n):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(n):
            for s in scheds:
                s.step()


@pytest.mark.parametrize("math", ["f32", "bf16x3"])
def test_mean_teacher_224_reference_trace(golden_dir, math):
    d = np.load(f"{golden_dir}/trace_mt224.npz")
    xl, yl, xu, masks, first = R.mt224_inputs(d)
    torch.manual_seed(1337)
    m = UNet(1, 4).to(DEV)
    m.math = math
    ema = deepcopy(m)
    for p in ema.parameters():
        p.requires_grad = False
    m.train()
    ema.train()
    st = MeanTeacherStep(m, ema, _opt_args())
    _spin([st.lr_scheduler], first - 1)
    xl, yl, xu = xl.to(DEV), yl.to(DEV), xu.to(DEV)
    rows = []
    for j, (ms, mt) in enumerate(masks):
        assert abs(st.optimizer.param_groups[0]["lr"] - float(d["lrs"][j])) < 1e-12
        m.external_dropout_masks, ema.external_dropout_masks = dev_masks(ms), dev_masks(mt)
        r = st.step(xl, yl, xu, first + j)          # the step's own consistency law (cur_itrs // 150 = 40)
        p = r["parts"].cpu()
        rows.append([float(r["loss"]), 0.5 * float(p[1]) + 0.5 * float(p[2]), float(p[5])])
    assert np.abs(np.array(rows) - d["losses"]).max() < TOL, (rows, d["losses"])
    assert R.sub_err(r["logits"].cpu(), d, "student_logits_last") < TOL
    assert R.sub_err(r["t_logits"].cpu(), d, "teacher_logits_last") < TOL


@pytest.mark.parametrize("math", ["f32", "bf16x3"])
def test_cps_96_reference_trace(golden_dir, math):
    d = np.load(f"{golden_dir}/trace_cps96.npz")
    xl, yl, xu, masks, first = R.cps96_inputs(d)
    torch.manual_seed(1337)
    m1 = UNet(3, 2).to(DEV)
    m2 = UNet(3, 2).to(DEV)
    m1.math = m2.math = math
    m1.train()
    m2.train()
    args = _opt_args()
    args.model1, args.model2 = _opt_args(), _opt_args()
    st = CPSStep(m1, m2, args)
    _spin([st.lr_scheduler1, st.lr_scheduler2], first - 1)
    xl, yl, xu = xl.to(DEV), yl.to(DEV), xu.to(DEV)
    rows = []
    for j, (ma, mb) in enumerate(masks):
        m1.external_dropout_masks, m2.external_dropout_masks = dev_masks(ma), dev_masks(mb)
        r = st.step(xl, yl, xu, first + j)
        rows.append(float(r["loss"]))
    assert np.abs(np.array(rows) - d["losses"][:, 0]).max() < TOL, (rows, d["losses"])
    assert R.sub_err(r["logits1"].cpu(), d, "logits1_last") < TOL
    assert R.sub_err(r["logits2"].cpu(), d, "logits2_last") < TOL


@pytest.mark.parametrize("math", ["f32", "bf16x3"])
def test_hpfg_224_reference_trace(golden_dir, math):
    d = np.load(f"{golden_dir}/trace_hpfg224.npz")
    xl, yl, xl1, yl1, xu, cms, masks = R.hpfg224_inputs(d)
    torch.manual_seed(1)
    m1 = UNet_Plus(1, 4).to(DEV)
    m2 = UNet_Plus(1, 4).to(DEV)
    m1.math = m2.math = math
    ema = deepcopy(m2)
    for p in ema.parameters():
        p.requires_grad = False
    m1.train()
    m2.train()
    nl, nu = xl.shape[0], xu.shape[0]
    args = _opt_args(batch_size=nl, unlabel_batch_size=nu)
    args.model1, args.model2 = _opt_args(weight_decay=5e-4), _opt_args(weight_decay=5e-4)
    st = HPFGStep(m1, m2, ema, args)
    _spin([st.lr_scheduler1, st.lr_scheduler2], int(d["cur_itrs"][0]) - 1)
    rep = nu // nl
    xl1r, yl1r = xl1.repeat(rep, 1, 1, 1).to(DEV), yl1.repeat(rep, 1, 1).to(DEV)
    xl, yl, xu = xl.to(DEV), yl.to(DEV), xu.to(DEV)
    rows = []
    for j, cur in enumerate(d["cur_itrs"]):
        ma, mb, mt = masks[j]
        m1.external_dropout_masks, m2.external_dropout_masks, ema.external_dropout_masks = dev_masks(ma), dev_masks(mb), dev_masks(mt)
        r = st.step(xl, yl, xl1r, yl1r, xu, cms[j].to(DEV), int(cur))
        rows.append([float(r["loss"]), float(r["contrast"]), float(r["parts2"][5])])
    rows, ref = np.array(rows), d["losses"]
    assert np.abs(rows[:, 0] - ref[:, 0]).max() < TOL, (rows, ref)
    assert np.abs(rows[:, 1] - ref[:, 4]).max() < TOL, (rows, ref)
    assert np.abs(rows[:, 2] - ref[:, 5]).max() < 1e-4, (rows, ref)
    assert R.sub_err(r["logits1"].cpu(), d, "logits1_last") < TOL
    assert R.sub_err(r["logits2"].cpu(), d, "logits2_last") < TOL
    assert R.sub_err(r["t_logits"].cpu(), d, "t_logits_last") < TOL


@pytest.mark.parametrize("math", ["f32", "bf16x3"])
def test_gradients_4x224_vs_oracle_and_error_model(golden_dir, math):
    """Every parameter gradient of one train-mode forward + backward (UNet(1,4), 0.5 CE + 0.5 Dice, 4 images @ 224^2, the reference run's
    dropout masks), per tensor, against the fp32 oracle (itself within 1e-5 of the reference: grads224.npz / test_oracle_golden.py).
    Bound per tensor k: 1e-3 + 2 x the spread of a committed control ensemble on that tensor (tests/trace_replay.py::grad_ensemble: the
    oracle in the mode's own arithmetic under perturbations at the mode's noise level, incl. the emulated split-bf16 arithmetic itself)."""
    d = np.load(f"{golden_dir}/grads224.npz")
    x, lab, masks = R.grads224_inputs(d)
    ref = R.replay_grads224(d)["grads"]
    torch.manual_seed(1)
    m = UNet(1, 4).to(DEV)
    m.math = math
    m.train()
    m.external_dropout_masks = dev_masks(masks)
    out = m(x.to(DEV))
    loss = Med_Sup_Loss(4)(out, lab.to(DEV))
    loss.backward()
    assert abs(float(loss) - float(d["loss"])) < 1e-4
    assert R.sub_err(out.detach().cpu(), d, "logits") < (2e-4 if math == "bf16x3" else 5e-5)
    got = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
    live = [k for k in got if float(d[f"g:{k}:norm"]) > 1e-6]          # (biases in front of a train-mode BatchNorm: zero up to rounding noise)
    assert len(live) == len(got) - 18
    from oracle import unet_ref
    nominal, ens = R.grad_ensemble(unet_ref.init_state(1, 1, 4), x, lab, masks, math, k_runs=4)
    ctl = {k: max(R.rel_l2(e[k], ref[k]) for e in ens + [nominal]) for k in live}
    err = {k: R.rel_l2(got[k], ref[k]) for k in live}
    bad = {k: (err[k], ctl[k]) for k in live if not err[k] < 1e-3 + 2.0 * ctl[k]}
    assert not bad, bad
    print(f"{math} gradients, 4 x 224^2 vs fp32 oracle: max {max(err.values()):.2e} median {float(np.median(list(err.values()))):.2e}; "
          f"control ensemble: max {max(ctl.values()):.2e} median {float(np.median(list(ctl.values()))):.2e}")
