"""Oracle replays of the golden step traces (tests/golden/trace_*.npz), shared by the CPU pinning tests and by the GPU tests' CONTROL runs.

A control run repeats a trace on the CPU oracle from conv weights perturbed by a relative 1e-6 (uniform in [-1e-6, 1e-6] per element,
about ten fp32 roundings) and measures how far the trace's final logits move.  The fixtures are tiny (N = 4..8 at 32..64 px: BatchNorm
over as few as 16 values at the bottleneck, followed by 2-3 SGD steps), so they amplify any such perturbation several hundred times:
that measured drift, not a hand-picked constant, is what bounds the split-bf16 ("bf16x3") math mode on them
(tests/test_gpu_steps.py::logit_tol).  Round 3: the control of the bf16x3 mode is the oracle itself running the split-bf16 arithmetic
(oracle/bf16x3_ref.py; `emulation_drift`, `grad_ensemble`), and the tests allow 1e-3 + 2x the control's drift.
"""
from __future__ import annotations

import numpy as np
import torch

from oracle import laws_ref, steps_ref, unet_ref

CONTROL_EPS = 1e-6
CONTROL_SEEDS = (0, 1)


def unpack_masks(d, prefix, n, hw):
    out = []
    for lvl in range(5):
        c, h = unet_ref.WIDTHS[lvl], hw >> lvl
        bits = np.unpackbits(d[f"{prefix}{lvl}"])[: n * c * h * h].reshape(n, c, h, h)
        out.append(torch.from_numpy(bits.astype(np.float32)))
    return out


def perturb(st, seed):
    """Relative perturbation of every conv weight of an oracle state, in place; seed None = nominal run."""
    if seed is None:
        return st
    g = torch.Generator().manual_seed(1000 + seed)
    for n_ in st:
        if n_.endswith(".weight") and st[n_].dim() == 4:
            r = 2 * torch.rand(st[n_].shape, generator=g, dtype=torch.float64) - 1
            st[n_] = (st[n_].double() * (1 + CONTROL_EPS * r)).float()
    return st


def replay_sup(d, seed=None):
    st, bufs = perturb(unet_ref.init_state(1, 1, 4), seed), {}
    table = laws_ref.cosine_table(0.01, 0, 1e-4, 1e-6, 200, 150)
    x, lab = torch.from_numpy(d["x"]), torch.from_numpy(d["labels"]).long()
    losses = [steps_ref.supervised_step(st, bufs, x, lab, laws_ref.cosine_lr(k + 1, table), 0.9, 5e-4, unpack_masks(d, f"it{k}_mask", 4, 32))["loss"]
              for k in range(4)]
    with torch.no_grad():
        fin = unet_ref.unet_forward(st, x, train=False)
    return {"losses": np.array(losses), "final_eval_logits": fin}


def replay_mt(d, seed=None):
    st = perturb(unet_ref.init_state(1337, 1, 4), seed)
    ema, bufs = unet_ref.clone_state(st), {}
    xl, yl, xu = torch.from_numpy(d["xl"]), torch.from_numpy(d["yl"]).long(), torch.from_numpy(d["xu"])
    rows = []
    for k in range(3):
        r = steps_ref.mean_teacher_step(st, ema, bufs, xl, yl, xu, laws_ref.medical_lr(k + 1, 0.01, 30000), float(d["cons_w"]),
                                        laws_ref.ema_alpha(k + 1, 0.99), 0.9, 1e-4, unpack_masks(d, f"it{k}_s", 4, 32), unpack_masks(d, f"it{k}_t", 4, 32))
        rows.append([r["loss"], r["sup"], r["cons"]])
    return {"losses": np.array(rows), "student_logits_last": r["logits"], "teacher_logits_last": r["t_logits"]}


def replay_ict(d, seed=None):
    st = perturb(unet_ref.init_state(1337, 1, 4), seed)
    ema, bufs = unet_ref.clone_state(st), {}
    xl, yl, xu = torch.from_numpy(d["xl"]), torch.from_numpy(d["yl"]).long(), torch.from_numpy(d["xu"])
    rows = []
    for k in range(3):
        r = steps_ref.ict_step(st, ema, bufs, xl, yl, xu, torch.from_numpy(d["mixes"][k]), laws_ref.medical_lr(k + 1, 0.01, 30000), float(d["cons_w"]),
                               laws_ref.ema_alpha(k + 1, 0.99), 0.9, 1e-4, unpack_masks(d, f"it{k}_s", 4, 32), unpack_masks(d, f"it{k}_a", 2, 32),
                               unpack_masks(d, f"it{k}_b", 2, 32))
        rows.append([r["loss"], r["sup"], r["cons"]])
    return {"losses": np.array(rows), "student_logits_last": r["logits"]}


def replay_uamt(d, seed=None):
    st = perturb(unet_ref.init_state(1337, 1, 4), seed)
    ema, bufs = perturb(unet_ref.init_state(None, 1, 4), None if seed is None else seed + 10), {}
    xl, yl, xu = torch.from_numpy(d["xl"]), torch.from_numpy(d["yl"]).long(), torch.from_numpy(d["xu"])
    rows = []
    for k in range(2):
        nz = torch.from_numpy(d["noise"][k])
        mt = [unpack_masks(d, f"it{k}_f1_", 2, 32)] + [unpack_masks(d, f"it{k}_f{j}_", 4, 32) for j in range(2, 6)]
        r = steps_ref.uamt_step(st, ema, bufs, xl, yl, xu, nz[:2], [nz[2 + 4 * i:6 + 4 * i] for i in range(4)], float(d["thresholds"][k]),
                                laws_ref.medical_lr(k + 1, 0.01, 30000), float(d["cons_w"]), laws_ref.ema_alpha(k + 1, 0.99), 0.9, 1e-4,
                                unpack_masks(d, f"it{k}_f0_", 4, 32), mt)
        rows.append([r["loss"], r["sup"], r["cons"]])
    return {"losses": np.array(rows), "uncertainty_last": r["uncertainty"], "student_logits_last": r["logits"]}


def replay_cps(d, seed=None):
    torch.manual_seed(1337)
    sa, sb = unet_ref.init_state(None, 3, 2), unet_ref.init_state(None, 3, 2)
    perturb(sa, seed)
    perturb(sb, None if seed is None else seed + 10)
    ba, bb = {}, {}
    xl, yl, xu = torch.from_numpy(d["xl"]), torch.from_numpy(d["yl"]).long(), torch.from_numpy(d["xu"])
    rows = []
    for k in range(2):
        lr = laws_ref.medical_lr(k + 1, 0.01, 30000)
        r = steps_ref.cps_step(sa, sb, ba, bb, xl, yl, xu, lr, lr, float(d["cons_w"]), 0.9, 1e-4, unpack_masks(d, f"it{k}_a", 4, 48),
                               unpack_masks(d, f"it{k}_b", 4, 48))
        rows.append([r["loss"], r["sup"], r["semi"]])
    return {"losses": np.array(rows), "logits1_last": r["logits1"], "logits2_last": r["logits2"]}


def replay_hpfg(d, seed=None, stepped_lr=False):
    """trace_hpfg.npz (constant lr 0.01, batch 2+2) and trace_hpfg2.npz (stepped Medical_LR, batch 2+6 with the labelled repeat)."""
    torch.manual_seed(1)
    sa, sb = unet_ref.init_state(None, 1, 4, True), unet_ref.init_state(None, 1, 4, True)
    perturb(sa, seed)
    perturb(sb, None if seed is None else seed + 10)
    se, ba, bb = unet_ref.clone_state(sb), {}, {}
    xl, xu = torch.from_numpy(d["xl"]), torch.from_numpy(d["xu"])
    rep = xu.shape[0] // xl.shape[0]
    xl1, yl1 = torch.from_numpy(d["xl1"]).repeat(rep, 1, 1, 1), torch.from_numpy(d["yl1"]).long().repeat(rep, 1, 1)
    n, hw = xl.shape[0] + xu.shape[0], xl.shape[-1]
    rows = []
    for j, cur in enumerate(d["cur_itrs"]):
        lr = laws_ref.medical_lr(int(cur), 0.01, 30000) if stepped_lr else 0.01
        r = steps_ref.hpfg_step(sa, sb, se, ba, bb, xl, torch.from_numpy(d["yl"]).long(), xl1, yl1, xu, torch.from_numpy(d["cutmix"][j]), int(cur), lr, lr, 0.1,
                                200.0, 0.99, 0.9, 5e-4, unpack_masks(d, f"it{j}_a", n, hw), unpack_masks(d, f"it{j}_b", n, hw),
                                unpack_masks(d, f"it{j}_t", n, hw))
        rows.append([r["loss"], r["sup"], r["semi"], r["pseudo_sup"], r["contrast"]])
    return {"losses": np.array(rows), "logits1_last": r["logits1"], "logits2_last": r["logits2"], "t_logits_last": r["t_logits"]}


def replay_s4cvnet(d, seed=None):
    torch.manual_seed(1337)
    sa, sb = unet_ref.init_state(None, 1, 4), unet_ref.init_state(None, 1, 4)
    perturb(sa, seed)
    perturb(sb, None if seed is None else seed + 10)
    se, ba, bb = unet_ref.clone_state(sb), {}, {}
    xl, yl, xu = torch.from_numpy(d["xl"]), torch.from_numpy(d["yl"]).long(), torch.from_numpy(d["xu"])
    nl, nu, hw = xl.shape[0], xu.shape[0], xl.shape[-1]
    rows = []
    for j, cur in enumerate(d["cur_itrs"]):
        lr = laws_ref.medical_lr(int(cur), 0.01, 30000)
        r = steps_ref.s4cvnet_step(sa, sb, se, ba, bb, xl, yl, xu, torch.from_numpy(d["noise"][j]), int(cur), lr, lr, 0.1, 200.0, 0.99, 0.9, 5e-4,
                                   unpack_masks(d, f"it{j}_a", nl + nu, hw), unpack_masks(d, f"it{j}_b", nl + nu, hw), unpack_masks(d, f"it{j}_t", nu, hw))
        rows.append([r["loss"], r["sup"], r["semi"], r["ps1"], r["ps2"], r["cons1"], r["cons2"]])
    return {"losses": np.array(rows), "logits1_last": r["logits1"], "logits2_last": r["logits2"], "t_logits_last": r["t_logits"]}


def unpack_labels2(packed, n):
    """Labels stored at 2 bits each (oracle/make_golden_r2.py) -> uint8 [n]."""
    bits = np.unpackbits(packed)[: 2 * n].reshape(n, 2)
    return (bits[:, 0] * 2 + bits[:, 1]).astype(np.uint8)


def sup224_inputs(d):
    """Inputs and per-iteration dropout masks of trace_sup224.npz, regenerated from its seeds and verified against its checksums."""
    from hpfg_amd.datasets.synthetic import synth_batch
    n, hw, iters, seed_x, seed_m = [int(v) for v in d["meta"]]
    x, lab = synth_batch(seed_x, n, hw, hw, 1, 4, 32)
    assert abs(float(x.double().sum()) - float(d["x_sum"])) < 1e-6 and int(lab.long().sum()) == int(d["lab_sum"]), "synthetic input stream differs"
    masks = []
    for k in range(1, iters + 1):
        torch.manual_seed(seed_m + k)
        ms = unet_ref.draw_dropout_masks(n, hw, hw)
        assert [float(m.sum()) for m in ms] == list(d["mask_sums"][k - 1]), "torch CPU generator stream differs from the fixture's"
        masks.append(ms)
    return x, lab, masks


def replay_sup224(d, seed=None):
    from oracle import losses_ref
    x, lab, masks = sup224_inputs(d)
    st, bufs = perturb(unet_ref.init_state(1, 1, 4), seed), {}
    table = laws_ref.cosine_table(0.01, 0, 1e-4, 1e-6, 200, 150)
    losses = [steps_ref.supervised_step(st, bufs, x, lab.long(), laws_ref.cosine_lr(k + 1, table), 0.9, 5e-4, masks[k])["loss"] for k in range(len(masks))]
    with torch.no_grad():
        fin = unet_ref.unet_forward(st, x, train=False)
    return {"losses": np.array(losses), "final_eval_logits": fin, "final_dice": losses_ref.mean_foreground_dice(fin.argmax(1).numpy(), lab.numpy(), 4)}


def control_drift(replay, d, keys, **kw):
    """max over the control seeds and `keys` of |perturbed - nominal| on the trace's final tensors."""
    nom = replay(d, None, **kw)
    drift = 0.0
    for s in CONTROL_SEEDS:
        got = replay(d, s, **kw)
        for k in keys:
            drift = max(drift, float((got[k] - nom[k]).abs().max()))
    return drift


# ---- round-3 fixtures (oracle/make_golden_r3.py): BASELINE-like sizes, inputs and dropout masks regenerated from seeds -------------
def _synth(seed, n, hw, in_ch, ncls, cell, want_sum, what):
    from hpfg_amd.datasets.synthetic import synth_batch
    x, lab = synth_batch(int(seed), int(n), int(hw), int(hw), in_ch, ncls, cell)
    assert abs(float(x.double().sum()) - float(want_sum)) < 1e-6, f"synthetic input stream differs ({what})"
    return x, lab


def _masks_from_seed(seed, shapes, want_sums):
    """Dropout keep-masks of one iteration: `shapes` = [(n, hw), ...] forwards in the reference's order, drawn from one seeded stream."""
    torch.manual_seed(int(seed))
    out = [unet_ref.draw_dropout_masks(n, hw, hw) for n, hw in shapes]
    got = [float(m.sum()) for ms in out for m in ms]
    assert got == [float(v) for v in want_sums], "torch CPU generator stream differs from the fixture's"
    return out


def mt224_inputs(d):
    nl, nu, hw, first, iters, sl, su, sm = [int(v) for v in d["meta"]]
    xl, yl = _synth(sl, nl, hw, 1, 4, 32, d["xl_sum"], "xl")
    xu, _ = _synth(su, nu, hw, 1, 4, 32, d["xu_sum"], "xu")
    assert int(yl.long().sum()) == int(d["yl_sum"])
    masks = [_masks_from_seed(sm + j, [(nl + nu, hw)] * 2, d["mask_sums"][j]) for j in range(iters)]
    return xl, yl, xu, masks, first


def replay_mt224(d, seed=None):
    xl, yl, xu, masks, first = mt224_inputs(d)
    st = perturb(unet_ref.init_state(1337, 1, 4), seed)
    ema, bufs = unet_ref.clone_state(st), {}
    rows = []
    for j, (ms, mt) in enumerate(masks):
        cur = first + j
        r = steps_ref.mean_teacher_step(st, ema, bufs, xl, yl.long(), xu, laws_ref.medical_lr(cur, 0.01, 30000), float(d["cons_w"]),
                                        laws_ref.ema_alpha(cur, 0.99), 0.9, 1e-4, ms, mt)
        rows.append([r["loss"], r["sup"], r["cons"]])
    return {"losses": np.array(rows), "student_logits_last": r["logits"], "teacher_logits_last": r["t_logits"]}


def cps96_inputs(d):
    nl, nu, hw, first, iters, sl, su, sm = [int(v) for v in d["meta"]]
    xl, yl = _synth(sl, nl, hw, 3, 2, 12, d["xl_sum"], "xl")
    xu, _ = _synth(su, nu, hw, 3, 2, 12, d["xu_sum"], "xu")
    masks = [_masks_from_seed(sm + j, [(nl + nu, hw)] * 2, d["mask_sums"][j]) for j in range(iters)]
    return xl, yl, xu, masks, first


def replay_cps96(d, seed=None):
    xl, yl, xu, masks, first = cps96_inputs(d)
    torch.manual_seed(1337)
    sa, sb = unet_ref.init_state(None, 3, 2), unet_ref.init_state(None, 3, 2)
    perturb(sa, seed)
    perturb(sb, None if seed is None else seed + 10)
    ba, bb, rows = {}, {}, []
    for j, (m1, m2) in enumerate(masks):
        lr = laws_ref.medical_lr(first + j, 0.01, 30000)
        cw = 0.1 * laws_ref.sigmoid_rampup((first + j) // 150, 200.0)
        r = steps_ref.cps_step(sa, sb, ba, bb, xl, yl.long(), xu, lr, lr, cw, 0.9, 1e-4, m1, m2)
        rows.append([r["loss"], r["sup"], r["semi"]])
    return {"losses": np.array(rows), "logits1_last": r["logits1"], "logits2_last": r["logits2"]}


def hpfg224_inputs(d):
    from hpfg_amd.utils import BoxMaskGenerator
    nl, nu, hw, sl, sl1, su, sm, srng = [int(v) for v in d["meta"]]
    xl, yl = _synth(sl, nl, hw, 1, 4, 32, d["xl_sum"], "xl")
    xl1, yl1 = _synth(sl1, nl, hw, 1, 4, 32, d["xl1_sum"], "xl1")
    xu, _ = _synth(su, nu, hw, 1, 4, 32, d["xu_sum"], "xu")
    gen = BoxMaskGenerator(prop_range=(0.25, 0.5), n_boxes=4, random_aspect_ratio=True, prop_by_area=True, within_bounds=True, invert=True)
    rng = np.random.RandomState(srng)
    cms = [torch.tensor(gen.generate_params(nu, (hw, hw), rng=rng), dtype=torch.float) for _ in d["cur_itrs"]]
    assert [float(c.sum()) for c in cms] == [float(v) for v in d["cutmix_sums"]], "CutMix mask stream differs from the fixture's"
    masks = [_masks_from_seed(sm + j, [(nl + nu, hw)] * 3, d["mask_sums"][j]) for j in range(len(d["cur_itrs"]))]
    return xl, yl, xl1, yl1, xu, cms, masks


def replay_hpfg224(d, seed=None):
    xl, yl, xl1, yl1, xu, cms, masks = hpfg224_inputs(d)
    torch.manual_seed(1)
    sa, sb = unet_ref.init_state(None, 1, 4, True), unet_ref.init_state(None, 1, 4, True)
    perturb(sa, seed)
    perturb(sb, None if seed is None else seed + 10)
    se, ba, bb = unet_ref.clone_state(sb), {}, {}
    rep = xu.shape[0] // xl.shape[0]
    xl1r, yl1r = xl1.repeat(rep, 1, 1, 1), yl1.long().repeat(rep, 1, 1)
    rows = []
    for j, cur in enumerate(d["cur_itrs"]):
        lr = laws_ref.medical_lr(int(cur), 0.01, 30000)
        ma, mb, mt = masks[j]
        r = steps_ref.hpfg_step(sa, sb, se, ba, bb, xl, yl.long(), xl1r, yl1r, xu, cms[j], int(cur), lr, lr, 0.1, 200.0, 0.99, 0.9, 5e-4, ma, mb, mt)
        rows.append([r["loss"], r["sup"], r["semi"], r["pseudo_sup"], r["contrast"]])
    return {"losses": np.array(rows), "logits1_last": r["logits1"], "logits2_last": r["logits2"], "t_logits_last": r["t_logits"]}


def grads224_inputs(d):
    n, hw, sx, sm = [int(v) for v in d["meta"]]
    x, lab = _synth(sx, n, hw, 1, 4, 32, d["x_sum"], "x")
    assert int(lab.long().sum()) == int(d["lab_sum"])
    (masks,) = _masks_from_seed(sm, [(n, hw)], d["mask_sums"])
    return x, lab, masks


def replay_grads224(d):
    from oracle import losses_ref
    x, lab, masks = grads224_inputs(d)
    st = unet_ref.init_state(1, 1, 4)
    names = steps_ref._train_state(st)
    o = unet_ref.unet_forward(st, x, True, masks)
    loss = losses_ref.med_sup_loss(o, lab.long())
    g = steps_ref._grads(loss, st, names)
    return {"loss": float(loss.detach()), "logits": o.detach(), "grads": g}


def sub_err(got: torch.Tensor, d, key: str) -> float:
    """max |got - fixture| over the stored sub-sample (every 8th pixel) of a logits tensor, and its checksums per element."""
    ref = torch.from_numpy(d[f"{key}_sub"])
    e = float((got[:, :, ::8, ::8].double() - ref.double()).abs().max())
    n = got.numel()
    e_sum = abs(float(got.double().sum()) - float(d[f"{key}_sum"])) / n
    e_abs = abs(float(got.double().abs().sum()) - float(d[f"{key}_abs"])) / n
    return max(e, e_sum, e_abs)


def emulation_drift(replay, d, keys, **kw):
    """The error model's own effect on a trace: |oracle with emulated split-bf16 convolutions - nominal fp32 oracle| on the trace's final
    tensors (oracle/bf16x3_ref.py).  This is the committed control of the bf16x3 tolerances: it perturbs every product exactly as the
    device arithmetic does (same hi / lo operands, same three partial products), not by a hand-picked epsilon."""
    from oracle import bf16x3_ref
    nom = replay(d, **kw)
    with bf16x3_ref.math_mode("bf16x3"):
        emu = replay(d, **kw)
    return max(float((emu[k] - nom[k]).abs().max()) for k in keys), emu


# ---- controls of the gradient bounds ---------------------------------------------------------------------------------------------------
def grad_ensemble(st, x, lab, masks, math: str, k_runs: int = 8, ncls_loss=None):
    """Control runs of one train-mode forward + backward (0.5 CE + 0.5 Dice) of the oracle, for the bounds of the device's gradients.

    A LeakyReLU sign or a max-pool arg-max that sits within rounding noise of a tie may fall either way in any correct implementation;
    one such flip moves every upstream weight gradient by 1e-3 ... 1e-1 (relative L2) on a 16..64-pixel input and averages out on
    224 x 224 ones.  How much, on THIS input, is measured by an ensemble: the oracle in the math mode's own arithmetic
    ("f32": fp32 products; "bf16x3": the emulated split-bf16 products of oracle/bf16x3_ref.py) re-run `k_runs` times from conv weights
    perturbed at the mode's noise level -- relative 1e-6 for exact fp32 (summation order, fp32-vs-fp64 BatchNorm statistics: about ten
    roundings), 2^-18 for split-bf16 (half an ulp of the lo word: what re-quantisation of an operand moves) -- plus once with fp64
    accumulation -- plus three SUMMATION-ORDER controls with unperturbed weights (round 5): the same network function evaluated on the
    mirrored problem (input, labels, dropout masks and every 3 x 3 kernel flipped along W, along H, along both; gradients flipped back).
    A convolution, BatchNorm's batch sums, bilinear x2 with align_corners and the Dice sums are all mirror-symmetric, so mathematically
    nothing changes -- but every sum runs over its terms in another order and a 2 x 2 max-pool tie resolves to another element, which is
    exactly what separates two correct kernels with different reduction orders (a correct kernel with a new tiling used to fail the
    16-pixel case on the luck of one tie; VERDICT r4 weak spot 1).  Returns (nominal gradients of that arithmetic, [ensemble gradients])."""
    from oracle import bf16x3_ref, losses_ref

    def run(mode, seed, eps, flip=()):
        st2 = unet_ref.clone_state(st)
        if seed is not None:
            g = torch.Generator().manual_seed(4000 + seed)
            for n_ in st2:
                if n_.endswith(".weight") and st2[n_].dim() == 4:
                    r = 2 * torch.rand(st2[n_].shape, generator=g, dtype=torch.float64) - 1
                    st2[n_] = (st2[n_].double() * (1 + eps * r)).float()
        xf, lf, mf = x, lab, masks
        if flip:
            for n_ in st2:
                if n_.endswith(".weight") and st2[n_].dim() == 4 and st2[n_].shape[-1] == 3:
                    st2[n_] = st2[n_].flip(flip).contiguous()
            xf, lf = x.flip(flip).contiguous(), lab.flip([d - 1 for d in flip]).contiguous()
            mf = None if masks is None else [m_.flip(flip).contiguous() for m_ in masks]
        names = steps_ref._train_state(st2)
        with bf16x3_ref.math_mode(mode):
            o = unet_ref.unet_forward(st2, xf, True, mf, track_running=False)
            gr = steps_ref._grads(losses_ref.med_sup_loss(o, lf.long()), st2, names)
        if flip:
            gr = {k_: (v.flip(flip).contiguous() if v.dim() == 4 and v.shape[-1] == 3 else v) for k_, v in gr.items()}
        return gr

    mode, eps = ("f32", 1e-6) if math == "f32" else ("bf16x3", 2.0 ** -18)
    nominal = run(mode, None, 0.0)
    ens = [run(mode + "_f64acc" if mode == "bf16x3" else "f64acc", None, 0.0)] + [run(mode, s, eps) for s in range(k_runs)]
    ens += [run(mode, None, 0.0, flip=f) for f in ((3,), (2,), (2, 3))]
    return nominal, ens


def rel_l2(a: torch.Tensor, b: torch.Tensor, floor: float = 1e-4) -> float:
    return float((a.double() - b.double()).norm() / max(floor, float(b.double().norm())))
