"""CPU: the oracle (oracle/*.py) reproduces every golden vector that oracle/make_golden.py captured from the REFERENCE's own
modules (/root/reference/model/unet.py, utils/loss/*.py, utils/scheduler/*.py, utils/utils.py).  This is what pins the oracle; the
GPU tests then compare the HIP path with the same fixtures and with the oracle."""
import json

import numpy as np
import pytest
import torch

from oracle import laws_ref, losses_ref, steps_ref, unet_ref
from tests import trace_replay as R

torch.set_num_threads(4)


def _unpack_masks(d, prefix, n, hw):
    out = []
    for lvl in range(5):
        c, h = unet_ref.WIDTHS[lvl], hw >> lvl
        bits = np.unpackbits(d[f"{prefix}{lvl}"])[: n * c * h * h].reshape(n, c, h, h)
        out.append(torch.from_numpy(bits.astype(np.float32)))
    return out


def test_anchor_values(golden_dir):
    a = json.load(open(f"{golden_dir}/anchors.json"))
    assert a["param_count"] == 1813764 and a["param_count_plus"] == 3663620
    st = unet_ref.init_state(1, 1, 4)
    assert sum(st[k].numel() for k in unet_ref.param_names(st)) == a["param_count"]
    for k in range(1, 7):
        assert abs(laws_ref.medical_lr(k, 0.01, 30000) - a["medical_lr_first6"][k - 1]) < 1e-12
    table = laws_ref.cosine_table(0.01, 0, 1e-4, 1e-6, 200, 150)
    for k in range(1, 7):
        assert abs(laws_ref.cosine_lr(k, table) - a["cosine_lr_first6"][k - 1]) < 1e-12
    assert abs(a["cosine_lr_first6"][0] - 1e-6) < 1e-7        # first optimizer step runs at ~final_lr (reference quirk)
    for e, v in zip(a["rampup_epochs"], a["rampup_sigmoid"]):
        assert abs(laws_ref.sigmoid_rampup(e, 200.0) - v) < 1e-12
    for e, v in zip(a["rampup_epochs"], a["rampup_linear"]):
        assert abs(laws_ref.linear_rampup(e, 200.0) - v) < 1e-12
    for s, v in zip((1, 2, 3, 200), a["ema_alpha_first4"]):
        assert abs(laws_ref.ema_alpha(s, 0.99) - v) < 1e-15
    m = laws_ref.box_masks(5, (64, 64), np.random.RandomState(1))
    assert float(m.sum()) == a["box_mask_seed1_n5_64_sum"]


def test_eval_forward_anchor(golden_dir):
    a = json.load(open(f"{golden_dir}/anchors.json"))
    st = unet_ref.init_state(1, 1, 4)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(8, 1, 224, 224, generator=g)
    t = torch.randint(0, 4, (8, 224, 224), generator=g)
    with torch.no_grad():
        y = unet_ref.unet_forward(st, x, train=False)
    assert abs(float(y.double().sum()) - a["eval_logits_sum"]) < 0.05
    assert abs(float(y.abs().mean()) - a["eval_logits_meanabs"]) < 1e-6
    assert np.allclose(y[0, :, 0, 0].numpy(), a["eval_logits_px00"], atol=1e-6)
    assert abs(float(losses_ref.med_sup_loss(y, t)) - a["eval_med_sup_loss"]) < 1e-6


@pytest.mark.parametrize("tag", ["a", "b"])
def test_train_forward_backward_fixture(golden_dir, tag):
    d = np.load(f"{golden_dir}/unet_fwd_bwd_{tag}.npz")
    n, hw, in_ch, ncls, seed, _ = [int(v) for v in d["meta"]]
    st = unet_ref.init_state(seed, in_ch, ncls)
    masks = _unpack_masks(d, "mask", n, hw)
    names = steps_ref._train_state(st)
    taps = {}
    out = unet_ref.unet_forward(st, torch.from_numpy(d["x"]), True, masks, taps=taps)
    loss = losses_ref.med_sup_loss(out, torch.from_numpy(d["labels"]).long())
    grads = steps_ref._grads(loss, st, names)
    assert float((out.detach() - torch.from_numpy(d["logits"])).abs().max()) < 1e-5
    assert abs(float(loss) - float(d["loss"])) < 1e-6
    for k in d.files:
        if k.startswith("grad/"):
            ref = torch.from_numpy(d[k])
            assert float((grads[k[5:]] - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max())), k
        if k.startswith("raw/"):
            assert float((taps[k[4:]].detach() - torch.from_numpy(d[k])).abs().max()) < 1e-5, k
        if k.startswith("bn/"):
            assert float((st[k[3:]] - torch.from_numpy(d[k])).abs().max()) < 1e-6, k


def test_losses_fixture(golden_dir):
    d = np.load(f"{golden_dir}/losses.npz")
    lg, tl, lab = torch.from_numpy(d["logits"]), torch.from_numpy(d["t_logits"]), torch.from_numpy(d["labels"])
    p = torch.softmax(lg, 1)
    assert abs(float(losses_ref.dice_loss(p, lab.unsqueeze(1))) - float(d["dice"])) < 1e-6
    assert abs(float(losses_ref.dice_loss(p, lab.float())) - float(d["dice_float"])) < 1e-6
    assert abs(float(losses_ref.med_sup_loss(lg, lab)) - float(d["med"])) < 1e-6
    assert abs(float(losses_ref.cross_entropy(lg, lab)) - float(d["ce"])) < 1e-6
    assert abs(float(losses_ref.mse_consistency(p, torch.softmax(tl, 1))) - float(d["mse"])) < 1e-7
    hg = (torch.from_numpy(d["hg0"]), torch.from_numpy(d["hg1"]))
    tg = (torch.from_numpy(d["tg0"]), torch.from_numpy(d["tg1"]))
    assert abs(float(losses_ref.dense_loss(hg, tg)) - float(d["dense"])) < 1e-5
    x = lg.clone().requires_grad_(True)
    comp = losses_ref.med_sup_loss(x[:1], lab[:1]) + 0.3 * losses_ref.mse_consistency(torch.softmax(x[1:], 1), torch.softmax(tl[1:], 1))
    comp.backward()
    assert abs(float(comp) - float(d["comp"])) < 1e-6
    assert float((x.grad - torch.from_numpy(d["comp_dlogits"])).abs().max()) < 1e-8


def test_supervised_and_mean_teacher_traces(golden_dir):
    d = np.load(f"{golden_dir}/trace_sup.npz")
    r = R.replay_sup(d)
    assert np.abs(r["losses"] - d["losses"]).max() < 2e-5
    fin = r["final_eval_logits"]
    assert float((fin - torch.from_numpy(d["final_eval_logits"])).abs().max()) < 2e-4
    assert abs(losses_ref.mean_foreground_dice(fin.argmax(1).numpy(), d["labels"], 4) - float(d["final_dice"])) < 1e-6

    d = np.load(f"{golden_dir}/trace_mt.npz")
    r = R.replay_mt(d)
    assert np.abs(r["losses"] - d["losses"]).max() < 2e-5
    assert float((r["student_logits_last"] - torch.from_numpy(d["student_logits_last"])).abs().max()) < 1e-4


def test_ict_trace(golden_dir):
    d = np.load(f"{golden_dir}/trace_ict.npz")
    r = R.replay_ict(d)
    assert np.abs(r["losses"] - d["losses"]).max() < 2e-5
    assert float((r["student_logits_last"] - torch.from_numpy(d["student_logits_last"])).abs().max()) < 1e-4


def test_uamt_trace(golden_dir):
    d = np.load(f"{golden_dir}/trace_uamt.npz")
    r = R.replay_uamt(d)
    assert np.abs(r["losses"] - d["losses"]).max() < 2e-5
    assert float((r["uncertainty_last"] - torch.from_numpy(d["uncertainty_last"])).abs().max()) < 1e-5


def test_hpfg2_trace_gate_repeat_and_stepped_schedulers(golden_dir):
    """main.py:142-143 (labelled batch repeated Nu//Nl = 3 times), :186-188 (no MSE before iteration 1000) and :211-212 (both
    Medical_LR schedulers stepped): three iterations of the reference's own modules at cur_itrs = 999, 1000, 1001, batch 2+6."""
    d = np.load(f"{golden_dir}/trace_hpfg2.npz")
    assert d["losses"][0, 5] == 0.0 and d["losses"][1, 5] > 0.0            # the consistency term switches on at 1000
    for cur, (lr1, lr2) in zip(d["cur_itrs"], d["lrs"]):
        assert abs(laws_ref.medical_lr(int(cur), 0.01, 30000) - lr1) < 1e-12 and lr1 == lr2
    r = R.replay_hpfg(d, stepped_lr=True)
    assert np.abs(r["losses"] - d["losses"][:, :5]).max() < 1e-4
    for k in ("logits1_last", "logits2_last", "t_logits_last"):
        assert float((r[k] - torch.from_numpy(d[k])).abs().max()) < 2e-4, k


def test_s4cvnet_trace(golden_dir):
    """2022_08_CVPR_S4CVNet_ACDC.py:107-167: three iterations of the reference's own modules across the iteration-1000 gate."""
    d = np.load(f"{golden_dir}/trace_s4cvnet.npz")
    assert d["losses"][0, 5] == 0.0 and d["losses"][1, 5] > 0.0 and d["losses"][1, 6] > 0.0
    r = R.replay_s4cvnet(d)
    assert np.abs(r["losses"] - d["losses"]).max() < 1e-4
    for k in ("logits1_last", "logits2_last", "t_logits_last"):
        assert float((r[k] - torch.from_numpy(d[k])).abs().max()) < 2e-4, k


def test_sup224_trace_cfg1_as_written(golden_dir):
    """BASELINE configs[0] at its real size: 10 supervised iterations on 8 slices of 224x224 (sup_ACDC.py:83-93), inputs and dropout
    masks regenerated from the seeds the reference run used (checksums stored in the fixture)."""
    d = np.load(f"{golden_dir}/trace_sup224.npz")
    r = R.replay_sup224(d)
    assert np.abs(r["losses"] - d["losses"]).max() < 5e-5
    fin = r["final_eval_logits"]
    assert float((fin[:, :, ::8, ::8] - torch.from_numpy(d["final_eval_logits_sub"])).abs().max()) < 5e-4
    assert abs(float(fin.double().sum()) - float(d["final_eval_logits_sum"])) < 1e-5 * fin.numel()      # mean deviation < 1e-5
    assert abs(r["final_dice"] - float(d["final_dice"])) < 1e-3


def test_augment_fixture_pins_the_augmentation_oracle(golden_dir):
    """oracle/augment_ref.py against outputs of the reference's own RandomGenerator (datasets/utils.py:99-117) on seeded slices."""
    import random
    from oracle import augment_ref
    d = np.load(f"{golden_dir}/augment.npz")
    n = int(d["n"])
    for k, seed in enumerate(d["seeds"]):
        img, lab = d[f"src_img{k % n}"], d[f"src_lab{k % n}"]
        oi, ol = augment_ref.random_generator(img, lab, (224, 224), random.Random(int(seed)), np.random.RandomState(int(seed)))
        assert np.array_equal(oi[0].astype(np.float16), d[f"img{k}"][0]) and abs(float(oi.astype(np.float64).sum()) - float(d[f"img{k}_sum"])) < 1e-6
        assert np.array_equal(ol.reshape(-1), R.unpack_labels2(d[f"lab{k}"], ol.size))


def test_segformer_oracle_matches_reference_fixture(golden_dir):
    """Groundwork for SURVEY.md section 8f row 1: oracle.segformer_ref against outputs of the reference's own SegFormer-B0."""
    from oracle import segformer_ref as S
    d = np.load(f"{golden_dir}/segformer_b0.npz")
    st = S.init_state(1337, 1, 4)
    assert sum(v.numel() for k, v in st.items() if v.is_floating_point() and "running" not in k) == int(d["n_params"]) == 3712036
    x, y = torch.from_numpy(d["x"]), torch.from_numpy(d["y"]).long()
    with torch.no_grad():
        assert float((S.segformer_forward(st, x, False) - torch.from_numpy(d["eval_logits"])).abs().max()) < 2e-5
    torch.manual_seed(99)
    dp, mask = S.draw_randomness(2)
    names = [k for k in st if st[k].is_floating_point() and "running" not in k]
    for k in names:
        st[k] = st[k].clone().requires_grad_(True)
    taps = {}
    out = S.segformer_forward(st, x, True, dp, mask, taps=taps)
    assert float((out - torch.from_numpy(d["train_logits"])).abs().max()) < 2e-5
    assert float((taps["stage4"] - torch.from_numpy(d["stage4"])).abs().max()) < 2e-5
    loss = losses_ref.med_sup_loss(out, y)
    assert abs(float(loss) - float(d["loss"])) < 1e-6
    gs = torch.autograd.grad(loss, [st[k] for k in names])
    for k, g in zip(names, gs):
        ref = d["g:" + k]
        got = np.array([float(g.sum()), float(g.abs().sum()), float(g.abs().max())])
        assert np.abs(got - ref).max() < 1e-4 * max(1.0, float(np.abs(ref).max())), k


def _ctct_draws(d, k):
    dp = [None if none else torch.from_numpy(x) for x, none in zip(d[f"it{k}_dp"], d[f"it{k}_dpnone"])]
    return dp, torch.from_numpy(d[f"it{k}_mask"])


def test_ctct_trace(golden_dir):
    """U-Net + SegFormer cross teaching (2021_12_MIDL_CTCT_ACDC.py:117-134): two iterations of the reference's own modules."""
    from oracle import segformer_ref as S
    d = np.load(f"{golden_dir}/trace_ctct.npz")
    st1 = unet_ref.init_state(1, 1, 4)
    st2 = S.init_state(None, 1, 4)
    bufs1, adam2 = {}, {}
    xl, yl, xu = torch.from_numpy(d["xl"]), torch.from_numpy(d["yl"]).long(), torch.from_numpy(d["xu"])
    rows = []
    for k in range(2):
        r = steps_ref.ctct_step(st1, st2, bufs1, adam2, xl, yl, xu, laws_ref.medical_lr(k + 1, 0.01, 30000), laws_ref.medical_lr(k + 1, 0.0008, 30000),
                                float(d["cons_w"]), 0.9, 5e-4, 0.05, _unpack_masks(d, f"it{k}_u", 4, 64), _ctct_draws(d, k))
        rows.append([r["loss"], r["sup1"], r["sup2"], r["ps1"], r["ps2"]])
    assert np.abs(np.array(rows) - d["losses"]).max() < 3e-5
    assert float((r["logits2"] - torch.from_numpy(d["logits2_last"])).abs().max()) < 1e-4


def test_cps_and_hpfg_traces(golden_dir):
    d = np.load(f"{golden_dir}/trace_cps.npz")
    r = R.replay_cps(d)
    assert np.abs(r["losses"] - d["losses"]).max() < 5e-5

    d = np.load(f"{golden_dir}/trace_hpfg.npz")
    r = R.replay_hpfg(d)
    assert np.abs(r["losses"] - d["losses"]).max() < 1e-4


# ---- round-3 fixtures: reference-module traces at BASELINE-like sizes (oracle/make_golden_r3.py) ---------------------------------
def _r3(name):
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))


@pytest.mark.parametrize("name,replay,keys,ncol", [
    ("trace_mt224.npz", "replay_mt224", ("student_logits_last", "teacher_logits_last"), 3),
    ("trace_cps96.npz", "replay_cps96", ("logits1_last", "logits2_last"), 3),
    ("trace_hpfg224.npz", "replay_hpfg224", ("logits1_last", "logits2_last", "t_logits_last"), 5),
])
def test_r3_trace_oracle_and_error_model(name, replay, keys, ncol):
    """(a) the fp32 oracle reproduces the reference's trace; (b) the SAME oracle with emulated split-bf16 convolutions (the device's
    error model, oracle/bf16x3_ref.py) stays inside the flat 1e-3 of BASELINE.json on losses and logits at these sizes."""
    from oracle import bf16x3_ref
    from tests import trace_replay as R
    d = _r3(name)
    fn = getattr(R, replay)
    nom = fn(d)
    assert np.abs(nom["losses"][:, :ncol] - d["losses"][:, :ncol]).max() < 1e-4
    for k in keys:
        assert R.sub_err(nom[k], d, k) < 5e-4, k
    with bf16x3_ref.math_mode("bf16x3"):
        emu = fn(d)
    assert np.abs(emu["losses"][:, :ncol] - d["losses"][:, :ncol]).max() < 1e-3
    for k in keys:
        assert R.sub_err(emu[k], d, k) < 1e-3, (k, R.sub_err(emu[k], d, k))


def test_r3_gradients_oracle_vs_reference():
    from tests import trace_replay as R
    d = _r3("grads224.npz")
    r = R.replay_grads224(d)
    assert abs(r["loss"] - float(d["loss"])) < 1e-6
    assert R.sub_err(r["logits"], d, "logits") < 2e-5
    for k, g in r["grads"].items():
        nrm = float(d[f"g:{k}:norm"])
        flat = g.reshape(-1)
        smp = flat[:: max(1, flat.numel() // 64)][:64].numpy()
        assert np.abs(smp - d[f"g:{k}:sample"]).max() <= 2e-4 * nrm + 2e-6, k      # (+2e-6: biases in front of a train-mode BatchNorm carry rounding noise only)
        assert abs(float(g.double().norm()) - nrm) <= 2e-4 * nrm + 2e-6, k
