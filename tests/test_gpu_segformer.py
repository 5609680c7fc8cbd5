"""SegFormer-B0 (hpfg_amd/model/segformer.py: HIP LayerNorm / attention / DWConv-GELU + library GEMMs) against the CPU oracle, which is
itself pinned to the reference module (tests/golden/segformer_b0.npz): same seed-initialised weights, same inputs, same random draws."""
import numpy as np
import pytest
import torch

from hpfg_amd.model import SegFormer, build_model
from hpfg_amd.utils import AttrDict, Med_Sup_Loss
from oracle import losses_ref, segformer_ref as S
from tests.helpers import maxerr

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_state_dict_and_golden_fixture(golden_dir):
    d = np.load(f"{golden_dir}/segformer_b0.npz")
    torch.manual_seed(1337)
    m = build_model(AttrDict(model="segformer", in_channels=1, num_classes=4, train_crop_size=[64, 64]))
    st = S.init_state(1337, 1, 4)
    sd = m.state_dict()
    assert list(sd.keys()) == list(st.keys())
    assert all(torch.equal(sd[k], st[k]) for k in st)              # same constructor order, same generator consumption
    m = m.to(DEV)
    x, y = torch.from_numpy(d["x"]).to(DEV), torch.from_numpy(d["y"]).to(DEV)
    m.eval()
    with torch.no_grad():
        assert maxerr(m(x).cpu(), torch.from_numpy(d["eval_logits"])) < 1e-3
    m.train()
    torch.manual_seed(99)
    m.external_draws = S.draw_randomness(2)
    out = m(x)
    assert maxerr(out.detach().cpu(), torch.from_numpy(d["train_logits"])) < 1e-3
    loss = Med_Sup_Loss(4)(out, y)
    assert abs(float(loss) - float(d["loss"])) < 1e-4
    loss.backward()
    for k, p in m.named_parameters():
        ref = d["g:" + k]
        got = np.array([float(p.grad.sum()), float(p.grad.abs().sum()), float(p.grad.abs().max())])
        assert np.abs(got - ref).max() < 2e-3 * max(1.0, float(np.abs(ref).max())), (k, got, ref)


@pytest.mark.parametrize("size,B,head", [(224, 2, True), (96, 3, True), (96, 3, False)])
def test_train_forward_backward_vs_oracle(size, B, head):
    """Full-size tokens (3136 / 784 / 196 / 49 queries against 49 keys at 224x224): logits, loss and every gradient tensor.
    head: SegFormerHead.fuse_lowres -- True (default) applies linear_fuse per stage at the stage's resolution and adds the resized maps, False is the
    reference's literal concat + one GEMM (model/segformer.py:309-315); the oracle follows the reference's form, both must agree with it."""
    torch.manual_seed(5)
    m = SegFormer(image_size=[size, size], in_channels=1, num_classes=4).to(DEV)
    m.decoder.fuse_lowres = head
    st = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(size)
    x = torch.randn(B, 1, size, size, generator=g)
    y = torch.randint(0, 4, (B, size, size), generator=g)
    torch.manual_seed(7)
    dp, mask = S.draw_randomness(B)
    m.train()
    m.external_draws = (dp, mask)
    out = m(x.to(DEV))
    loss = Med_Sup_Loss(4)(out, y.to(DEV))
    loss.backward()
    names = [k for k in st if st[k].is_floating_point() and "running" not in k]
    for k in names:
        st[k] = st[k].requires_grad_(True)
    ro = S.segformer_forward(st, x, True, dp, mask)
    rl = losses_ref.med_sup_loss(ro, y)
    rg = dict(zip(names, torch.autograd.grad(rl, [st[k] for k in names])))
    assert maxerr(out.detach().cpu(), ro.detach()) < 1e-3
    assert abs(float(loss) - float(rl)) < 1e-4
    # Per-tensor relative L2 error.  The head applies ReLU right after a BatchNorm with beta = 0: of its 1.6 M pre-activations one or two
    # lie within fp32 rounding of zero, and which side they fall on depends on the summation order of the GEMM in front (library kernel
    # choice).  One flipped gate removes one element of the gradient: ~1/sqrt(1.6e6) = 8e-4 of the gradient's norm, inherited by every
    # tensor upstream -- the signature is d(beta) = sum g off by ~1e-4 while d(gamma) = sum g*xhat (xhat ~ 0 there) stays exact.  Without a
    # flip every tensor agrees to ~2e-6 (checked against an fp64 run of the oracle).  Hence: 1e-2 per tensor, 5e-3 for the median.
    # (the biases in front of the head's BatchNorm have an exactly zero gradient: both sides hold ~1e-8 of rounding noise, hence the floor)
    errs = {k: float((p.grad.cpu().double() - rg[k].double()).norm() / max(1e-5, float(rg[k].double().norm()))) for k, p in m.named_parameters()}
    bad = {k: v for k, v in errs.items() if not v < 1e-2}
    assert not bad, bad
    assert float(np.median(list(errs.values()))) < 5e-3, sorted(((v, k) for k, v in errs.items()), reverse=True)[:20]
    assert maxerr(m.state_dict()["decoder.linear_fuse.bn.running_var"].cpu(), st["decoder.linear_fuse.bn.running_var"]) < 1e-4


def test_ctct_step_trace(golden_dir):
    """CTCTStep (U-Net on the HIP engine + SegFormer, SGD + AdamW) against two iterations of the reference's own modules."""
    from hpfg_amd import engine as E
    from hpfg_amd.model import UNet
    from hpfg_amd.train import CTCTStep
    d = np.load(f"{golden_dir}/trace_ctct.npz")
    torch.manual_seed(1)
    m1 = UNet(1, 4)
    m2 = SegFormer(image_size=[64, 64], in_channels=1, num_classes=4)          # second construction continues the generator, as in the driver
    m1, m2 = m1.to(DEV), m2.to(DEV)
    m1.train()
    m2.train()
    opt = dict(opt="sgd", lr=0.01, momentum=0.9, weight_decay=5e-4, sched="medical", total_itrs=30000, step_size=1500, warmup_epochs=1, warmup_lr=1e-4,
               min_lr=1e-6)
    a = AttrDict(dict(model1=AttrDict(opt), model2=AttrDict(dict(opt, opt="adamW", lr=0.0008, weight_decay=0.05)), consistency=0.1, consistency_rampup=200.0))
    st = CTCTStep(m1, m2, a)
    xl, yl, xu = (torch.from_numpy(d[k]).to(DEV) for k in ("xl", "yl", "xu"))
    rows = []
    for k in range(2):
        masks = {}
        for lvl in range(5):
            c, h = E.WIDTHS[lvl], 64 >> lvl
            bits = np.unpackbits(d[f"it{k}_u{lvl}"])[: 4 * c * h * h].reshape(4, c, h, h)
            masks[E.enc_prefix(lvl) + ".0"] = torch.from_numpy(bits).permute(0, 2, 3, 1).contiguous().to(DEV)
        m1.external_dropout_masks = masks
        dp = [None if none else torch.from_numpy(x) for x, none in zip(d[f"it{k}_dp"], d[f"it{k}_dpnone"])]
        m2.external_draws = (dp, torch.from_numpy(d[f"it{k}_mask"]))
        r = st.step(xl, yl, xu, k + 1, cons_w=float(d["cons_w"]))
        p1, p2 = r["parts1"].cpu(), r["parts2"].cpu()
        rows.append([float(r["loss"]), 0.5 * float(p1[1]) + 0.5 * float(p1[2]), 0.5 * float(p2[1]) + 0.5 * float(p2[2]), float(p1[4]), float(p2[4])])
    assert np.abs(np.array(rows) - d["losses"]).max() < 1e-3, (rows, d["losses"])
    assert maxerr(r["logits1"].cpu(), torch.from_numpy(d["logits1_last"])) < 1e-3
    # The second iteration's SegFormer forward runs on weights after ONE AdamW step, and Adam's first step is lr * sign(g) for every
    # element: wherever a gradient is at rounding level its sign -- hence a full +-8e-4 step -- differs between two fp32 evaluations.
    # The losses above still agree to 1e-3; the logits are bounded by a few such steps.
    assert maxerr(r["logits2"].cpu(), torch.from_numpy(d["logits2_last"])) < 2e-2


def test_ctct_step_captures_into_a_graph():
    """The whole cross-teaching step (both networks, both optimizers: FusedSGD and capturable AdamW) replays from one hipGraph."""
    from hpfg_amd.datasets.synthetic import synth_batch
    from hpfg_amd.model import UNet
    from hpfg_amd.train import CTCTStep, GraphedStep
    torch.manual_seed(3)
    m1, m2 = UNet(1, 4).to(DEV), SegFormer(image_size=[64, 64], in_channels=1, num_classes=4).to(DEV)
    m1.train()
    m2.train()
    opt = dict(opt="sgd", lr=0.01, momentum=0.9, weight_decay=5e-4, sched="medical", total_itrs=30000, step_size=1500, warmup_epochs=1, warmup_lr=1e-4,
               min_lr=1e-6)
    a = AttrDict(dict(model1=AttrDict(opt), model2=AttrDict(dict(opt, opt="adamW", lr=0.0008, weight_decay=0.05)), consistency=0.1, consistency_rampup=200.0))
    xl, yl = synth_batch(1, 4, 64, 64, 1, 4, 8)
    xu, _ = synth_batch(2, 4, 64, 64, 1, 4, 8)
    xl, yl, xu = xl.to(DEV), yl.to(DEV), xu.to(DEV)
    gs = GraphedStep(CTCTStep(m1, m2, a), [xl, yl, xu], warmup=2, alias_inputs=True)
    w0 = m2.decoder.linear_pred.weight.detach().clone()
    losses = [float(gs.step([xl, yl, xu], 3 + k)["loss"]) for k in range(12)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert float((m2.decoder.linear_pred.weight.detach() - w0).abs().max()) > 1e-4          # AdamW ran inside the graph
    assert abs(float(a.model2.lr) - 0.0008) < 1e-12 and float(gs.s.optimizer2.param_groups[0]["lr"]) < 0.0008      # the tensor lr follows the schedule


def test_eval_path_runs_the_segformer():
    """hpfg_amd.val.test_single_volume (the reference's evaluation signature) on a SegFormer: slices resized to the network size, eval-mode
    forward, arg-max and Dice counts on the device -- against the oracle forward on the same weights."""
    from hpfg_amd.val import test_single_volume
    torch.manual_seed(11)
    m = SegFormer(image_size=[64, 64], in_channels=1, num_classes=4).to(DEV)
    st = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(4)
    vol = torch.rand(1, 5, 80, 72, generator=g)
    lab = torch.randint(0, 4, (1, 5, 80, 72), generator=g)
    got = np.array(test_single_volume(vol, lab, m, classes=4, patch_size=[64, 64]))
    assert m.training            # the evaluation restores the mode it found
    from scipy.ndimage import zoom                      # the reference's resize (val.py:274,280), slice by slice
    xs = np.stack([zoom(sl, (64 / 80, 64 / 72), order=0) for sl in vol[0].numpy()])
    with torch.no_grad():
        logits = S.segformer_forward(st, torch.from_numpy(xs).float().unsqueeze(1), False)
    pred = np.stack([zoom(p, (80 / 64, 72 / 64), order=0) for p in logits.argmax(1).numpy().astype(np.uint8)])
    want = [losses_ref.binary_dice(pred == c, lab[0].numpy() == c) for c in range(1, 4)]
    assert np.abs(got[:, 0] - np.array(want)).max() < 2e-3, (got[:, 0], want)
    assert got.shape == (3, 2) and np.isfinite(got).all()


def test_ctct_step_full_size_vs_oracle():
    """BASELINE configs[4] at its real size: one CTCT iteration (2021_12_MIDL_CTCT_ACDC.py:117-134) of U-Net + SegFormer-B0 on 8 labelled +
    24 unlabelled 224x224 images against oracle.steps_ref.ctct_step (pinned to the reference's own modules by tests/golden/trace_ctct.npz):
    every loss term, both networks' logits, and the SGD / AdamW updates through the logits of a second iteration's forward."""
    from hpfg_amd.datasets.synthetic import synth_batch
    from hpfg_amd.model import UNet
    from hpfg_amd.train import CTCTStep
    from hpfg_amd.utils import AttrDict
    from oracle import laws_ref, steps_ref
    from tests.helpers import engine_masks, state_from_module
    NL, NU, HW = 8, 24, 224
    torch.manual_seed(1)
    m1, m2 = UNet(1, 4).to(DEV), SegFormer(image_size=[HW, HW], in_channels=1, num_classes=4).to(DEV)
    m1.train()
    m2.train()
    s1, s2 = state_from_module(m1), {k: v.detach().cpu().clone() for k, v in m2.state_dict().items()}
    opt = dict(opt="sgd", lr=0.01, momentum=0.9, weight_decay=5e-4, sched="medical", total_itrs=30000, step_size=1500, warmup_epochs=1, warmup_lr=1e-4, min_lr=1e-6)
    a = AttrDict(dict(model1=AttrDict(opt), model2=AttrDict(dict(opt, opt="adamW", lr=0.0008, weight_decay=0.05)), consistency=0.1, consistency_rampup=200.0))
    st = CTCTStep(m1, m2, a)
    xl, yl = synth_batch(31, NL, HW, HW, 1, 4, 32)
    xu, _ = synth_batch(32, NU, HW, HW, 1, 4, 32)
    torch.manual_seed(77)
    draws = S.draw_randomness(NL + NU)
    m2.external_draws = draws
    w = 0.05
    r = st.step(xl.to(DEV), yl.to(DEV), xu.to(DEV), 1, cons_w=w)
    eng = next(iter(m1._engines.values()))[0]
    masks = engine_masks(eng, m1._seed_counter, NL + NU, HW, HW)
    bufs1, adam2 = {}, {}
    ro = steps_ref.ctct_step(s1, s2, bufs1, adam2, xl, yl.long(), xu, laws_ref.medical_lr(1, 0.01, 30000), laws_ref.medical_lr(1, 0.0008, 30000), w, 0.9, 5e-4, 0.05,
                             masks, draws)
    p1, p2 = r["parts1"].cpu(), r["parts2"].cpu()
    got = [float(r["loss"]), 0.5 * float(p1[1]) + 0.5 * float(p1[2]), 0.5 * float(p2[1]) + 0.5 * float(p2[2]), float(p1[4]), float(p2[4])]
    ref = [ro["loss"], ro["sup1"], ro["sup2"], ro["ps1"], ro["ps2"]]
    assert max(abs(x - y) for x, y in zip(got, ref)) < 1e-3, (got, ref)
    assert maxerr(r["logits1"].cpu(), ro["logits1"]) < 1e-3
    assert maxerr(r["logits2"].cpu(), ro["logits2"]) < 1e-3
    # the updates: the U-Net after one SGD step agrees to 1e-3 in its next (eval) forward; the SegFormer after one AdamW step -- lr * sign(g)
    # for every element, so gradients at rounding level move by a full +-8e-4 either way (see test_ctct_step_trace) -- to a few such steps
    m1.eval()
    m2.eval()
    with torch.no_grad():
        e1 = m1(xl.to(DEV)).cpu()
        e2 = m2(xl.to(DEV)).cpu()
        from oracle import unet_ref
        f1 = unet_ref.unet_forward(s1, xl, train=False)
        f2 = S.segformer_forward(s2, xl, False)
    assert maxerr(e1, f1) < 1e-3
    assert maxerr(e2, f2) < 2e-2
