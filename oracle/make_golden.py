"""Oracle pinning (test infrastructure; runs ONLY in the build container, never on the GPU box).

Imports the reference's own hot-path files from /root/reference one by one (they cannot be
imported as packages: SURVEY.md section 8c), drives them with seeded synthetic inputs, checks that
the oracle restatement in this directory reproduces them, and writes the reference's outputs as
small golden fixtures under tests/golden/ (data only: inputs by seed, outputs as arrays).

    python -m oracle.make_golden            # validate + (re)write tests/golden/*.npz|json

The fixtures are what tests/test_oracle_golden.py re-checks without the reference being present.
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch

from . import laws_ref, losses_ref, steps_ref, unet_ref

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    # utils/utils.py imports two absent third-party names at module top that the hot path never calls
    if "easydict" not in sys.modules:
        m = types.ModuleType("easydict")
        m.EasyDict = dict
        sys.modules["easydict"] = m
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tr = types.ModuleType("torchvision.transforms")
        fn = types.ModuleType("torchvision.transforms.functional")
        fn.normalize = lambda *a, **k: None
        tv.transforms = tr
        tr.functional = fn
        sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "torchvision.transforms.functional": fn})
    R = types.SimpleNamespace()
    R.unet = _load("ref_unet", "model/unet.py")
    R.dice = _load("ref_diceloss", "utils/loss/diceloss.py")
    R.med = _load("ref_medloss", "utils/loss/medloss.py")
    R.dense = _load("ref_dense", "utils/loss/dense_loss.py")
    R.medlr = _load("ref_medlr", "utils/scheduler/medical_lr.py")
    R.coslr = _load("ref_coslr", "utils/scheduler/warmup_cosine.py")
    R.utils = _load("ref_utils", "utils/utils.py")
    return R


def synth_batch(seed, n, h, w, in_ch=1, ncls=4, cell=8):
    """Spatially coherent labels + noisy image (same law as hpfg_amd.datasets.synthetic)."""
    g = torch.Generator().manual_seed(seed)
    lab = torch.randint(0, ncls, (n, h // cell, w // cell), generator=g)
    lab = lab.repeat_interleave(cell, 1).repeat_interleave(cell, 2)
    img = lab.to(torch.float32).unsqueeze(1) / max(ncls - 1, 1) + 0.1 * torch.randn(n, 1, h, w, generator=g)
    return img.expand(n, in_ch, h, w).contiguous(), lab.to(torch.uint8)


def pack(mask):
    return np.packbits(mask.numpy().astype(np.uint8).ravel())


def state_np(sd):
    return {k: v.detach().numpy().copy() for k, v in sd.items()}


def close(a, b, tol, what):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    assert err <= tol, f"{what}: max abs err {err:.3e} > {tol:.1e}"
    return err


def grads_summary(named_grads):
    return {k: np.array([float(g.sum()), float(g.abs().sum()), float(g.abs().max())], dtype=np.float64) for k, g in named_grads.items()}


def main():
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    R = load_reference()
    report = OrderedDict()

    # ---- 1. initial state by seed -------------------------------------------------------------
    for plus, ctor in ((False, R.unet.UNet), (True, R.unet.UNet_Plus)):
        torch.manual_seed(1)
        net = ctor(1, 4)
        st = unet_ref.init_state(1, 1, 4, plus)
        sd = net.state_dict()
        assert list(sd.keys()) == list(st.keys()), "state_dict key order"
        for k in sd:
            assert torch.equal(sd[k], st[k]), k
    torch.manual_seed(1337)
    net = R.unet.UNet(3, 2)
    st = unet_ref.init_state(1337, 3, 2)
    assert all(torch.equal(net.state_dict()[k], st[k]) for k in st)
    report["init_state"] = "bit-exact for UNet(1,4), UNet_Plus(1,4), UNet(3,2)"

    # ---- 2. survey anchor (eval forward of UNet(1,4), seed 1) ------------------------------------
    torch.manual_seed(1)
    net = R.unet.UNet(1, 4)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(8, 1, 224, 224, generator=g)
    t = torch.randint(0, 4, (8, 224, 224), generator=g)
    net.eval()
    with torch.no_grad():
        y = net(x)
        l = R.med.Med_Sup_Loss(4)(y, t)
    st = unet_ref.init_state(1, 1, 4)
    with torch.no_grad():
        yo = unet_ref.unet_forward(st, x, train=False)
    report["eval_forward_max_err"] = close(y, yo, 1e-5, "eval forward")
    anchors = {
        "param_count": int(sum(p.numel() for p in net.parameters())),
        "param_count_plus": int(sum(p.numel() for p in R.unet.UNet_Plus(1, 4).parameters())),
        "eval_logits_sum": float(y.double().sum()), "eval_logits_meanabs": float(y.abs().mean()),
        "eval_logits_px00": [float(v) for v in y[0, :, 0, 0]], "eval_med_sup_loss": float(l),
    }

    # ---- 3. train-mode forward/backward, dropout drawn from torch's generator -------------------
    fx = {}
    for tag, (n, hw, in_ch, ncls, seed) in {"a": (2, 32, 1, 4, 1), "b": (3, 48, 3, 2, 1337)}.items():
        torch.manual_seed(seed)
        net = (R.unet.UNet(in_ch, ncls))
        x, lab = synth_batch(100 + seed, n, hw, hw, in_ch, ncls)
        net.train()
        torch.manual_seed(7)
        out = net(x)
        loss = R.med.Med_Sup_Loss(ncls)(out, lab.long())
        loss.backward()
        gref = {k: p.grad.clone() for k, p in net.named_parameters()}
        sd_after = net.state_dict()
        # oracle with the same generator state
        st = unet_ref.init_state(seed, in_ch, ncls)
        torch.manual_seed(7)
        masks = unet_ref.draw_dropout_masks(n, hw, hw)
        names = steps_ref._train_state(st)
        taps = {}
        oo = unet_ref.unet_forward(st, x, True, masks, taps=taps)
        lo = losses_ref.med_sup_loss(oo, lab.long())
        go = steps_ref._grads(lo, st, names)
        steps_ref._detach_state(st)
        e1 = close(out, oo, 1e-5, f"train fwd {tag}")
        e2 = close(loss, lo, 1e-6, f"loss {tag}")
        e3 = max(close(gref[k], go[k], 2e-5 * max(1.0, float(gref[k].abs().max())), f"grad {k}") for k in gref)
        for k in sd_after:
            close(sd_after[k], st[k], 1e-6, f"buffers {k}")
        report[f"train_fwd_bwd_{tag}"] = {"logits": e1, "loss": e2, "grads": e3}
        fx[tag] = dict(
            meta=np.array([n, hw, in_ch, ncls, seed, 100 + seed], dtype=np.int64),
            x=x.numpy(), labels=lab.numpy(), logits=out.detach().numpy(), loss=np.float64(loss.item()),
            **{f"mask{i}": pack(m) for i, m in enumerate(masks)},
            **{f"grad_sum/{k}": v for k, v in grads_summary(gref).items()},
            **{f"grad/{k}": gref[k].numpy() for k in ("encoder.in_conv.conv_conv.0.weight", "encoder.down4.maxpool_conv.1.conv_conv.4.weight",
                                                     "decoder.up1.conv1x1.weight", "decoder.up4.conv.conv_conv.0.weight", "decoder.out_conv.weight",
                                                     "decoder.out_conv.bias", "encoder.down2.maxpool_conv.1.conv_conv.1.weight",
                                                     "encoder.down2.maxpool_conv.1.conv_conv.1.bias")},
            **{f"bn/{k}": v.numpy() for k, v in sd_after.items() if "running" in k},
            **{f"raw/{k}": v.detach().numpy() for k, v in taps.items() if k in ("encoder.in_conv.conv_conv.0", "encoder.down4.maxpool_conv.1.conv_conv.4", "decoder.up1.conv1x1")},
        )
        np.savez_compressed(os.path.join(OUT, f"unet_fwd_bwd_{tag}.npz"), **fx[tag])

    # ---- 4. losses ---------------------------------------------------------------------------
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(3, 4, 16, 16, generator=g)
    tl = torch.randn(3, 4, 16, 16, generator=g)
    lab = torch.randint(0, 4, (3, 16, 16), generator=g)
    lab[0, :2] = 255
    ref_dice = R.dice.DiceLoss(4)
    p = torch.softmax(logits, 1)
    v_dice = ref_dice(p, lab.unsqueeze(1))
    v_dice_float = ref_dice(p, lab.float().unsqueeze(1))
    v_med = R.med.Med_Sup_Loss(4)(logits, lab)
    v_ce = torch.nn.CrossEntropyLoss(ignore_index=255)(logits, lab)
    v_mse = torch.mean((p - torch.softmax(tl, 1)) ** 2)
    close(v_dice, losses_ref.dice_loss(p, lab.unsqueeze(1)), 1e-6, "dice")
    close(v_dice_float, losses_ref.dice_loss(p, lab.float()), 1e-6, "dice float")
    close(v_med, losses_ref.med_sup_loss(logits, lab), 1e-6, "med")
    close(v_ce, losses_ref.cross_entropy(logits, lab), 1e-6, "ce")
    hg = (torch.randn(6, 128, generator=g), torch.randn(6, 128, 16, generator=g))
    tg = (torch.randn(6, 128, generator=g), torch.randn(6, 128, 16, generator=g))
    v_dense = R.dense.Dense_Loss(6, torch.device("cpu"))(hg, tg)
    close(v_dense, losses_ref.dense_loss(hg, tg), 1e-5, "dense")
    # gradient of the composite loss w.r.t. logits (what seg_loss_bwd must reproduce)
    lg = logits.clone().requires_grad_(True)
    comp = R.med.Med_Sup_Loss(4)(lg[:1], lab[:1]) + 0.3 * torch.mean((torch.softmax(lg[1:], 1) - torch.softmax(tl[1:], 1)) ** 2)
    comp.backward()
    np.savez_compressed(os.path.join(OUT, "losses.npz"), logits=logits.numpy(), t_logits=tl.numpy(), labels=lab.numpy().astype(np.int64),
                        dice=np.float64(v_dice), dice_float=np.float64(v_dice_float), med=np.float64(v_med), ce=np.float64(v_ce), mse=np.float64(v_mse),
                        dense=np.float64(v_dense), hg0=hg[0].numpy(), hg1=hg[1].numpy(), tg0=tg[0].numpy(), tg1=tg[1].numpy(),
                        comp=np.float64(comp.item()), comp_dlogits=lg.grad.numpy())
    report["losses"] = "dice/dice(float target)/med_sup/ce/dense within 1e-5"

    # ---- 5. host laws ---------------------------------------------------------------------------
    dummy = torch.nn.Linear(1, 1)
    opt = torch.optim.SGD(dummy.parameters(), lr=0.01, momentum=0.9)
    sch = R.medlr.Medical_LR(opt, 0.01, 30000)
    med = []
    for _ in range(6):
        med.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    opt = torch.optim.SGD(dummy.parameters(), lr=0.01, momentum=0.9)
    sch = R.coslr.CosineWarmupLR_Scheduler(opt, warmup_epochs=0, warmup_lr=1e-4, num_epochs=30000 // 200, base_lr=0.01, final_lr=1e-6, iter_per_epoch=200)
    cos = []
    for _ in range(6):
        cos.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    table = laws_ref.cosine_table(0.01, 0, 1e-4, 1e-6, 200, 150)
    for k in range(1, 7):
        close(med[k - 1], laws_ref.medical_lr(k, 0.01, 30000), 1e-12, "medical lr")
        close(cos[k - 1], laws_ref.cosine_lr(k, table), 1e-12, "cosine lr")
    ramps = {"sigmoid": [R.utils.sigmoid_rampup(e, 200.0) for e in (0, 1, 50, 199, 200, 500)],
             "linear": [R.utils.linear_rampup(e, 200.0) for e in (0, 1, 50, 199, 200, 500)]}
    for i, e in enumerate((0, 1, 50, 199, 200, 500)):
        close(ramps["sigmoid"][i], laws_ref.sigmoid_rampup(e, 200.0), 1e-12, "sigmoid")
        close(ramps["linear"][i], laws_ref.linear_rampup(e, 200.0), 1e-12, "linear")
    gen = R.utils.BoxMaskGenerator(prop_range=(0.25, 0.5), n_boxes=4, random_aspect_ratio=True, prop_by_area=True, within_bounds=True, invert=True)
    np.random.seed(1)
    mref = gen.generate_params(5, (64, 64))
    mo = laws_ref.box_masks(5, (64, 64), np.random.RandomState(1))
    np.random.seed(1)
    mo2 = laws_ref.box_masks(5, (64, 64), np.random)
    assert np.array_equal(mref, mo2) and np.array_equal(mref, mo), "box masks"
    anchors.update({"medical_lr_first6": med, "cosine_lr_first6": cos, "rampup_epochs": [0, 1, 50, 199, 200, 500], **{f"rampup_{k}": v for k, v in ramps.items()},
                    "ema_alpha_first4": [min(1 - 1 / (s + 1), 0.99) for s in (1, 2, 3, 200)],
                    "box_mask_seed1_n5_64_sum": float(mref.sum()), "box_mask_seed1_rowsums": [float(v) for v in mref.reshape(5, -1).sum(1)]})
    np.savez_compressed(os.path.join(OUT, "box_masks.npz"), masks=np.packbits(mref.astype(np.uint8)), shape=np.array(mref.shape))
    report["laws"] = "medical/cosine lr, ramp-ups, box masks exact"

    # ---- 6. step traces (reference modules + torch.optim + reference schedulers) ----------------
    traces = {}
    # 6a supervised, cfg1 law at reduced size: UNet(1,4), N=4, 32x32, cosine schedule, 4 iterations
    torch.manual_seed(1)
    net = R.unet.UNet(1, 4)
    net.train()
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
    sch = R.coslr.CosineWarmupLR_Scheduler(opt, warmup_epochs=0, warmup_lr=1e-4, num_epochs=150, base_lr=0.01, final_lr=1e-6, iter_per_epoch=200)
    crit = R.med.Med_Sup_Loss(4)
    x, lab = synth_batch(11, 4, 32, 32)
    st = unet_ref.init_state(1, 1, 4)
    bufs = {}
    ref_losses, or_losses, all_masks = [], [], []
    for k in range(1, 5):
        torch.manual_seed(1000 + k)
        out = net(x)
        loss = crit(out, lab.long())
        opt.zero_grad()
        loss.backward()
        opt.step()
        sch.step()
        ref_losses.append(loss.item())
        torch.manual_seed(1000 + k)
        masks = unet_ref.draw_dropout_masks(4, 32, 32)
        all_masks.append(masks)
        r = steps_ref.supervised_step(st, bufs, x, lab.long(), laws_ref.cosine_lr(k, table), 0.9, 5e-4, masks)
        or_losses.append(r["loss"])
    close(ref_losses, or_losses, 2e-5, "sup trace")
    net.eval()
    with torch.no_grad():
        fin = net(x)
        fo = unet_ref.unet_forward(st, x, train=False)
    close(fin, fo, 2e-4, "sup final logits")
    dice_ref = losses_ref.mean_foreground_dice(fin.argmax(1).numpy(), lab.numpy(), 4)
    traces["sup"] = dict(x=x.numpy(), labels=lab.numpy(), losses=np.array(ref_losses), final_eval_logits=fin.numpy(), final_dice=np.float64(dice_ref),
                         **{f"it{k}_mask{i}": pack(m) for k, ms in enumerate(all_masks) for i, m in enumerate(ms)})
    report["sup_trace_err"] = float(np.abs(np.array(ref_losses) - np.array(or_losses)).max())

    # 6b mean teacher, N=2+2, 32x32, 3 iterations (cur_itrs 1..3 => weight 0; also run with forced weight)
    torch.manual_seed(1337)
    net = R.unet.UNet(1, 4)
    import copy
    ema = copy.deepcopy(net)
    for p_ in ema.parameters():
        p_.requires_grad = False
    net.train()
    ema.train()
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    sch = R.medlr.Medical_LR(opt, 0.01, 30000)
    xl, yl = synth_batch(21, 2, 32, 32)
    xu, _ = synth_batch(22, 2, 32, 32)
    st = unet_ref.init_state(1337, 1, 4)
    est = unet_ref.clone_state(st)
    bufs = {}
    rl, ol, mm = [], [], []
    cons_w = 0.1 * R.utils.sigmoid_rampup(40, 200.0)   # as if cur_itrs//150 == 40
    for k in range(1, 4):
        xx = torch.cat([xl, xu], 0)
        torch.manual_seed(2000 + k)
        out = net(xx)
        soft = torch.softmax(out, 1)
        with torch.no_grad():
            eo = ema(xx)
            es = torch.softmax(eo, 1)
        sup = crit(out[:2], yl.long())
        cons = torch.mean((soft[2:] - es[2:]) ** 2)
        loss = sup + cons_w * cons
        opt.zero_grad()
        loss.backward()
        opt.step()
        sch.step()
        R.utils.update_ema_variables(net, ema, 0.99, k)
        rl.append([loss.item(), sup.item(), cons.item()])
        torch.manual_seed(2000 + k)
        ms = unet_ref.draw_dropout_masks(4, 32, 32)
        mt = unet_ref.draw_dropout_masks(4, 32, 32)
        mm.append((ms, mt))
        r = steps_ref.mean_teacher_step(st, est, bufs, xl, yl.long(), xu, laws_ref.medical_lr(k, 0.01, 30000), cons_w, laws_ref.ema_alpha(k, 0.99), 0.9, 1e-4, ms, mt)
        ol.append([r["loss"], r["sup"], r["cons"]])
    close(rl, ol, 2e-5, "mt trace")
    for k_, v_ in ema.state_dict().items():
        close(v_, est[k_], 1e-5, f"mt ema {k_}")
    traces["mt"] = dict(xl=xl.numpy(), yl=yl.numpy(), xu=xu.numpy(), cons_w=np.float64(cons_w), losses=np.array(rl),
                        student_logits_last=out.detach().numpy(), teacher_logits_last=eo.numpy(),
                        **{f"it{k}_{w}{i}": pack(m) for k, (ms, mt) in enumerate(mm) for w, mlist in (("s", ms), ("t", mt)) for i, m in enumerate(mlist)})
    report["mt_trace_err"] = float(np.abs(np.array(rl) - np.array(ol)).max())

    # 6c CPS, UNet(3,2) x2, N=2+2, 48x48, 2 iterations
    torch.manual_seed(1337)
    n1 = R.unet.UNet(3, 2)
    n2 = R.unet.UNet(3, 2)
    n1.train()
    n2.train()
    o1 = torch.optim.SGD(n1.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    o2 = torch.optim.SGD(n2.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    s1 = R.medlr.Medical_LR(o1, 0.01, 30000)
    s2 = R.medlr.Medical_LR(o2, 0.01, 30000)
    crit2 = R.med.Med_Sup_Loss(2)
    xl, yl = synth_batch(31, 2, 48, 48, 3, 2)
    xu, _ = synth_batch(32, 2, 48, 48, 3, 2)
    torch.manual_seed(1337)
    sa = unet_ref.init_state(None, 3, 2)
    sb = unet_ref.init_state(None, 3, 2)
    ba, bb = {}, {}
    rl, ol, mm = [], [], []
    cw = 0.1 * R.utils.sigmoid_rampup(60, 200.0)
    for k in range(1, 3):
        xx = torch.cat([xl, xu], 0)
        torch.manual_seed(3000 + k)
        a = n1(xx)
        b = n2(xx)
        sup = crit2(a[:2], yl.long()) + crit2(b[:2], yl.long())
        pa = torch.argmax(torch.softmax(a[2:], 1).detach(), 1)
        pb = torch.argmax(torch.softmax(b[2:], 1).detach(), 1)
        semi = crit2(a[2:], pb) + crit2(b[2:], pa)
        loss = sup + cw * semi
        o1.zero_grad()
        o2.zero_grad()
        loss.backward()
        o1.step()
        o2.step()
        s1.step()
        s2.step()
        rl.append([loss.item(), sup.item(), semi.item()])
        torch.manual_seed(3000 + k)
        m1 = unet_ref.draw_dropout_masks(4, 48, 48)
        m2 = unet_ref.draw_dropout_masks(4, 48, 48)
        mm.append((m1, m2))
        lr = laws_ref.medical_lr(k, 0.01, 30000)
        r = steps_ref.cps_step(sa, sb, ba, bb, xl, yl.long(), xu, lr, lr, cw, 0.9, 1e-4, m1, m2)
        ol.append([r["loss"], r["sup"], r["semi"]])
    close(rl, ol, 5e-5, "cps trace")
    traces["cps"] = dict(xl=xl.numpy(), yl=yl.numpy(), xu=xu.numpy(), cons_w=np.float64(cw), losses=np.array(rl),
                         logits1_last=a.detach().numpy(), logits2_last=b.detach().numpy(),
                         **{f"it{k}_{w}{i}": pack(m) for k, (m1, m2) in enumerate(mm) for w, mlist in (("a", m1), ("b", m2)) for i, m in enumerate(mlist)})
    report["cps_trace_err"] = float(np.abs(np.array(rl) - np.array(ol)).max())

    # 6d HPFG, UNet_Plus x3, N=2+2, 64x64, iterations at cur_itrs = 1499, 1500 region (weights active, MSE on)
    torch.manual_seed(1)
    m1 = R.unet.UNet_Plus(1, 4)
    m2 = R.unet.UNet_Plus(1, 4)
    em = copy.deepcopy(m2)
    for p_ in em.parameters():
        p_.requires_grad = False
    m1.train()
    m2.train()
    o1 = torch.optim.SGD(m1.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
    o2 = torch.optim.SGD(m2.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
    torch.manual_seed(1)
    sa = unet_ref.init_state(None, 1, 4, True)
    sb = unet_ref.init_state(None, 1, 4, True)
    se = unet_ref.clone_state(sb)
    ba, bb = {}, {}
    dense = R.dense.Dense_Loss(4, torch.device("cpu"))
    ce = torch.nn.CrossEntropyLoss(ignore_index=255)
    dl = R.dice.DiceLoss(4)
    xl, yl = synth_batch(41, 2, 64, 64)
    xl1, yl1 = synth_batch(42, 2, 64, 64)
    xu, _ = synth_batch(43, 2, 64, 64)
    rng = np.random.RandomState(1)
    rl, ol, mm, cms = [], [], [], []
    for j, cur in enumerate((1499, 30000)):
        cm = torch.tensor(gen.generate_params(2, (64, 64), rng=rng), dtype=torch.float)
        cms.append(cm)
        mix = torch.cat([xl, xl1 * (1.0 - cm) + xu * cm], 0)
        torch.manual_seed(4000 + j)
        a, _, _ = m1(mix)
        sa_ = torch.softmax(a, 1)
        vol = torch.cat([xl, xu], 0)
        b, h1, h2 = m2(vol)
        sb_ = torch.softmax(b, 1)
        with torch.no_grad():
            eo, eh1, eh2 = em(vol)
            es = torch.softmax(eo.detach(), 1)
        l1 = 0.5 * (ce(a[:2], yl.long()) + dl(sa_[:2], yl.long().unsqueeze(1)))
        l2 = 0.5 * (ce(b[:2], yl.long()) + dl(sb_[:2], yl.long().unsqueeze(1)))
        sup = l1 + l2
        con = dense(h1, eh1) + dense(h2, eh2)
        c2 = cm.squeeze(1)
        pseudo = yl1.long() * (1.0 - c2) + torch.argmax(es[2:], 1) * c2
        ps = dl(sa_[2:], pseudo.unsqueeze(1))
        w = 0.1 * R.utils.linear_rampup(cur // 150, 200.0)
        cons2 = 0.0 if cur < 1000 else torch.mean((sb_[2:] - es[2:]) ** 2)
        semi = 7 * w * ps + w * 0.0 + w * cons2 + w * con
        loss = sup + semi
        o1.zero_grad()
        o2.zero_grad()
        loss.backward()
        o1.step()
        o2.step()
        alpha = min(1 - 1 / (cur + 1), 0.99)
        with torch.no_grad():
            for part in ("encoder", "decoder"):
                for pe, pm in zip(getattr(m2, part).parameters(), getattr(m1, part).parameters()):
                    pe.data.mul_(alpha).add_(pm.data, alpha=1 - alpha)
        R.utils.update_ema_variables(m2, em, 0.99, cur)
        rl.append([loss.item(), sup.item(), float(semi), ps.item(), con.item()])
        torch.manual_seed(4000 + j)
        ma = unet_ref.draw_dropout_masks(4, 64, 64)
        mb = unet_ref.draw_dropout_masks(4, 64, 64)
        mt = unet_ref.draw_dropout_masks(4, 64, 64)
        mm.append((ma, mb, mt))
        r = steps_ref.hpfg_step(sa, sb, se, ba, bb, xl, yl.long(), xl1, yl1.long(), xu, cm, cur, 0.01, 0.01, 0.1, 200.0, 0.99, 0.9, 5e-4, ma, mb, mt)
        ol.append([r["loss"], r["sup"], r["semi"], r["pseudo_sup"], r["contrast"]])
    close(rl, ol, 1e-4, "hpfg trace")
    for k_, v_ in em.state_dict().items():
        close(v_, se[k_], 1e-5, f"hpfg ema {k_}")
    for k_, v_ in m2.state_dict().items():
        close(v_, sb[k_], 1e-5, f"hpfg m2 {k_}")
    traces["hpfg"] = dict(xl=xl.numpy(), yl=yl.numpy(), xl1=xl1.numpy(), yl1=yl1.numpy(), xu=xu.numpy(), cur_itrs=np.array([1499, 30000]),
                          cutmix=np.stack([c.numpy() for c in cms]), losses=np.array(rl), logits1_last=a.detach().numpy(), logits2_last=b.detach().numpy(),
                          t_logits_last=eo.numpy(),
                          **{f"it{k}_{w}{i}": pack(m) for k, trip in enumerate(mm) for w, mlist in zip("abt", trip) for i, m in enumerate(mlist)})
    report["hpfg_trace_err"] = float(np.abs(np.array(rl) - np.array(ol)).max())

    for name, d in traces.items():
        np.savez_compressed(os.path.join(OUT, f"trace_{name}.npz"), **d)
    with open(os.path.join(OUT, "anchors.json"), "w") as f:
        json.dump(anchors, f, indent=1)
    with open(os.path.join(OUT, "pinning_report.json"), "w") as f:
        json.dump({"torch": torch.__version__, "reference": "fakerlove1/HPFG @ /root/reference", "checks": report}, f, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
