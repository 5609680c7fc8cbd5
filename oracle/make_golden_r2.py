"""Oracle pinning, second set (test infrastructure; runs ONLY in the build container, never on the GPU box).

Drives further pieces of the REFERENCE (imported file by file from /root/reference, as oracle/make_golden.py does), checks the oracle
restatements against them and writes the reference's outputs as fixtures under tests/golden/:

  augment.npz       datasets/utils.py:73-117  RandomGenerator (rot90+flip | ndimage.rotate(order=0) -> zoom(order=0)) on seeded slices,
                    seeded `random` + `np.random`                                  -> pins oracle/augment_ref.py
  trace_hpfg2.npz   main.py:125-212 with the branches the first HPFG trace misses: Nu//Nl = 3 label repeat (:142-143, batch 2+6 like the
                    reference YAML's 8+24), cur_itrs = 999 / 1000 / 1001 across the `cur_itrs < 1000` gate (:186-188), both Medical_LR
                    schedulers stepped every iteration (:211-212)
  trace_s4cvnet.npz 2022_08_CVPR_S4CVNet_ACDC.py:107-167 (two U-Net students + EMA teacher on noisy unlabelled input), iterations 999..1001
  trace_sup224.npz  sup_ACDC.py:83-93, BASELINE configs[0] as written: UNet(1,4), 8 synthetic 224x224 slices, SGD + cosine schedule, 10
                    iterations: per-iteration loss, final eval logits (sub-sampled + checksums) and mean foreground Dice.  Inputs and
                    dropout masks are regenerated from seeds (15 MB of masks otherwise); their checksums are stored so that a different
                    CPU generator stream fails loudly.

    python -m oracle.make_golden_r2
"""
from __future__ import annotations

import copy
import os
import random

import numpy as np
import torch

from . import augment_ref, laws_ref, losses_ref, steps_ref, unet_ref
from .make_golden import OUT, _load, close, load_reference, pack, synth_batch


def augment_fixture(report):
    R = load_reference()
    import sys
    tv = sys.modules["torchvision"]
    if not hasattr(tv.transforms, "ToTensor"):          # names datasets/utils.py touches only inside color_jitter (never called here)
        tv.transforms.ToTensor = tv.transforms.ColorJitter = object
    du = _load("ref_dataset_utils", "datasets/utils.py")
    g = np.random.RandomState(7)
    shapes = [(216, 256), (232, 200), (256, 256), (174, 208), (224, 224), (154, 187)]
    slices = []
    for h, w in shapes:
        lab = g.randint(0, 4, (h // 8 + 1, w // 8 + 1)).repeat(8, 0).repeat(8, 1)[:h, :w].astype(np.uint8)
        img = (lab / 3.0 + 0.1 * g.randn(h, w)).astype(np.float32)
        slices.append((img, lab))
    gen = du.RandomGenerator((224, 224))
    out = {"n": np.int64(len(slices)), "seeds": [], "branch": []}
    k = 0
    for rep in range(4):                                  # every slice under four seeds: all three branches occur
        for i, (img, lab) in enumerate(slices):
            seed = 100 * rep + i
            random.seed(seed)
            np.random.seed(seed)
            s = gen(img, lab)
            py, nr = random.Random(seed), np.random.RandomState(seed)
            oi, ol = augment_ref.random_generator(img, lab, (224, 224), py, nr)
            assert np.array_equal(s["image"].numpy(), oi) and np.array_equal(s["mask"].numpy(), ol), (rep, i)
            py = random.Random(seed)
            a = py.random()
            out["branch"].append(0 if a > 0.5 else (1 if py.random() > 0.5 else 2))
            out["seeds"].append(seed)
            out[f"img{k}"] = s["image"].numpy().astype(np.float16)       # values are label/3 + noise: compared after the same cast
            out[f"img{k}_sum"] = np.float64(s["image"].double().sum())
            out[f"lab{k}"] = np.packbits(np.unpackbits(s["mask"].numpy().reshape(-1, 1), axis=1)[:, 6:].reshape(-1))     # 2 bits per label
            k += 1
    for i, (img, lab) in enumerate(slices):
        out[f"src_img{i}"], out[f"src_lab{i}"] = img, lab
    out["seeds"], out["branch"] = np.array(out["seeds"]), np.array(out["branch"])
    assert set(out["branch"].tolist()) == {0, 1, 2}
    np.savez_compressed(os.path.join(OUT, "augment.npz"), **out)
    report["augment"] = f"RandomGenerator bit-equal on {k} (slice, seed) cases; branches {np.bincount(out['branch']).tolist()}"


def hpfg2_trace(report):
    R = load_reference()
    gen = R.utils.BoxMaskGenerator(prop_range=(0.25, 0.5), n_boxes=4, random_aspect_ratio=True, prop_by_area=True, within_bounds=True, invert=True)
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
    s1, s2 = R.medlr.Medical_LR(o1, 0.01, 30000), R.medlr.Medical_LR(o2, 0.01, 30000)
    first = 999
    for _ in range(first - 1):                           # the schedulers as they stand when iteration 999 begins
        o1.step()
        o2.step()
        s1.step()
        s2.step()
    torch.manual_seed(1)
    sa = unet_ref.init_state(None, 1, 4, True)
    sb = unet_ref.init_state(None, 1, 4, True)
    se = unet_ref.clone_state(sb)
    ba, bb = {}, {}
    NL, NU, HW = 2, 6, 64
    dense = R.dense.Dense_Loss(NL + NU, torch.device("cpu"))              # main.py:89: batch_size + unlabel_batch_size
    ce = torch.nn.CrossEntropyLoss(ignore_index=255)
    dl = R.dice.DiceLoss(4)
    xl, yl = synth_batch(51, NL, HW, HW)
    xl1_, yl1_ = synth_batch(52, NL, HW, HW)
    xu, _ = synth_batch(53, NU, HW, HW)
    rng = np.random.RandomState(3)
    rl, ol, mm, cms, lrs = [], [], [], [], []
    c1 = 0.0
    for j, cur in enumerate((999, 1000, 1001)):
        rep = NU // NL
        xl1 = xl1_.repeat(rep, 1, 1, 1)
        yl1 = yl1_.repeat(rep, 1, 1).long()
        cm = torch.tensor(gen.generate_params(NU, (HW, HW), rng=rng), dtype=torch.float)
        cms.append(cm)
        mix = torch.cat([xl, xl1 * (1.0 - cm) + xu * cm], 0)
        torch.manual_seed(6000 + j)
        a, _, _ = m1(mix)
        sa_ = torch.softmax(a, 1)
        vol = torch.cat([xl, xu], 0)
        b, h1, h2 = m2(vol)
        sb_ = torch.softmax(b, 1)
        with torch.no_grad():
            eo, eh1, eh2 = em(vol)
            es = torch.softmax(eo.detach(), 1)
        l1 = 0.5 * (ce(a[:NL], yl.long()) + dl(sa_[:NL], yl.long().unsqueeze(1)))
        l2 = 0.5 * (ce(b[:NL], yl.long()) + dl(sb_[:NL], yl.long().unsqueeze(1)))
        sup = l1 + l2
        con = dense(h1, eh1) + dense(h2, eh2)
        c2 = cm.squeeze(1)
        pseudo = yl1 * (1.0 - c2) + torch.argmax(es[NL:], 1) * c2
        ps = dl(sa_[NL:], pseudo.unsqueeze(1))
        w = 0.1 * R.utils.linear_rampup(cur // 150, 200.0)
        if cur < 1000:
            c1, cons2 = 0.0, 0.0
        else:
            cons2 = torch.mean((sb_[NL:] - es[NL:]) ** 2)
        semi = 7 * w * ps + w * c1 + w * cons2 + w * con
        loss = sup + semi
        lrs.append([o1.param_groups[0]["lr"], o2.param_groups[0]["lr"]])
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
        s1.step()
        s2.step()
        rl.append([loss.item(), sup.item(), float(semi), ps.item(), con.item(), float(cons2)])
        torch.manual_seed(6000 + j)
        ma = unet_ref.draw_dropout_masks(NL + NU, HW, HW)
        mb = unet_ref.draw_dropout_masks(NL + NU, HW, HW)
        mt = unet_ref.draw_dropout_masks(NL + NU, HW, HW)
        mm.append((ma, mb, mt))
        lr = laws_ref.medical_lr(cur, 0.01, 30000)
        assert abs(lr - lrs[-1][0]) < 1e-12, (lr, lrs[-1])
        r = steps_ref.hpfg_step(sa, sb, se, ba, bb, xl, yl.long(), xl1, yl1, xu, cm, cur, lr, lr, 0.1, 200.0, 0.99, 0.9, 5e-4, ma, mb, mt)
        ol.append([r["loss"], r["sup"], r["semi"], r["pseudo_sup"], r["contrast"]])
    close(np.array(rl)[:, :5], ol, 1e-4, "hpfg2 trace")
    assert rl[0][5] == 0.0 and rl[1][5] > 0.0
    for k_, v_ in em.state_dict().items():
        close(v_, se[k_], 1e-5, f"hpfg2 ema {k_}")
    d = dict(xl=xl.numpy(), yl=yl.numpy(), xl1=xl1_.numpy(), yl1=yl1_.numpy(), xu=xu.numpy(), cur_itrs=np.array([999, 1000, 1001]), lrs=np.array(lrs),
             cutmix=np.stack([c.numpy() for c in cms]), losses=np.array(rl), logits1_last=a.detach().numpy(), logits2_last=b.detach().numpy(),
             t_logits_last=eo.numpy(),
             **{f"it{k}_{w}{i}": pack(m) for k, trip in enumerate(mm) for w, mlist in zip("abt", trip) for i, m in enumerate(mlist)})
    np.savez_compressed(os.path.join(OUT, "trace_hpfg2.npz"), **d)
    report["hpfg2_trace_err"] = float(np.abs(np.array(rl)[:, :5] - np.array(ol)).max())


def sup224_trace(report):
    R = load_reference()
    torch.manual_seed(1)
    net = R.unet.UNet(1, 4)
    net.train()
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
    sch = R.coslr.CosineWarmupLR_Scheduler(opt, warmup_epochs=0, warmup_lr=1e-4, num_epochs=150, base_lr=0.01, final_lr=1e-6, iter_per_epoch=200)
    crit = R.med.Med_Sup_Loss(4)
    N, HW, ITERS = 8, 224, 10
    x, lab = synth_batch(1234, N, HW, HW, 1, 4, 32)
    st = unet_ref.init_state(1, 1, 4)
    bufs = {}
    table = laws_ref.cosine_table(0.01, 0, 1e-4, 1e-6, 200, 150)
    ref_losses, or_losses, msums = [], [], []
    for k in range(1, ITERS + 1):
        torch.manual_seed(7000 + k)
        out = net(x)
        loss = crit(out, lab.long())
        opt.zero_grad()
        loss.backward()
        opt.step()
        sch.step()
        ref_losses.append(loss.item())
        torch.manual_seed(7000 + k)
        masks = unet_ref.draw_dropout_masks(N, HW, HW)
        msums.append([float(m.sum()) for m in masks])
        r = steps_ref.supervised_step(st, bufs, x, lab.long(), laws_ref.cosine_lr(k, table), 0.9, 5e-4, masks)
        or_losses.append(r["loss"])
    close(ref_losses, or_losses, 5e-5, "sup224 trace")
    net.eval()
    with torch.no_grad():
        fin = net(x)
        fo = unet_ref.unet_forward(st, x, train=False)
    close(fin, fo, 5e-4, "sup224 final logits")
    dice = losses_ref.mean_foreground_dice(fin.argmax(1).numpy(), lab.numpy(), 4)
    d = dict(meta=np.array([N, HW, ITERS, 1234, 7000]), x_sum=np.float64(x.double().sum()), lab_sum=np.int64(lab.long().sum()), mask_sums=np.array(msums),
             losses=np.array(ref_losses), final_eval_logits_sub=fin[:, :, ::8, ::8].numpy(), final_eval_logits_sum=np.float64(fin.double().sum()),
             final_eval_logits_abs=np.float64(fin.double().abs().sum()), final_pred=np.packbits(np.unpackbits(
                 fin.argmax(1).numpy().astype(np.uint8).reshape(-1, 1), axis=1)[:, 6:].reshape(-1)), final_dice=np.float64(dice))
    np.savez_compressed(os.path.join(OUT, "trace_sup224.npz"), **d)
    report["sup224_trace_err"] = float(np.abs(np.array(ref_losses) - np.array(or_losses)).max())
    report["sup224_losses"] = [round(v, 5) for v in ref_losses]
    report["sup224_dice"] = float(dice)


def s4cvnet_trace(report):
    """2022_08_CVPR_S4CVNet_ACDC.py:107-167 with U-Nets as both students (the reference's s4cvnet_unet YAML), batch 2+4, 64x64,
    iterations 999 / 1000 / 1001 (the consistency gate opens at 1000)."""
    R = load_reference()
    torch.manual_seed(1337)
    m1 = R.unet.UNet(1, 4)
    m2 = R.unet.UNet(1, 4)
    em = copy.deepcopy(m2)
    for p_ in em.parameters():
        p_.requires_grad = False
    m1.train()
    m2.train()
    o1 = torch.optim.SGD(m1.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
    o2 = torch.optim.SGD(m2.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
    s1, s2 = R.medlr.Medical_LR(o1, 0.01, 30000), R.medlr.Medical_LR(o2, 0.01, 30000)
    for _ in range(998):
        o1.step()
        o2.step()
        s1.step()
        s2.step()
    torch.manual_seed(1337)
    sa = unet_ref.init_state(None, 1, 4)
    sb = unet_ref.init_state(None, 1, 4)
    se = unet_ref.clone_state(sb)
    ba, bb = {}, {}
    NL, NU, HW = 2, 4, 64
    ce = torch.nn.CrossEntropyLoss(ignore_index=255)
    dl = R.dice.DiceLoss(4)
    xl, yl = synth_batch(61, NL, HW, HW)
    xu, _ = synth_batch(62, NU, HW, HW)
    yl = yl.long()
    rl, ol, mm, noises = [], [], [], []
    c1 = c2 = 0.0
    for j, cur in enumerate((999, 1000, 1001)):
        torch.manual_seed(8000 + j)
        nz = torch.randn_like(xu)
        noises.append(nz)
        ema_in = xu + torch.clamp(nz * 0.1, -0.2, 0.2)
        vol = torch.cat([xl, xu], 0)
        torch.manual_seed(8100 + j)
        a = m1(vol)
        sa_ = torch.softmax(a, 1)
        b = m2(vol)
        sb_ = torch.softmax(b, 1)
        with torch.no_grad():
            eo = em(ema_in)
            es = torch.softmax(eo, 1)
        l1 = 0.5 * (ce(a[:NL], yl) + dl(sa_[:NL], yl.unsqueeze(1)))
        l2 = 0.5 * (ce(b[:NL], yl) + dl(sb_[:NL], yl.unsqueeze(1)))
        sup = l1 + l2
        p1 = torch.argmax(sa_[NL:].detach(), 1)
        p2 = torch.argmax(sb_[NL:].detach(), 1)
        ps1 = dl(sa_[NL:], p2.unsqueeze(1))
        ps2 = dl(sb_[NL:], p1.unsqueeze(1))
        w = 0.1 * R.utils.linear_rampup(cur // 150, 200.0)
        if cur < 1000:
            c1 = c2 = 0.0
        else:
            c1 = torch.mean((sa_[NL:] - es) ** 2)
            c2 = torch.mean((sb_[NL:] - es) ** 2)
        semi = (7 * w * ps1 + w * c1) + (7 * w * ps2 + w * c2)
        loss = sup + semi
        o1.zero_grad()
        o2.zero_grad()
        loss.backward()
        o1.step()
        o2.step()
        R.utils.update_ema_variables(m2, em, 0.99, cur)
        s1.step()
        s2.step()
        rl.append([loss.item(), sup.item(), float(semi), ps1.item(), ps2.item(), float(c1), float(c2)])
        torch.manual_seed(8100 + j)
        ma = unet_ref.draw_dropout_masks(NL + NU, HW, HW)
        mb = unet_ref.draw_dropout_masks(NL + NU, HW, HW)
        mt = unet_ref.draw_dropout_masks(NU, HW, HW)
        mm.append((ma, mb, mt))
        lr = laws_ref.medical_lr(cur, 0.01, 30000)
        r = steps_ref.s4cvnet_step(sa, sb, se, ba, bb, xl, yl, xu, nz, cur, lr, lr, 0.1, 200.0, 0.99, 0.9, 5e-4, ma, mb, mt)
        ol.append([r["loss"], r["sup"], r["semi"], r["ps1"], r["ps2"], r["cons1"], r["cons2"]])
    close(rl, ol, 1e-4, "s4cvnet trace")
    for k_, v_ in em.state_dict().items():
        close(v_, se[k_], 1e-5, f"s4cvnet ema {k_}")
    d = dict(xl=xl.numpy(), yl=yl.numpy().astype(np.uint8), xu=xu.numpy(), noise=np.stack([n.numpy() for n in noises]), cur_itrs=np.array([999, 1000, 1001]),
             losses=np.array(rl), logits1_last=a.detach().numpy(), logits2_last=b.detach().numpy(), t_logits_last=eo.numpy(),
             **{f"it{k}_{w}{i}": pack(m) for k, trip in enumerate(mm) for w, mlist in zip("abt", trip) for i, m in enumerate(mlist)})
    np.savez_compressed(os.path.join(OUT, "trace_s4cvnet.npz"), **d)
    report["s4cvnet_trace_err"] = float(np.abs(np.array(rl) - np.array(ol)).max())


def main():
    torch.set_num_threads(8)
    report = {}
    augment_fixture(report)
    hpfg2_trace(report)
    sup224_trace(report)
    s4cvnet_trace(report)
    import json
    with open(os.path.join(OUT, "pinning_report_r2.json"), "w") as f:
        json.dump({"torch": torch.__version__, "reference": "fakerlove1/HPFG @ /root/reference", "checks": report}, f, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
