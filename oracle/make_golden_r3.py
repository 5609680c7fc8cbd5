"""Oracle pinning, third set (test infrastructure; runs ONLY in the build container, never on the GPU box).

Reference-module fixtures at sizes where BatchNorm is not degenerate (the first sets are 32..64 px with 4 images: 16..64 samples per
channel at the bottleneck, which amplifies any 1e-6 perturbation past 1e-3 within 2-3 optimizer steps).  Each trace drives the REFERENCE's
own modules (imported file by file from /root/reference like oracle/make_golden.py), checks the oracle restatement against them and
stores the reference's outputs:

  trace_mt224.npz    2017_03_NIPS_Mean-Teacher_ACDC.py:82-113, UNet(1,4) + EMA teacher, 2 + 2 images of 224 x 224, iterations 6001..6003
                     (consistency weight 0.1 * sigmoid_rampup(40, 200), Medical_LR stepped 6000 times before, EMA alpha 0.99)
  trace_cps96.npz    2021_06_CVPR_CPS_ACDC.py:95-120, UNet(3,2) x 2, 4 + 4 images of 96 x 96 (the LIDC configuration's shape), iterations 6001, 6002
  trace_hpfg224.npz  main.py:125-212, UNet_Plus x 2 + EMA teacher, 2 + 2 images of 224 x 224, iterations 1000, 1001 (MSE gate open)
  grads224.npz       one train-mode forward + backward of UNet(1,4) + Med_Sup_Loss on 4 images of 224 x 224: loss, logits and every parameter
                     gradient of the reference (norms + strided samples)

Inputs and dropout masks are regenerated from seeds by the tests (their checksums are stored, so that a different CPU generator stream
fails loudly); logits are stored sub-sampled (every 8th pixel) with their sum / absolute-sum checksums.

    python -m oracle.make_golden_r3
"""
from __future__ import annotations

import copy
import json
import os

import numpy as np
import torch

from . import laws_ref, losses_ref, steps_ref, unet_ref
from .make_golden import OUT, close, load_reference, synth_batch


def sub(t):
    t = t.detach()
    return dict(sub=t[:, :, ::8, ::8].numpy().copy(), sum=np.float64(t.double().sum()), abs=np.float64(t.double().abs().sum()))


def put(d, key, t):
    for k, v in sub(t).items():
        d[f"{key}_{k}"] = v


def mask_sums(ms):
    return [float(m.sum()) for m in ms]


def _spin(opts, scheds, n):
    """The schedulers as they stand after n iterations (an optimizer step without gradients changes nothing)."""
    for _ in range(n):
        for o in opts:
            o.step()
        for s in scheds:
            s.step()


def mt224(report):
    R = load_reference()
    NL, NU, HW, FIRST, ITERS = 2, 2, 224, 6001, 3
    torch.manual_seed(1337)
    net = R.unet.UNet(1, 4)
    ema = copy.deepcopy(net)
    for p_ in ema.parameters():
        p_.requires_grad = False
    net.train()
    ema.train()
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    sch = R.medlr.Medical_LR(opt, 0.01, 30000)
    _spin([opt], [sch], FIRST - 1)
    crit = R.med.Med_Sup_Loss(4)
    args = dict(consistency=0.1, consistency_rampup=200.0)
    xl, yl = synth_batch(121, NL, HW, HW, 1, 4, 32)
    xu, _ = synth_batch(122, NU, HW, HW, 1, 4, 32)
    st = unet_ref.init_state(1337, 1, 4)
    est = unet_ref.clone_state(st)
    bufs = {}
    rl, ol, msum, lrs = [], [], [], []
    for j in range(ITERS):
        cur = FIRST + j
        xx = torch.cat([xl, xu], 0)
        torch.manual_seed(12000 + j)
        out = net(xx)
        soft = torch.softmax(out, 1)
        with torch.no_grad():
            eo = ema(xx)
            es = torch.softmax(eo, 1)
        sup = crit(out[:NL], yl.long())
        cons = torch.mean((soft[NL:] - es[NL:]) ** 2)
        w = args["consistency"] * R.utils.sigmoid_rampup(cur // 150, args["consistency_rampup"])
        loss = sup + w * cons
        lrs.append(opt.param_groups[0]["lr"])
        opt.zero_grad()
        loss.backward()
        opt.step()
        sch.step()
        R.utils.update_ema_variables(net, ema, 0.99, cur)
        rl.append([loss.item(), sup.item(), cons.item()])
        torch.manual_seed(12000 + j)
        ms = unet_ref.draw_dropout_masks(NL + NU, HW, HW)
        mt = unet_ref.draw_dropout_masks(NL + NU, HW, HW)
        msum.append(mask_sums(ms) + mask_sums(mt))
        lr = laws_ref.medical_lr(cur, 0.01, 30000)
        assert abs(lr - lrs[-1]) < 1e-12, (lr, lrs[-1])
        r = steps_ref.mean_teacher_step(st, est, bufs, xl, yl.long(), xu, lr, w, laws_ref.ema_alpha(cur, 0.99), 0.9, 1e-4, ms, mt)
        ol.append([r["loss"], r["sup"], r["cons"]])
    close(rl, ol, 5e-5, "mt224 trace")
    close(out, r["logits"], 5e-4, "mt224 student logits")
    close(eo, r["t_logits"], 5e-4, "mt224 teacher logits")
    d = dict(meta=np.array([NL, NU, HW, FIRST, ITERS, 121, 122, 12000]), xl_sum=np.float64(xl.double().sum()), xu_sum=np.float64(xu.double().sum()),
             yl_sum=np.int64(yl.long().sum()), mask_sums=np.array(msum), losses=np.array(rl), cons_w=np.float64(w), lrs=np.array(lrs))
    put(d, "student_logits_last", out)
    put(d, "teacher_logits_last", eo)
    np.savez_compressed(os.path.join(OUT, "trace_mt224.npz"), **d)
    report["mt224_trace_err"] = float(np.abs(np.array(rl) - np.array(ol)).max())
    report["mt224_losses"] = [[round(v, 6) for v in row] for row in rl]


def cps96(report):
    R = load_reference()
    NL, NU, HW, FIRST, ITERS = 4, 4, 96, 6001, 2
    torch.manual_seed(1337)
    n1 = R.unet.UNet(3, 2)
    n2 = R.unet.UNet(3, 2)
    n1.train()
    n2.train()
    o1 = torch.optim.SGD(n1.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    o2 = torch.optim.SGD(n2.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    s1, s2 = R.medlr.Medical_LR(o1, 0.01, 30000), R.medlr.Medical_LR(o2, 0.01, 30000)
    _spin([o1, o2], [s1, s2], FIRST - 1)
    crit2 = R.med.Med_Sup_Loss(2)
    xl, yl = synth_batch(131, NL, HW, HW, 3, 2, 12)
    xu, _ = synth_batch(132, NU, HW, HW, 3, 2, 12)
    torch.manual_seed(1337)
    sa = unet_ref.init_state(None, 3, 2)
    sb = unet_ref.init_state(None, 3, 2)
    ba, bb = {}, {}
    rl, ol, msum = [], [], []
    for j in range(ITERS):
        cur = FIRST + j
        xx = torch.cat([xl, xu], 0)
        torch.manual_seed(13000 + j)
        a = n1(xx)
        b = n2(xx)
        sup = crit2(a[:NL], yl.long()) + crit2(b[:NL], yl.long())
        pa = torch.argmax(torch.softmax(a[NL:], 1).detach(), 1)
        pb = torch.argmax(torch.softmax(b[NL:], 1).detach(), 1)
        semi = crit2(a[NL:], pb) + crit2(b[NL:], pa)
        cw = 0.1 * R.utils.sigmoid_rampup(cur // 150, 200.0)
        loss = sup + cw * semi
        o1.zero_grad()
        o2.zero_grad()
        loss.backward()
        o1.step()
        o2.step()
        s1.step()
        s2.step()
        rl.append([loss.item(), sup.item(), semi.item()])
        torch.manual_seed(13000 + j)
        m1 = unet_ref.draw_dropout_masks(NL + NU, HW, HW)
        m2 = unet_ref.draw_dropout_masks(NL + NU, HW, HW)
        msum.append(mask_sums(m1) + mask_sums(m2))
        lr = laws_ref.medical_lr(cur, 0.01, 30000)
        r = steps_ref.cps_step(sa, sb, ba, bb, xl, yl.long(), xu, lr, lr, cw, 0.9, 1e-4, m1, m2)
        ol.append([r["loss"], r["sup"], r["semi"]])
    close(rl, ol, 1e-4, "cps96 trace")
    d = dict(meta=np.array([NL, NU, HW, FIRST, ITERS, 131, 132, 13000]), xl_sum=np.float64(xl.double().sum()), xu_sum=np.float64(xu.double().sum()),
             yl_sum=np.int64(yl.long().sum()), mask_sums=np.array(msum), losses=np.array(rl), cons_w=np.float64(cw))
    put(d, "logits1_last", a)
    put(d, "logits2_last", b)
    np.savez_compressed(os.path.join(OUT, "trace_cps96.npz"), **d)
    report["cps96_trace_err"] = float(np.abs(np.array(rl) - np.array(ol)).max())
    report["cps96_losses"] = [[round(v, 6) for v in row] for row in rl]


def hpfg224(report):
    R = load_reference()
    NL, NU, HW = 2, 2, 224
    CUR = (1000, 1001)
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
    _spin([o1, o2], [s1, s2], CUR[0] - 1)
    torch.manual_seed(1)
    sa = unet_ref.init_state(None, 1, 4, True)
    sb = unet_ref.init_state(None, 1, 4, True)
    se = unet_ref.clone_state(sb)
    ba, bb = {}, {}
    dense = R.dense.Dense_Loss(NL + NU, torch.device("cpu"))
    ce = torch.nn.CrossEntropyLoss(ignore_index=255)
    dl = R.dice.DiceLoss(4)
    xl, yl = synth_batch(141, NL, HW, HW, 1, 4, 32)
    xl1, yl1 = synth_batch(142, NL, HW, HW, 1, 4, 32)
    xu, _ = synth_batch(143, NU, HW, HW, 1, 4, 32)
    rng = np.random.RandomState(5)
    rl, ol, msum, cms = [], [], [], []
    for j, cur in enumerate(CUR):
        rep = NU // NL
        xl1r, yl1r = xl1.repeat(rep, 1, 1, 1), yl1.repeat(rep, 1, 1).long()
        cm = torch.tensor(gen.generate_params(NU, (HW, HW), rng=rng), dtype=torch.float)
        cms.append(cm)
        mix = torch.cat([xl, xl1r * (1.0 - cm) + xu * cm], 0)
        torch.manual_seed(14000 + j)
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
        pseudo = yl1r * (1.0 - c2) + torch.argmax(es[NL:], 1) * c2
        ps = dl(sa_[NL:], pseudo.unsqueeze(1))
        w = 0.1 * R.utils.linear_rampup(cur // 150, 200.0)
        cons2 = 0.0 if cur < 1000 else torch.mean((sb_[NL:] - es[NL:]) ** 2)
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
        s1.step()
        s2.step()
        rl.append([loss.item(), sup.item(), float(semi), ps.item(), con.item(), float(cons2)])
        torch.manual_seed(14000 + j)
        ma = unet_ref.draw_dropout_masks(NL + NU, HW, HW)
        mb = unet_ref.draw_dropout_masks(NL + NU, HW, HW)
        mt = unet_ref.draw_dropout_masks(NL + NU, HW, HW)
        msum.append(mask_sums(ma) + mask_sums(mb) + mask_sums(mt))
        lr = laws_ref.medical_lr(cur, 0.01, 30000)
        r = steps_ref.hpfg_step(sa, sb, se, ba, bb, xl, yl.long(), xl1r, yl1r, xu, cm, cur, lr, lr, 0.1, 200.0, 0.99, 0.9, 5e-4, ma, mb, mt)
        ol.append([r["loss"], r["sup"], r["semi"], r["pseudo_sup"], r["contrast"]])
    close(np.array(rl)[:, :5], ol, 1e-4, "hpfg224 trace")
    d = dict(meta=np.array([NL, NU, HW, 141, 142, 143, 14000, 5]), cur_itrs=np.array(CUR), xl_sum=np.float64(xl.double().sum()), xl1_sum=np.float64(xl1.double().sum()),
             xu_sum=np.float64(xu.double().sum()), yl_sum=np.int64(yl.long().sum()), mask_sums=np.array(msum), losses=np.array(rl),
             cutmix_sums=np.array([float(c.sum()) for c in cms]))
    put(d, "logits1_last", a)
    put(d, "logits2_last", b)
    put(d, "t_logits_last", eo)
    np.savez_compressed(os.path.join(OUT, "trace_hpfg224.npz"), **d)
    report["hpfg224_trace_err"] = float(np.abs(np.array(rl)[:, :5] - np.array(ol)).max())
    report["hpfg224_losses"] = [[round(v, 6) for v in row] for row in rl]


def grads224(report):
    R = load_reference()
    N, HW = 4, 224
    torch.manual_seed(1)
    net = R.unet.UNet(1, 4)
    net.train()
    crit = R.med.Med_Sup_Loss(4)
    x, lab = synth_batch(151, N, HW, HW, 1, 4, 32)
    torch.manual_seed(15000)
    out = net(x)
    loss = crit(out, lab.long())
    loss.backward()
    torch.manual_seed(15000)
    masks = unet_ref.draw_dropout_masks(N, HW, HW)
    st = unet_ref.init_state(1, 1, 4)
    names = steps_ref._train_state(st)
    o = unet_ref.unet_forward(st, x, True, masks)
    l_ = losses_ref.med_sup_loss(o, lab.long())
    g = steps_ref._grads(l_, st, names)
    close(out, o, 2e-5, "grads224 logits")
    close(loss, l_, 1e-6, "grads224 loss")
    d = dict(meta=np.array([N, HW, 151, 15000]), x_sum=np.float64(x.double().sum()), lab_sum=np.int64(lab.long().sum()), mask_sums=np.array(mask_sums(masks)),
             loss=np.float64(loss.item()))
    put(d, "logits", out)
    worst = 0.0
    for k, p_ in net.named_parameters():
        gr = p_.grad.detach()
        nrm = float(gr.double().norm())
        if nrm > 1e-6:                                   # (biases in front of a train-mode BatchNorm have a zero gradient up to rounding noise)
            worst = max(worst, float((gr - g[k]).double().norm()) / nrm)
        flat = gr.reshape(-1)
        d[f"g:{k}:norm"] = np.float64(nrm)
        d[f"g:{k}:sum"] = np.float64(gr.double().sum())
        d[f"g:{k}:sample"] = flat[:: max(1, flat.numel() // 64)][:64].numpy().copy()
    assert worst < 2e-4, worst
    np.savez_compressed(os.path.join(OUT, "grads224.npz"), **d)
    report["grads224_oracle_vs_reference_rel_l2_max"] = worst


def main():
    torch.set_num_threads(8)
    report = {}
    mt224(report)
    cps96(report)
    hpfg224(report)
    grads224(report)
    with open(os.path.join(OUT, "pinning_report_r3.json"), "w") as f:
        json.dump({"torch": torch.__version__, "reference": "fakerlove1/HPFG @ /root/reference", "checks": report}, f, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
