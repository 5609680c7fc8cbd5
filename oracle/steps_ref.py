"""Oracle (test infrastructure): the per-iteration step laws of the reference drivers, restated
functionally on oracle state dicts (CPU, fp32, torch autograd for the backward).

  supervised_step     /root/reference/sup_ACDC.py:83-93
  mean_teacher_step   /root/reference/2017_03_NIPS_Mean-Teacher_ACDC.py:82-113
  cps_step            /root/reference/2021_06_CVPR_CPS_ACDC.py:83-120
  hpfg_step           /root/reference/main.py:125-212 (incl. update_ema_variables_backbone :68-76)
  sgd_update          torch.optim.SGD(momentum, weight_decay) as built by utils/__init__.py:15-16
  ema_update          utils/utils.py:82-86 (parameters only, never BN buffers)

The learning rate, consistency weight and EMA alpha are passed in (see laws_ref) so that a test
can drive oracle and HIP path with the same scalars.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from . import laws_ref, losses_ref, unet_ref


def _grads(loss, st, names):
    ps = [st[n] for n in names]
    gs = torch.autograd.grad(loss, ps, allow_unused=True)
    return {n: (torch.zeros_like(p) if g is None else g) for n, p, g in zip(names, ps, gs)}


def sgd_update(st, grads: Dict[str, torch.Tensor], bufs: Dict[str, torch.Tensor], lr: float, momentum: float, weight_decay: float):
    with torch.no_grad():
        for n, g in grads.items():
            p = st[n]
            d = g + weight_decay * p if weight_decay != 0 else g.clone()
            if momentum != 0:
                if n not in bufs:
                    bufs[n] = d.clone()
                else:
                    bufs[n].mul_(momentum).add_(d)
                d = bufs[n]
            p.add_(d, alpha=-lr)


def ema_update(teacher, student, alpha: float, prefixes: Optional[List[str]] = None):
    with torch.no_grad():
        for n in unet_ref.param_names(student):
            if prefixes is not None and not any(n.startswith(p) for p in prefixes):
                continue
            teacher[n].mul_(alpha).add_(student[n].detach(), alpha=1.0 - alpha)


def _train_state(st):
    names = unet_ref.param_names(st)
    for n in names:
        st[n].requires_grad_(True)
    return names


def _detach_state(st):
    for v in st.values():
        if v.is_floating_point():
            v.requires_grad_(False)


def supervised_step(st, bufs, x, y, lr, momentum=0.9, weight_decay=5e-4, drop_masks=None):
    names = _train_state(st)
    logits = unet_ref.unet_forward(st, x, True, drop_masks)
    loss = losses_ref.med_sup_loss(logits, y)
    g = _grads(loss, st, names)
    _detach_state(st)
    sgd_update(st, g, bufs, lr, momentum, weight_decay)
    return {"loss": float(loss.detach()), "logits": logits.detach(), "grads": g}


def mean_teacher_step(st, ema, bufs, xl, yl, xu, lr, cons_w, alpha, momentum=0.9, weight_decay=1e-4,
                      masks_student=None, masks_teacher=None):
    names = _train_state(st)
    nl = xl.shape[0]
    x = torch.cat([xl, xu], 0)
    out = unet_ref.unet_forward(st, x, True, masks_student)
    soft = torch.softmax(out, 1)
    with torch.no_grad():
        t_out = unet_ref.unet_forward(ema, x, True, masks_teacher)   # teacher stays in train mode (:70)
        t_soft = torch.softmax(t_out, 1)
    sup = losses_ref.med_sup_loss(out[:nl], yl)
    cons = losses_ref.mse_consistency(soft[nl:], t_soft[nl:])
    loss = sup + cons_w * cons
    g = _grads(loss, st, names)
    _detach_state(st)
    sgd_update(st, g, bufs, lr, momentum, weight_decay)
    ema_update(ema, st, alpha)
    return {"loss": float(loss.detach()), "sup": float(sup.detach()), "cons": float(cons.detach()), "logits": out.detach(), "t_logits": t_out, "grads": g}


def ict_step(st, ema, bufs, xl, yl, xu, mix, lr, cons_w, alpha, momentum=0.9, weight_decay=1e-4, masks_student=None, masks_t0=None,
             masks_t1=None):
    """Interpolation consistency training, 2022_02_ISBI_ICT-MedSeg_ACDC.py:110-143: mix [nu/2,1,1,1] factors; the teacher (train
    mode) sees the two unlabelled halves separately, the student the labelled batch and their mix."""
    names = _train_state(st)
    nl, nu = xl.shape[0], xu.shape[0]
    u0, u1 = xu[:nu // 2], xu[nu // 2:]
    mixed = u0 * (1.0 - mix) + u1 * mix
    out = unet_ref.unet_forward(st, torch.cat([xl, mixed], 0), True, masks_student)
    soft = torch.softmax(out, 1)
    with torch.no_grad():
        e0 = torch.softmax(unet_ref.unet_forward(ema, u0, True, masks_t0), 1)
        e1 = torch.softmax(unet_ref.unet_forward(ema, u1, True, masks_t1), 1)
        target = e0 * (1.0 - mix) + e1 * mix
    sup = losses_ref.med_sup_loss(out[:nl], yl)
    cons = torch.mean((soft[nl:] - target) ** 2)
    loss = sup + cons_w * cons
    g = _grads(loss, st, names)
    _detach_state(st)
    sgd_update(st, g, bufs, lr, momentum, weight_decay)
    ema_update(ema, st, alpha)
    return {"loss": float(loss.detach()), "sup": float(sup.detach()), "cons": float(cons.detach()), "logits": out.detach(), "target": target}


def uamt_step(st, ema, bufs, xl, yl, xu, noise0, noises, threshold, lr, cons_w, alpha, momentum=0.9, weight_decay=1e-4,
              masks_student=None, masks_teacher=None):
    """Uncertainty-aware Mean Teacher, 2019_07_MICCAI_Uncertainty_Aware_ACDC.py:124-170.  noise0 [nu,...] and noises (T/2 fields of
    [2*nu,...]) are the raw normal draws of :130 / :142; masks_teacher lists the dropout masks of the 1 + T/2 teacher forwards."""
    names = _train_state(st)
    nl, nu = xl.shape[0], xu.shape[0]
    mt = masks_teacher or [None] * (1 + len(noises))
    out = unet_ref.unet_forward(st, torch.cat([xl, xu], 0), True, masks_student)
    soft = torch.softmax(out, 1)
    with torch.no_grad():
        ema_out = unet_ref.unet_forward(ema, xu + torch.clamp(noise0 * 0.1, -0.2, 0.2), True, mt[0])
        xr = xu.repeat(2, 1, 1, 1)
        preds = torch.cat([unet_ref.unet_forward(ema, xr + torch.clamp(nz * 0.1, -0.2, 0.2), True, mt[1 + i]) for i, nz in enumerate(noises)], 0)
        T = preds.shape[0] // nu
        pm = torch.softmax(preds, 1).reshape(T, nu, *preds.shape[1:]).mean(0)
        unc = -1.0 * torch.sum(pm * torch.log(pm + 1e-6), dim=1, keepdim=True)
    sup = losses_ref.med_sup_loss(out[:nl], yl)
    dist = (soft[nl:] - torch.softmax(ema_out, 1)) ** 2
    mask = (unc < threshold).float()
    cons = torch.sum(mask * dist) / (2 * torch.sum(mask) + 1e-16)
    loss = sup + cons_w * cons
    g = _grads(loss, st, names)
    _detach_state(st)
    sgd_update(st, g, bufs, lr, momentum, weight_decay)
    ema_update(ema, st, alpha)
    return {"loss": float(loss.detach()), "sup": float(sup.detach()), "cons": float(cons.detach()), "logits": out.detach(), "mask": mask,
            "uncertainty": unc, "t_logits": ema_out}


def _grads_joint(loss, states_names):
    """One backward through several networks, like the reference's single loss.backward()."""
    flat = [st[n] for st, names in states_names for n in names]
    gs = torch.autograd.grad(loss, flat, allow_unused=True)
    out, k = [], 0
    for st, names in states_names:
        d = {}
        for n in names:
            d[n] = torch.zeros_like(st[n]) if gs[k] is None else gs[k]
            k += 1
        out.append(d)
    return out


def cps_step(st1, st2, bufs1, bufs2, xl, yl, xu, lr1, lr2, cons_w, momentum=0.9, weight_decay=1e-4, masks1=None, masks2=None):
    n1, n2 = _train_state(st1), _train_state(st2)
    nl = xl.shape[0]
    x = torch.cat([xl, xu], 0)
    o1 = unet_ref.unet_forward(st1, x, True, masks1)
    o2 = unet_ref.unet_forward(st2, x, True, masks2)
    sup = losses_ref.med_sup_loss(o1[:nl], yl) + losses_ref.med_sup_loss(o2[:nl], yl)
    p1 = torch.argmax(o1[nl:].detach(), 1)      # argmax(softmax(.)) == argmax(.)
    p2 = torch.argmax(o2[nl:].detach(), 1)
    semi = losses_ref.med_sup_loss(o1[nl:], p2) + losses_ref.med_sup_loss(o2[nl:], p1)
    loss = sup + cons_w * semi
    g1, g2 = _grads_joint(loss, [(st1, n1), (st2, n2)])
    _detach_state(st1)
    _detach_state(st2)
    sgd_update(st1, g1, bufs1, lr1, momentum, weight_decay)
    sgd_update(st2, g2, bufs2, lr2, momentum, weight_decay)
    return {"loss": float(loss.detach()), "sup": float(sup.detach()), "semi": float(torch.as_tensor(semi).detach()), "logits1": o1.detach(), "logits2": o2.detach(),
            "grads1": g1, "grads2": g2}


def s4cvnet_step(st1, st2, ema, bufs1, bufs2, xl, yl, xu, noise, cur_itrs, lr1, lr2, consistency, rampup, ema_decay, momentum=0.9, weight_decay=5e-4,
                 masks1=None, masks2=None, masks_t=None):
    """2022_08_CVPR_S4CVNet_ACDC.py:107-167 with two U-Nets: both students see [labelled ; unlabelled], the EMA teacher of model2 sees the
    unlabelled images plus clamp(noise * 0.1, +-0.2) (noise = the raw normal draw of :109); cross Dice pseudo supervision (7w) and, from
    iteration 1000 on, the softmax MSE of each student against the teacher (w)."""
    n1, n2 = _train_state(st1), _train_state(st2)
    nl = xl.shape[0]
    vol = torch.cat([xl, xu], 0)
    o1 = unet_ref.unet_forward(st1, vol, True, masks1)
    s1 = torch.softmax(o1, 1)
    o2 = unet_ref.unet_forward(st2, vol, True, masks2)
    s2 = torch.softmax(o2, 1)
    with torch.no_grad():
        ot = unet_ref.unet_forward(ema, xu + torch.clamp(noise * 0.1, -0.2, 0.2), True, masks_t)
        st_ = torch.softmax(ot, 1)
    loss1 = 0.5 * (losses_ref.cross_entropy(o1[:nl], yl) + losses_ref.dice_loss(s1[:nl], yl.unsqueeze(1)))
    loss2 = 0.5 * (losses_ref.cross_entropy(o2[:nl], yl) + losses_ref.dice_loss(s2[:nl], yl.unsqueeze(1)))
    sup = loss1 + loss2
    p1 = torch.argmax(s1[nl:].detach(), 1)
    p2 = torch.argmax(s2[nl:].detach(), 1)
    ps1 = losses_ref.dice_loss(s1[nl:], p2.unsqueeze(1))
    ps2 = losses_ref.dice_loss(s2[nl:], p1.unsqueeze(1))
    w = consistency * laws_ref.linear_rampup(cur_itrs // 150, rampup)
    if cur_itrs < 1000:
        c1 = c2 = 0.0
    else:
        c1 = losses_ref.mse_consistency(s1[nl:], st_)
        c2 = losses_ref.mse_consistency(s2[nl:], st_)
    semi = (7 * w * ps1 + w * c1) + (7 * w * ps2 + w * c2)
    loss = sup + semi
    g1, g2 = _grads_joint(loss, [(st1, n1), (st2, n2)])
    _detach_state(st1)
    _detach_state(st2)
    sgd_update(st1, g1, bufs1, lr1, momentum, weight_decay)
    sgd_update(st2, g2, bufs2, lr2, momentum, weight_decay)
    ema_update(ema, st2, laws_ref.ema_alpha(cur_itrs, ema_decay))
    return {"loss": float(loss.detach()), "sup": float(sup.detach()), "semi": float(torch.as_tensor(semi).detach()), "ps1": float(ps1.detach()),
            "ps2": float(ps2.detach()), "cons1": float(torch.as_tensor(c1).detach()), "cons2": float(torch.as_tensor(c2).detach()), "logits1": o1.detach(),
            "logits2": o2.detach(), "t_logits": ot}


def adamw_update(st, grads, state, lr: float, weight_decay: float, betas=(0.9, 0.999), eps: float = 1e-8):
    """torch.optim.AdamW (decoupled weight decay, bias-corrected moments) over a state dict, in place; `state` keeps step / m / v."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    b1, b2 = betas
    with torch.no_grad():
        for k, g in grads.items():
            p = st[k]
            p.mul_(1.0 - lr * weight_decay)
            m = state.setdefault("m:" + k, torch.zeros_like(p))
            v = state.setdefault("v:" + k, torch.zeros_like(p))
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
            p.addcdiv_(m, denom, value=-lr / (1 - b1 ** t))


def ctct_step(st1, st2, bufs1, adam2, xl, yl, xu, lr1, lr2, cons_w, momentum=0.9, wd1=5e-4, wd2=0.05, masks1=None, draws2=None):
    """Cross teaching between the U-Net (st1, SGD) and SegFormer-B0 (st2, AdamW), 2021_12_MIDL_CTCT_ACDC.py:117-134.
    masks1: U-Net dropout masks; draws2: (drop-path draws, Dropout2d mask) of the SegFormer forward."""
    from . import segformer_ref
    n1 = _train_state(st1)
    n2 = [k for k in st2 if st2[k].is_floating_point() and "running" not in k]
    for k in n2:
        st2[k] = st2[k].detach().requires_grad_(True)
    nl = xl.shape[0]
    x = torch.cat([xl, xu], 0)
    o1 = unet_ref.unet_forward(st1, x, True, masks1)
    dp, mask = draws2 if draws2 is not None else (None, None)
    o2 = segformer_ref.segformer_forward(st2, x, True, dp, mask)
    s1, s2 = torch.softmax(o1, 1), torch.softmax(o2, 1)
    loss1 = 0.5 * (losses_ref.cross_entropy(o1[:nl], yl) + losses_ref.dice_loss(s1[:nl], yl.unsqueeze(1)))
    loss2 = 0.5 * (losses_ref.cross_entropy(o2[:nl], yl) + losses_ref.dice_loss(s2[:nl], yl.unsqueeze(1)))
    p1, p2 = torch.argmax(s1[nl:].detach(), 1), torch.argmax(s2[nl:].detach(), 1)
    ps1 = losses_ref.dice_loss(s1[nl:], p2.unsqueeze(1))
    ps2 = losses_ref.dice_loss(s2[nl:], p1.unsqueeze(1))
    loss = (loss1 + cons_w * ps1) + (loss2 + cons_w * ps2)
    gs = torch.autograd.grad(loss, [st1[k] for k in n1] + [st2[k] for k in n2])
    g1, g2 = dict(zip(n1, gs[:len(n1)])), dict(zip(n2, gs[len(n1):]))
    _detach_state(st1)
    for k in n2:
        st2[k] = st2[k].detach()
    sgd_update(st1, g1, bufs1, lr1, momentum, wd1)
    adamw_update(st2, g2, adam2, lr2, wd2)
    return {"loss": float(loss.detach()), "sup1": float(loss1.detach()), "sup2": float(loss2.detach()), "ps1": float(ps1.detach()), "ps2": float(ps2.detach()),
            "logits1": o1.detach(), "logits2": o2.detach()}


def hpfg_step(st1, st2, ema, bufs1, bufs2, xl, yl, xl1, yl1, xu, cutmix_mask, cur_itrs, lr1, lr2, consistency, rampup,
              ema_decay, momentum=0.9, weight_decay=5e-4, masks1=None, masks2=None, masks_t=None):
    """xl/yl: labelled batch; xl1/yl1: second labelled batch (already repeated to Nu, main.py:142-143);
    cutmix_mask [Nu,1,H,W] float 0/1."""
    n1, n2 = _train_state(st1), _train_state(st2)
    nl = xl.shape[0]
    m = cutmix_mask
    mix = torch.cat([xl, xl1 * (1.0 - m) + xu * m], 0)
    o1, _, _ = unet_ref.unet_forward(st1, mix, True, masks1, plus=True)
    s1 = torch.softmax(o1, 1)
    vol = torch.cat([xl, xu], 0)
    o2, h1, h2 = unet_ref.unet_forward(st2, vol, True, masks2, plus=True)
    s2 = torch.softmax(o2, 1)
    with torch.no_grad():
        ot, th1, th2 = unet_ref.unet_forward(ema, vol, True, masks_t, plus=True)
        st_ = torch.softmax(ot, 1)
    loss1 = 0.5 * (losses_ref.cross_entropy(o1[:nl], yl) + losses_ref.dice_loss(s1[:nl], yl.unsqueeze(1)))
    loss2 = 0.5 * (losses_ref.cross_entropy(o2[:nl], yl) + losses_ref.dice_loss(s2[:nl], yl.unsqueeze(1)))
    sup = loss1 + loss2
    contrast = losses_ref.dense_loss(h1, th1) + losses_ref.dense_loss(h2, th2)
    m2 = m.squeeze(1)
    pseudo = yl1.to(torch.float32) * (1.0 - m2) + torch.argmax(st_[nl:], 1).to(torch.float32) * m2
    pseudo_sup = losses_ref.dice_loss(s1[nl:], pseudo.unsqueeze(1))
    w = consistency * laws_ref.linear_rampup(cur_itrs // 150, rampup)
    cons2 = 0.0 if cur_itrs < 1000 else losses_ref.mse_consistency(s2[nl:], st_[nl:])
    semi = 7 * w * pseudo_sup + w * 0.0 + w * cons2 + w * contrast
    loss = sup + semi
    g1, g2 = _grads_joint(loss, [(st1, n1), (st2, n2)])
    _detach_state(st1)
    _detach_state(st2)
    sgd_update(st1, g1, bufs1, lr1, momentum, weight_decay)
    sgd_update(st2, g2, bufs2, lr2, momentum, weight_decay)
    alpha = laws_ref.ema_alpha(cur_itrs, ema_decay)
    ema_update(st2, st1, alpha, prefixes=["encoder.", "decoder."])   # model2 backbone pulled toward model1 (main.py:208)
    ema_update(ema, st2, alpha)
    return {"loss": float(loss.detach()), "sup": float(sup.detach()), "semi": float(torch.as_tensor(semi).detach()), "pseudo_sup": float(pseudo_sup.detach()),
            "contrast": float(contrast.detach()), "logits1": o1.detach(), "logits2": o2.detach(), "t_logits": ot,
            "grads1": g1, "grads2": g2}
