"""Writes tests/golden/trace_uamt.npz: 2 iterations of the reference's uncertainty-aware Mean-Teacher step
(2019_07_MICCAI_Uncertainty_Aware_ACDC.py:124-170) driven with the reference's own UNet / Med_Sup_Loss / Medical_LR /
update_ema_variables (loaded by path like make_golden.py; the driver file itself cannot be imported: it needs tensorboardX /
medpy, so its loop body is followed line by line here), side by side with oracle.steps_ref.uamt_step, and checks that the two
agree.  Student and teacher are two separate constructions, as in the driver (:86-87).  The entropy threshold of the driver
(<= ln 2) masks every pixel of an untrained 4-class net, so the trace uses a threshold near the median entropy instead.
Run once in the build container:  python -m oracle.make_golden_uamt
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import laws_ref, steps_ref, unet_ref
from .make_golden import close, load_reference, pack, synth_batch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
T = 8


def main():
    torch.set_num_threads(4)
    R = load_reference()
    crit = R.med.Med_Sup_Loss(4)
    torch.manual_seed(1337)
    net = R.unet.UNet(1, 4)
    ema = R.unet.UNet(1, 4)
    for p_ in ema.parameters():
        p_.requires_grad = False
    net.train()
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    sch = R.medlr.Medical_LR(opt, 0.01, 30000)
    xl, yl = synth_batch(51, 2, 32, 32)
    xu, _ = synth_batch(52, 2, 32, 32)
    st = unet_ref.init_state(1337, 1, 4)
    est = unet_ref.init_state(None, 1, 4)            # second construction continues the generator, like build_model twice
    bufs = {}
    cons_w = 0.1 * R.utils.sigmoid_rampup(40, 200.0)
    g = torch.Generator().manual_seed(77)
    rl, ol, mm, nz_all, thr_all, frac = [], [], [], [], [], []
    for k in range(1, 3):
        noise0 = torch.randn(xu.shape, generator=g)
        noises = [torch.randn(2 * xu.shape[0], *xu.shape[1:], generator=g) for _ in range(T // 2)]
        nz_all.append(torch.cat([noise0] + noises, 0).numpy())
        torch.manual_seed(4000 + k)
        out = net(torch.cat([xl, xu], 0))
        soft = torch.softmax(out, 1)
        with torch.no_grad():
            ema_out = ema(xu + torch.clamp(noise0 * 0.1, -0.2, 0.2))
        xr = xu.repeat(2, 1, 1, 1)
        stride = xr.shape[0] // 2
        preds = torch.zeros([stride * T, 4, 32, 32])
        for i in range(T // 2):
            with torch.no_grad():
                preds[2 * stride * i:2 * stride * (i + 1)] = ema(xr + torch.clamp(noises[i] * 0.1, -0.2, 0.2))
        preds = torch.softmax(preds, dim=1).reshape(T, stride, 4, 32, 32)
        preds = torch.mean(preds, dim=0)
        unc = -1.0 * torch.sum(preds * torch.log(preds + 1e-6), dim=1, keepdim=True)
        thr = float(unc.median()) + 1e-3
        sup = crit(out[:2], yl.long())
        dist = (soft[2:] - torch.softmax(ema_out, dim=1)) ** 2
        mask = (unc < thr).float()
        cons = torch.sum(mask * dist) / (2 * torch.sum(mask) + 1e-16)
        loss = sup + cons_w * cons
        opt.zero_grad()
        loss.backward()
        opt.step()
        sch.step()
        R.utils.update_ema_variables(net, ema, 0.99, k)
        rl.append([loss.item(), sup.item(), cons.item()])
        thr_all.append(thr)
        frac.append(float(mask.mean()))
        torch.manual_seed(4000 + k)                     # the dropout draws of the six forwards, in the same order
        ms = unet_ref.draw_dropout_masks(4, 32, 32)
        mt = [unet_ref.draw_dropout_masks(2, 32, 32)] + [unet_ref.draw_dropout_masks(4, 32, 32) for _ in range(T // 2)]
        mm.append([ms] + mt)
        r = steps_ref.uamt_step(st, est, bufs, xl, yl.long(), xu, noise0, noises, thr, laws_ref.medical_lr(k, 0.01, 30000), cons_w,
                                laws_ref.ema_alpha(k, 0.99), 0.9, 1e-4, ms, mt)
        ol.append([r["loss"], r["sup"], r["cons"]])
        assert float((r["mask"] - mask).abs().sum()) <= 2, "oracle mask differs from the reference's"
    close(rl, ol, 2e-5, "uamt trace")
    for k_, v_ in ema.state_dict().items():
        close(v_, est[k_], 1e-5, f"uamt ema {k_}")
    err = float(np.abs(np.array(rl) - np.array(ol)).max())
    np.savez_compressed(os.path.join(OUT, "trace_uamt.npz"), xl=xl.numpy(), yl=yl.numpy(), xu=xu.numpy(), cons_w=np.float64(cons_w),
                        losses=np.array(rl), noise=np.stack(nz_all), thresholds=np.array(thr_all), mask_frac=np.array(frac),
                        student_logits_last=out.detach().numpy(), mask_last=pack(mask), uncertainty_last=unc.numpy(), oracle_err=np.float64(err),
                        **{f"it{k}_f{j}_{i}": pack(m) for k, sets in enumerate(mm) for j, ml in enumerate(sets) for i, m in enumerate(ml)})
    print("trace_uamt.npz written; reference vs oracle max |d loss| =", err, "mask fractions", frac, "losses", rl)


if __name__ == "__main__":
    main()
