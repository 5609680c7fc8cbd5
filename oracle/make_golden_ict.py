"""Writes tests/golden/trace_ict.npz: 3 iterations of the reference's ICT step (2022_02_ISBI_ICT-MedSeg_ACDC.py:110-143) driven
with the reference's own UNet / Med_Sup_Loss pieces / Medical_LR / update_ema_variables (loaded by path like make_golden.py;
the driver file itself cannot be imported: it needs tensorboardX / medpy), side by side with oracle.steps_ref.ict_step, and
checks that the two agree.  Run once in the build container:  python -m oracle.make_golden_ict
"""
from __future__ import annotations

import copy
import os

import numpy as np
import torch

from . import laws_ref, steps_ref, unet_ref
from .make_golden import close, load_reference, pack, synth_batch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    torch.set_num_threads(4)
    R = load_reference()
    crit = R.med.Med_Sup_Loss(4)
    torch.manual_seed(1337)
    net = R.unet.UNet(1, 4)
    ema = copy.deepcopy(net)
    for p_ in ema.parameters():
        p_.requires_grad = False
    net.train()
    ema.train()
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    sch = R.medlr.Medical_LR(opt, 0.01, 30000)
    xl, yl = synth_batch(41, 2, 32, 32)
    xu, _ = synth_batch(42, 4, 32, 32)
    st = unet_ref.init_state(1337, 1, 4)
    est = unet_ref.clone_state(st)
    bufs = {}
    cons_w = 0.1 * R.utils.sigmoid_rampup(40, 200.0)
    rng = np.random.RandomState(5)
    rl, ol, mm, mixes = [], [], [], []
    for k in range(1, 4):
        mix = torch.tensor(rng.beta(0.2, 0.2, size=(2, 1, 1, 1)), dtype=torch.float)
        mixes.append(mix.numpy())
        u0, u1 = xu[:2], xu[2:]
        mixed = u0 * (1.0 - mix) + u1 * mix
        torch.manual_seed(3000 + k)
        out = net(torch.cat([xl, mixed], 0))
        soft = torch.softmax(out, 1)
        with torch.no_grad():
            e0 = torch.softmax(ema(u0), dim=1)
            e1 = torch.softmax(ema(u1), dim=1)
            target = e0 * (1.0 - mix) + e1 * mix
        sup = crit(out[:2], yl.long())                 # 0.5 * (CE + Dice(softmax)) = the driver's supervised_loss (:131-134)
        cons = torch.mean((soft[2:] - target) ** 2)
        loss = sup + cons_w * cons
        opt.zero_grad()
        loss.backward()
        opt.step()
        sch.step()
        R.utils.update_ema_variables(net, ema, 0.99, k)
        rl.append([loss.item(), sup.item(), cons.item()])
        torch.manual_seed(3000 + k)                     # the dropout draws of the three forwards, in the same order
        ms = unet_ref.draw_dropout_masks(4, 32, 32)
        m0 = unet_ref.draw_dropout_masks(2, 32, 32)
        m1 = unet_ref.draw_dropout_masks(2, 32, 32)
        mm.append((ms, m0, m1))
        r = steps_ref.ict_step(st, est, bufs, xl, yl.long(), xu, mix, laws_ref.medical_lr(k, 0.01, 30000), cons_w, laws_ref.ema_alpha(k, 0.99),
                               0.9, 1e-4, ms, m0, m1)
        ol.append([r["loss"], r["sup"], r["cons"]])
    close(rl, ol, 2e-5, "ict trace")
    for k_, v_ in ema.state_dict().items():
        close(v_, est[k_], 1e-5, f"ict ema {k_}")
    err = float(np.abs(np.array(rl) - np.array(ol)).max())
    np.savez_compressed(os.path.join(OUT, "trace_ict.npz"), xl=xl.numpy(), yl=yl.numpy(), xu=xu.numpy(), cons_w=np.float64(cons_w),
                        losses=np.array(rl), mixes=np.stack(mixes), student_logits_last=out.detach().numpy(), target_last=target.numpy(),
                        oracle_err=np.float64(err),
                        **{f"it{k}_{w}{i}": pack(m) for k, trip in enumerate(mm) for w, ml in zip(("s", "a", "b"), trip) for i, m in enumerate(ml)})
    print("trace_ict.npz written; reference vs oracle max |d loss| =", err)


if __name__ == "__main__":
    main()
