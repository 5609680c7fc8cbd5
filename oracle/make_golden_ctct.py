"""Writes tests/golden/trace_ctct.npz: 2 iterations of the reference's CTCT step (2021_12_MIDL_CTCT_ACDC.py:117-134) driven with the
reference's own UNet, SegFormer, CrossEntropyLoss / DiceLoss, SGD + AdamW and Medical_LR (modules loaded by path; the driver file itself
needs tensorboardX / medpy, so its loop body is followed line by line), side by side with oracle.steps_ref.ctct_step.
Run once in the build container:  python -m oracle.make_golden_ctct
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn

from . import laws_ref, segformer_ref as S, steps_ref, unet_ref
from .make_golden import _load, close, load_reference, pack, synth_batch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
HW = 64


def main():
    torch.set_num_threads(4)
    R = load_reference()
    seg = _load("ref_segformer", "model/segformer.py")
    torch.manual_seed(1)
    net1 = R.unet.UNet(1, 4)
    net2 = seg.SegFormer(image_size=[HW, HW], in_channels=1, num_classes=4)
    st1 = unet_ref.init_state(1, 1, 4)
    st2 = S.init_state(None, 1, 4)
    net1.train(), net2.train()
    opt1 = torch.optim.SGD(net1.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
    opt2 = torch.optim.AdamW(net2.parameters(), lr=0.0008, weight_decay=0.05)
    sch1, sch2 = R.medlr.Medical_LR(opt1, 0.01, 30000), R.medlr.Medical_LR(opt2, 0.0008, 30000)
    ce, dice = nn.CrossEntropyLoss(ignore_index=255), R.dice.DiceLoss(4)
    xl, yl = synth_batch(71, 2, HW, HW)
    xu, _ = synth_batch(72, 2, HW, HW)
    cons_w = 0.1 * R.utils.sigmoid_rampup(40, 200.0)
    bufs1, adam2 = {}, {}
    rl, ol, mm, dd = [], [], [], []
    for k in range(1, 3):
        x = torch.cat([xl, xu], 0)
        torch.manual_seed(6000 + k)
        o1 = net1(x)
        s1 = torch.softmax(o1, dim=1)
        o2 = net2(x)
        s2 = torch.softmax(o2, dim=1)
        loss1 = 0.5 * (ce(o1[:2], yl.long()) + dice(s1[:2], yl.long().unsqueeze(1)))
        loss2 = 0.5 * (ce(o2[:2], yl.long()) + dice(s2[:2], yl.long().unsqueeze(1)))
        p1, p2 = torch.argmax(s1[2:].detach(), dim=1), torch.argmax(s2[2:].detach(), dim=1)
        ps1, ps2 = dice(s1[2:], p2.unsqueeze(1)), dice(s2[2:], p1.unsqueeze(1))
        loss = (loss1 + cons_w * ps1) + (loss2 + cons_w * ps2)
        opt1.zero_grad()
        opt2.zero_grad()
        loss.backward()
        opt1.step()
        opt2.step()
        sch1.step()
        sch2.step()
        rl.append([loss.item(), loss1.item(), loss2.item(), ps1.item(), ps2.item()])
        torch.manual_seed(6000 + k)                      # the same draws, in the order the two forwards made them
        m1 = unet_ref.draw_dropout_masks(4, HW, HW)
        dp, mask = S.draw_randomness(4)
        mm.append(m1)
        dd.append((dp, mask))
        r = steps_ref.ctct_step(st1, st2, bufs1, adam2, xl, yl.long(), xu, laws_ref.medical_lr(k, 0.01, 30000), laws_ref.medical_lr(k, 0.0008, 30000), cons_w,
                                0.9, 5e-4, 0.05, m1, (dp, mask))
        ol.append([r["loss"], r["sup1"], r["sup2"], r["ps1"], r["ps2"]])
    close(rl, ol, 3e-5, "ctct trace")
    # AdamW's first step moves every element by exactly lr * sign(g): where a gradient is at rounding level its sign is not stable between
    # two fp32 evaluations, so isolated elements may differ by up to 2 * lr.  Check the bulk, not the maximum.
    tot = off = 0
    for k_, v_ in net2.state_dict().items():
        if v_.is_floating_point():
            d_ = (v_ - st2[k_]).abs()
            tot += d_.numel()
            off += int((d_ > 1e-4).sum())
    assert off / tot < 1e-3, f"{off} of {tot} SegFormer weights differ by more than 1e-4 after two AdamW steps"
    close(o2, r["logits2"], 1e-4, "second-iteration SegFormer logits")
    err = float(np.abs(np.array(rl) - np.array(ol)).max())
    extra = {}
    for k, (dp, mask) in enumerate(dd):
        extra[f"it{k}_dp"] = np.stack([np.zeros((4, 1, 1), np.float32) if d is None else d.numpy() for d in dp])
        extra[f"it{k}_dpnone"] = np.array([d is None for d in dp])
        extra[f"it{k}_mask"] = mask.numpy()
    np.savez_compressed(os.path.join(OUT, "trace_ctct.npz"), xl=xl.numpy(), yl=yl.numpy(), xu=xu.numpy(), cons_w=np.float64(cons_w), losses=np.array(rl),
                        logits1_last=o1.detach().numpy(), logits2_last=o2.detach().numpy(), oracle_err=np.float64(err),
                        **{f"it{k}_u{i}": pack(m) for k, ml in enumerate(mm) for i, m in enumerate(ml)}, **extra)
    print("trace_ctct.npz written; reference vs oracle max |d loss| =", err, "losses", rl)


if __name__ == "__main__":
    main()
