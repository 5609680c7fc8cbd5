"""Writes tests/golden/segformer_b0.npz: the reference's SegFormer-B0 (model/segformer.py, loaded by path) on a small batch --
eval-mode logits, train-mode logits with the stochastic draws replayed, the loss of the reference's Med_Sup_Loss and a summary of every
parameter gradient -- and checks oracle.segformer_ref against all of it (state_dict keys, values, outputs, gradients).
Run once in the build container:  python -m oracle.make_golden_segformer
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import losses_ref, segformer_ref as S
from .make_golden import _load, close, load_reference, synth_batch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    torch.set_num_threads(4)
    R = load_reference()
    seg = _load("ref_segformer", "model/segformer.py")
    torch.manual_seed(1337)
    net = seg.SegFormer(image_size=[64, 64], in_channels=1, num_classes=4, model_name="B0")
    st = S.init_state(1337, 1, 4)
    sd = net.state_dict()
    assert list(sd.keys()) == list(st.keys()), "state_dict keys / order differ"
    for k in sd:
        close(sd[k], st[k], 0.0, f"init {k}")
    x, y = synth_batch(61, 2, 64, 64)
    net.eval()
    with torch.no_grad():
        ev = net(x)
        close(ev, S.segformer_forward(st, x, False), 2e-5, "eval logits")
    net.train()
    torch.manual_seed(99)
    out = net(x)
    loss = R.med.Med_Sup_Loss(4)(out, y.long())
    loss.backward()
    torch.manual_seed(99)
    dp, mask = S.draw_randomness(2)
    names = [k for k in st if st[k].is_floating_point() and "running" not in k]
    for k in names:
        st[k] = st[k].clone().requires_grad_(True)
    taps = {}
    o2 = S.segformer_forward(st, x, True, dp, mask, taps=taps)
    l2 = losses_ref.med_sup_loss(o2, y.long())
    gs = torch.autograd.grad(l2, [st[k] for k in names])
    close(out, o2, 2e-5, "train logits")
    close(loss, l2, 1e-6, "loss")
    ref_g = dict(net.named_parameters())
    gsum = {}
    for k, g in zip(names, gs):
        close(ref_g[k].grad, g, 2e-5 * max(1.0, float(ref_g[k].grad.abs().max())), f"grad {k}")
        gsum["g:" + k] = np.array([float(ref_g[k].grad.sum()), float(ref_g[k].grad.abs().sum()), float(ref_g[k].grad.abs().max())])
    close(net.state_dict()["decoder.linear_fuse.bn.running_var"], st["decoder.linear_fuse.bn.running_var"], 1e-6, "running_var")
    np.savez_compressed(os.path.join(OUT, "segformer_b0.npz"), x=x.numpy(), y=y.numpy(), eval_logits=ev.numpy(), train_logits=out.detach().numpy(),
                        loss=np.float64(loss.item()), stage4=taps["stage4"].detach().numpy(),
                        n_params=np.int64(sum(p.numel() for p in net.parameters())), **gsum)
    print("segformer_b0.npz written;", len(names), "parameter tensors,", sum(p.numel() for p in net.parameters()), "parameters; loss", loss.item())


if __name__ == "__main__":
    main()
