"""GPU: the forwards of a step's networks as multi-network launches (hpfg_conv_fwd_multi, model.unet.forward_multi; round 5).

A multi-network launch runs the single-network kernel's body per network (the network is a grid index), so per network it must produce what
``net(x)`` produces BIT FOR BIT: raw outputs, BatchNorm sums (integer accumulators: order-free), running statistics, side tensors -- and
therefore the same losses, gradients and weights after any number of steps.  Reference ops: model/unet.py:61-117 forward of every network of
2017_03_NIPS_Mean-Teacher_ACDC.py:95-101 / 2021_06_CVPR_CPS_ACDC.py:95-101 / main.py:152-161."""
import ctypes as C
from copy import deepcopy

import pytest
import torch

from hpfg_amd import _lib as L
from hpfg_amd.datasets.synthetic import synth_batch
from hpfg_amd.model import UNet, reset_dropout_streams
from hpfg_amd.model.unet import forward_multi
from hpfg_amd.train import GraphedStep, MeanTeacherStep, batch_pair
from tests.dp_rank_worker import _frozen, opt_args

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _mt_run(paired, size, nl, nu, iters, graph_from=None):
    reset_dropout_streams()
    torch.manual_seed(11)
    m = UNet(1, 4).to(DEV)
    ema = _frozen(m)
    m.train()
    st = MeanTeacherStep(m, ema, opt_args(), None)
    st.paired = paired
    xl, yl = synth_batch(3, nl, size, size, 1, 4, 8)
    xu, _ = synth_batch(4, nu, size, size, 1, 4, 8)
    xl, xu = batch_pair(xl.to(DEV), xu.to(DEV))
    inputs = [xl, yl.to(DEV), xu]
    rows, runner = [], None
    for it in range(1, iters + 1):
        if graph_from is not None and it >= graph_from:
            if runner is None:
                runner = GraphedStep(st, inputs, warmup=0, alias_inputs=True)
            r = runner.step(inputs, it)
        else:
            r = st.step(*inputs, it)
        rows.append(torch.cat([r["parts"].detach().reshape(-1), r["logits"].detach().reshape(-1)[:64], r["t_logits"].detach().reshape(-1)[:64]]).clone())
    torch.cuda.synchronize()
    bufs = torch.cat([b.detach().reshape(-1).float() for b in ema.buffers()])
    return torch.stack(rows).cpu(), m.flat_params.detach().cpu().clone(), ema.flat_params.detach().cpu().clone(), bufs.cpu(), m.flat_grads.detach().cpu().clone()


@pytest.mark.parametrize("size,nl,nu,iters", [(64, 2, 2, 4), (48, 3, 1, 3), (224, 2, 2, 2)])
def test_mean_teacher_step_paired_equals_two_streams(size, nl, nu, iters):
    ref = _mt_run(False, size, nl, nu, iters)
    got = _mt_run(True, size, nl, nu, iters)
    for a, b, what in zip(got, ref, ("losses / logits", "student weights", "teacher weights", "teacher BatchNorm buffers", "student gradients")):
        assert torch.equal(a, b), what


def test_paired_step_captures_and_replays():
    ref = _mt_run(True, 64, 2, 2, 5)
    got = _mt_run(True, 64, 2, 2, 5, graph_from=2)
    for a, b in zip(got, ref):
        assert torch.equal(a, b)


@pytest.mark.parametrize("in_ch,ncls,size,n", [(3, 2, 96, 4), (1, 4, 32, 2)])
def test_three_networks_one_launch_per_layer(in_ch, ncls, size, n):
    """MAX_NETS networks, mixed trainable / frozen, against one forward each; then the trainable ones back-propagate."""
    def build():
        reset_dropout_streams()
        torch.manual_seed(5)
        nets = [UNet(in_ch, ncls).to(DEV) for _ in range(2)]
        nets.append(_frozen(nets[1]))
        for m in nets:
            m.train()
        return nets
    x = synth_batch(9, n, size, size, in_ch, ncls, 8)[0].to(DEV)
    xs = [x, x.flip(0).contiguous(), x]
    w = torch.randn(n, ncls, size, size, device=DEV)

    def finish(nets, outs):
        loss = sum((o * w).sum() for o, m in zip(outs, nets) if o.requires_grad)
        loss.backward()
        torch.cuda.synchronize()
        return ([o.detach().cpu().clone() for o in outs], [m.flat_grads.detach().cpu().clone() for m in nets[:2]],
                [torch.cat([b.detach().reshape(-1).float() for b in m.buffers()]).cpu() for m in nets])

    nets = build()
    ref = finish(nets, [m(xi) if any(p.requires_grad for p in m.parameters()) else m(xi).detach() for m, xi in zip(nets, xs)])
    nets = build()
    outs = forward_multi(nets, xs, grad=[True, True, False])
    assert outs is not None and outs[0].requires_grad and outs[1].requires_grad and not outs[2].requires_grad
    got = finish(nets, outs)
    for ga, ra in zip(got, ref):
        for a, b in zip(ga, ra):
            assert torch.equal(a, b)


def test_forward_multi_declines_what_it_cannot_pair():
    torch.manual_seed(1)
    a, b = UNet(1, 4).to(DEV), UNet(1, 4).to(DEV)
    x = torch.randn(2, 1, 32, 32, device=DEV)
    a.train(), b.eval()
    assert forward_multi([a, b], [x, x]) is None                      # eval-mode network: its own launches
    b.train()
    assert forward_multi([a, b], [x, x[:1]]) is None                   # different batch shapes
    b.math = "f32"
    assert forward_multi([a, b], [x, x]) is None                       # exact-fp32 math has no multi-network kernels
    assert forward_multi([a, a], [x, x]) is None                       # the same network twice


def test_conv_fwd_multi_refuses_mismatched_descriptors():
    lib = L.load()
    z = torch.zeros(2 * 8 * 8 * 32, device=DEV)
    cas = (L.ConvArgs * 2)()
    for k, ca in enumerate(cas):
        ca.a0.z, ca.a0.mode, ca.a0.C, ca.a0.Hs, ca.a0.Ws, ca.a0.pstride = z.data_ptr(), L.ACT_BNACT, 32, 8, 8, 32
        ca.a0.bn, ca.a0.bn_stride = z.data_ptr(), 32
        ca.wpk, ca.out, ca.out_pstride, ca.Cout, ca.CoutPad = z.data_ptr(), z.data_ptr(), 32, 32, 32
        ca.N, ca.H, ca.W, ca.taps, ca.math = 2, 8, 8, 9, L.MATH_BF16X3
    cas[1].H = 4
    assert lib.hpfg_conv_fwd_multi(cas, 2, None) < 0 and b"another layer shape" in lib.hpfg_last_error()
    cas[1].H = 8
    cas[1].math = L.MATH_F32
    assert lib.hpfg_conv_fwd_multi(cas, 2, None) < 0
    cas[1].math = L.MATH_BF16X3
    cas[0].a0.mode = cas[1].a0.mode = L.ACT_DZ
    assert lib.hpfg_conv_fwd_multi(cas, 2, None) < 0 and b"forward loader kinds" in lib.hpfg_last_error()
    assert lib.hpfg_conv_fwd_multi(cas, 4, None) < 0
