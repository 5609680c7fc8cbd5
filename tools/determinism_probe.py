"""Run two Mean-Teacher steps twice from the same seed and report the first tensors that differ between the runs (raw conv
outputs, BatchNorm tables, activation gradients, parameter gradients) plus the first conv against torch -- used to localise a
run-to-run difference that only showed with the teacher forward overlapped on a second stream (GPU only, diagnostics)."""
import os, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_fullsize as T
from hpfg_amd.train import MeanTeacherStep
from hpfg_amd.datasets.synthetic import synth_batch
def run():
    from hpfg_amd.model.unet import reset_dropout_streams
    reset_dropout_streams()          # every network instance owns a dropout stream: the second run must start where the first one did
    a = T._cfg("mean_teacher_unet_30k_224x224_ACDC.yaml")
    torch.manual_seed(a.seed)
    m = T.build_model(a).to(T.DEV); e = T._teacher(m); m.train(); e.train()
    st = MeanTeacherStep(m, e, a)
    xl, yl = synth_batch(1, 8, 224, 224, 1, 4, 32); xu, _ = synth_batch(2, 8, 224, 224, 1, 4, 32)
    xl, yl, xu = xl.to(T.DEV), yl.to(T.DEV), xu.to(T.DEV)
    r = st.step(xl, yl, xu, 1, cons_w=0.05)
    torch.cuda.synchronize()
    p1, e1 = m.flat_params.clone(), e.flat_params.clone()
    mom1 = st.optimizer.flat_momentum.clone() if hasattr(st.optimizer, "flat_momentum") else None
    wt = dict(e.named_parameters())["encoder.in_conv.conv_conv.0.weight"].detach().clone()
    bt = dict(e.named_parameters())["encoder.in_conv.conv_conv.0.bias"].detach().clone()
    w1 = dict(m.named_parameters())["encoder.in_conv.conv_conv.0.weight"].detach().clone()
    b1 = dict(m.named_parameters())["encoder.in_conv.conv_conv.0.bias"].detach().clone()
    r = st.step(xl, yl, xu, 2, cons_w=0.05)
    torch.cuda.synchronize()
    x = torch.cat([xl, xu], 0)
    ref = torch.nn.functional.conv2d(x, w1, b1, padding=1).permute(0, 2, 3, 1)
    eng0 = next(iter(m._engines.values()))[0]
    d = (eng0.z["encoder.in_conv.conv_conv.0"] - ref).abs()
    bad = (d > 1e-3)
    reft = torch.nn.functional.conv2d(x, wt, bt, padding=1).permute(0, 2, 3, 1)
    zz = eng0.z["encoder.in_conv.conv_conv.0"]
    idx = bad.nonzero()
    print("bad elems matching TEACHER-weight result:", int(((zz - reft).abs() < 1e-4)[bad].sum()), "of", int(bad.sum()))
    print("bad channels", sorted(set(idx[:, 3].tolist())), "cols sample", sorted(set(idx[:, 2].tolist()))[:40])
    print("sample", [(int(a), int(b), int(c), int(dd), float(zz[a, b, c, dd]), float(ref[a, b, c, dd])) for a, b, c, dd in idx[:6].tolist()])
    print("first conv vs torch: maxerr", float(d.max()), "bad elems", int(bad.sum()), "bad images", sorted(set(bad.nonzero()[:, 0].tolist()))[:16],
          "rows", sorted(set(bad.nonzero()[:, 1].tolist()))[:20])
    eng = next(iter(m._engines.values()))[0]
    teng = next(iter(e._engines.values()))[0]
    return dict(z={k: v.clone() for k, v in eng.z.items()}, bn={k: v.clone() for k, v in eng.bn.items()}, tz={k: v.clone() for k, v in teng.z.items()},
                dA={k: v.clone() for k, v in eng.dA.items()}, g=m.flat_grads.clone(), gnames={k: v.grad.clone() for k, v in m.named_parameters() if v.grad is not None},
                order=[s.name for s in eng.order], logits=r["logits"].clone(), p1=p1, e1=e1, mom1=mom1)
a = run(); b = run()
print("logits", torch.equal(a["logits"], b["logits"]), "p1", torch.equal(a["p1"], b["p1"]), "e1", torch.equal(a["e1"], b["e1"]), "mom1", None if a["mom1"] is None else torch.equal(a["mom1"], b["mom1"]))
for n in a["order"]:
    if n in a["z"] and not torch.equal(a["z"][n], b["z"][n]): print("student z differs:", n, float((a["z"][n] - b["z"][n]).abs().max()))
    if n in a["tz"] and not torch.equal(a["tz"][n], b["tz"][n]): print("teacher z differs:", n)
    if n in a["bn"] and not torch.equal(a["bn"][n], b["bn"][n]):
        d = (a["bn"][n] - b["bn"][n]).abs().amax(1); print("student bn differs:", n, d.tolist())
    if n in a["dA"] and not torch.equal(a["dA"][n], b["dA"][n]): print("dA differs:", n, float((a["dA"][n] - b["dA"][n]).abs().max()))
bad = [k for k in a["gnames"] if not torch.equal(a["gnames"][k], b["gnames"][k])]
print("grads differing:", len(bad), bad[:10])
