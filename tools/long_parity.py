"""Long-horizon parity run: ITERS Mean-Teacher iterations of the HIP step and of the fp32 CPU oracle side by side (same
weights, batches and dropout masks), then held-out Dice of student and teacher.  usage: python tools/long_parity.py [iters] [hw] [math ...] [--nl=N --nu=N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import tests.test_gpu_train_parity as T  # noqa: E402
from oracle import losses_ref, unet_ref  # noqa: E402

opts = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--")}      # --nl=8 --nu=8: labelled / unlabelled images per batch
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
iters = int(argv[0]) if len(argv) > 0 else 500
T.ITERS = iters
T.HW = int(argv[1]) if len(argv) > 1 else 64
T.NL, T.NU = opts.get("--nl", T.NL), opts.get("--nu", T.NU)
maths = argv[2:] or ["bf16x3", "f32"]
print(f"# long-horizon parity: {iters} Mean-Teacher iterations, {T.NL} + {T.NU} images of {T.HW} x {T.HW}, HIP step vs fp32 CPU oracle (same weights, batches, dropout masks)", flush=True)
torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
t0 = time.time()
losses, st, ema_st = T._train_oracle("f32")
print(f"oracle f32: {iters} iterations in {time.time() - t0:.1f} s; loss first/last 5: {losses[:5].mean():.4f} {losses[-5:].mean():.4f}", flush=True)
xe, ye = T.synth_batch(777, 16, T.HW, T.HW)
with torch.no_grad():
    ref = {"student": unet_ref.unet_forward(st, xe, False), "teacher": unet_ref.unet_forward(ema_st, xe, False)}
dref = {k: losses_ref.mean_foreground_dice(v.argmax(1).numpy(), ye.numpy(), 4) for k, v in ref.items()}
print("oracle Dice", dref, flush=True)
if "ctl" in maths:
    t0 = time.time()
    _, a, b = T._train_oracle("f64acc")
    with torch.no_grad():
        ctl = {"student": unet_ref.unet_forward(a, xe, False), "teacher": unet_ref.unet_forward(b, xe, False)}
    for k in ctl:
        d = losses_ref.mean_foreground_dice(ctl[k].argmax(1).numpy(), ye.numpy(), 4)
        print(f"control (fp64-accumulating oracle, {time.time() - t0:.0f} s) {k}: Dice {d:.5f} (delta {d - dref[k]:+.2e}), max|dlogit| {float((ctl[k] - ref[k]).abs().max()):.3e}", flush=True)
for math in [m for m in maths if m != "ctl"]:
    from copy import deepcopy
    from hpfg_amd.model import UNet
    from hpfg_amd.train import MeanTeacherStep
    from hpfg_amd.utils import AttrDict
    torch.manual_seed(1337)
    m = UNet(1, 4).to(T.DEV)
    m.math = math
    ema = deepcopy(m)
    for p in ema.parameters():
        p.requires_grad = False
    m.train(), ema.train()
    step = MeanTeacherStep(m, ema, AttrDict(dict(T.ARGS)))
    got = []
    for k in range(1, iters + 1):
        xl, yl, xu, ms, mt = T._batch(k)
        m.external_dropout_masks, ema.external_dropout_masks = T._device_masks(ms), T._device_masks(mt)
        got.append(step.step(xl.to(T.DEV), yl.to(T.DEV), xu.to(T.DEV), k, cons_w=T.CONS_W)["loss"])
    got = torch.stack(got).cpu().numpy()
    dl = np.abs(got - losses)
    ks = sorted(set([0, 1, 2, 4] + list(range(9, iters, max(1, iters // 10))) + [iters - 1]))
    print(f"{math} loss trace (iteration: oracle / HIP): " + "  ".join(f"{k + 1}: {losses[k]:.5f} / {got[k]:.5f}" for k in ks if k < iters), flush=True)
    print(f"{math}: max|dloss| {dl.max():.3e} (first 25: {dl[:25].max():.2e}, last 25: {dl[-25:].max():.2e}); loss last 5 {got[-5:].mean():.4f}", flush=True)
    m.eval(), ema.eval()
    m.external_dropout_masks = ema.external_dropout_masks = None
    for net, who in ((m, "student"), (ema, "teacher")):
        with torch.no_grad():
            o = net(xe.to(T.DEV)).cpu()
        d = losses_ref.mean_foreground_dice(o.argmax(1).numpy(), ye.numpy(), 4)
        print(f"{math} {who}: Dice {d:.5f} (delta {d - dref[who]:+.2e}), max|dlogit| {float((o - ref[who]).abs().max()):.3e}", flush=True)
