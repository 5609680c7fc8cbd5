import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_fullsize import _mt_run
runs = []
for r in range(2):
    m, e, st, out, _, res = _mt_run(steps=int(os.environ.get('STEPS','2')))
    torch.cuda.synchronize()
    snap = {}
    for tag, net in (("student", m), ("teacher", e)):
        eng = next(iter(net._engines.values()))[0]
        for k, v in eng.z.items():
            snap[f"{tag}.z.{k}"] = v.clone()
        for k, v in eng.bn.items():
            snap[f"{tag}.bn.{k}"] = v.clone()
    snap["logits"] = res["logits"].clone()
    snap["t_logits"] = res["t_logits"].clone()
    snap["grads"] = m.flat_grads.clone()
    snap["params"] = m.flat_params.clone()
    snap["ema"] = e.flat_params.clone()
    for i, o in enumerate(out):
        snap[f"parts{i}"] = o.clone()
    runs.append(snap)
bad = [k for k in runs[0] if not torch.equal(runs[0][k], runs[1][k])]
print("differing:", len(bad), "teacher keys:", len([k for k in bad if k.startswith("teacher")]), "first:", bad[:3])
for k in bad[:12]:
    d = (runs[0][k] - runs[1][k]).abs()
    print(k, float(d.max()), int((d > 0).sum()), "of", d.numel())
k = "student.z.encoder.in_conv.conv_conv.0"
if k in bad:
    d = (runs[0][k] - runs[1][k]).abs().amax(-1)      # [N,H,W]
    idx = d.nonzero()
    print("images:", sorted(set(idx[:, 0].tolist())))
    tiles = sorted(set((int(a), int(b) // 16, int(c) // 16) for a, b, c in idx.tolist()))
    print("tiles (n, ty, tx):", tiles[:40], len(tiles))
    n0, ty, tx = tiles[0]
    sub = d[n0, ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16]
    print((sub > 0).int())
