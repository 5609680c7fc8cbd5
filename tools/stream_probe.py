"""What a plain streaming kernel reaches on this box at the thin layers' sizes: torch's elementwise kernels (y = x + 1, y.copy_(x)) over
N x 224 x 224 x 16 fp32 (51.4 MB in, 51.4 MB out at N = 16), timed with events over 200 launches -- the practical ceiling to hold the
30 us of conv_thin_kernel<1,1,BNACT> (same bytes) against.   usage: python tools/stream_probe.py > profiles/r05_stream_probe.txt"""
import torch
dev = torch.device("cuda:0")
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print("# bytes_in+out MB | kernel | us | TB/s")
for nimg in (8, 16, 32, 64):
    x = torch.randn(nimg, 224, 224, 16, device=dev); y = torch.empty_like(x)
    mb = 2 * x.numel() * 4 / 1e6
    for name, fn in (("add", lambda: torch.add(x, 1.0, out=y)), ("copy", lambda: y.copy_(x)), ("read-only sum", lambda: x.sum())):
        us = t(fn)
        b = mb if name != "read-only sum" else mb / 2
        print(f"{b:8.1f} MB  N={nimg:3d} {name:14s} {us:8.2f} us  {b / us:6.2f} TB/s")
    # a chain: z = x + 1 ; y = z + 1 (the second reads what the first just wrote: MALL / L2 reuse as inside a step)
    z = torch.empty_like(x)
    us = t(lambda: (torch.add(x, 1.0, out=z), torch.add(z, 1.0, out=y)))
    print(f"{2 * mb:8.1f} MB  N={nimg:3d} {'add -> add':14s} {us:8.2f} us  {2 * mb / us:6.2f} TB/s")
