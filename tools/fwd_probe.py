"""Forward-only graph replay of the U-Net engine (diagnostics): us per train-mode forward of FWD_N (default 8) x 224^2 images.
Under `rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/fwd_probe.py`, `python tools/trace_gaps.py DIR` lists the last forward kernel by kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpfg_amd.datasets.synthetic import synth_batch
from hpfg_amd.model import UNet, reset_dropout_streams
DEV = torch.device("cuda:0")
for n in (int(os.environ.get('FWD_N', '8')),):
    reset_dropout_streams(); torch.manual_seed(11)
    m = UNet(1, 4).to(DEV); m.train()
    x, _ = synth_batch(5, n, 224, 224, 1, 4, cell=8)
    x = x.to(DEV)
    for grad in (False,):
        ctx = torch.enable_grad() if grad else torch.no_grad()
        with ctx:
            for _ in range(3): m(x)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                with torch.cuda.graph(g, stream=s):
                    y = m(x)
        torch.cuda.synchronize()
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): g.replay()
        b.record(); torch.cuda.synchronize()
        print(f"N={n} needs_grad={grad}: {a.elapsed_time(b) / 50 * 1e3:8.1f} us per forward", flush=True)
