"""Minimal hipGraph captures with forked streams, each in a child process (a fault in hipStreamEndCapture must not end the probe):
  flat        origin -> A, origin -> B, both joined into the origin                      (the Mean-Teacher step's shape)
  nested      origin -> A -> B (B forked from the FORKED stream A), B joined into A, A into the origin
  nested+     the same, and the origin also waits for B directly
  nested-thr  the inner fork / join issued from a second host thread (autograd runs backward nodes on its own thread)
  sibling     A and B BOTH forked from the origin up front; later B waits for A (a cross edge between two parallel streams, not a fork),
              A waits for B, and the origin joins both directly     (-thr: the cross edges from a second thread; -transitive: B only via A)
usage: python tools/nested_fork_probe.py            (parent)   |   python tools/nested_fork_probe.py <variant>   (child)"""
import subprocess
import sys
import threading

VARIANTS = ["flat", "nested", "nested+", "nested-thr", "sibling", "sibling-thr", "sibling-transitive"]


def child(v):
    import faulthandler
    faulthandler.enable()
    import torch
    dev = torch.device("cuda:0")
    x = torch.ones(1 << 20, device=dev)
    y = torch.zeros_like(x)
    z = torch.zeros_like(x)
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        g.capture_begin(capture_error_mode="thread_local")
        x.mul_(2.0)
        if v == "flat":
            a.wait_stream(cap)
            b.wait_stream(cap)
            with torch.cuda.stream(a):
                y.add_(x)
            with torch.cuda.stream(b):
                z.add_(x)
            cap.wait_stream(a)
            cap.wait_stream(b)
        elif v.startswith("sibling"):
            a.wait_stream(cap)
            b.wait_stream(cap)

            def inner():
                with torch.cuda.stream(a):
                    y.add_(x)
                    b.wait_stream(a)
                    with torch.cuda.stream(b):
                        z.add_(y)
                    y.mul_(3.0)
                    a.wait_stream(b)
            if v == "sibling-thr":
                t = threading.Thread(target=inner)
                t.start()
                t.join()
            else:
                inner()
            cap.wait_stream(a)
            if v != "sibling-transitive":
                cap.wait_stream(b)
        else:
            a.wait_stream(cap)

            def inner():
                with torch.cuda.stream(a):
                    y.add_(x)
                    b.wait_stream(a)
                    with torch.cuda.stream(b):
                        z.add_(y)
                    y.mul_(3.0)
                    a.wait_stream(b)
            if v == "nested-thr":
                t = threading.Thread(target=inner)
                t.start()
                t.join()
            else:
                inner()
            cap.wait_stream(a)
            if v == "nested+":
                cap.wait_stream(b)
        x.add_(y)
        g.capture_end()
    torch.cuda.current_stream().wait_stream(cap)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print(v, "ok", float(x[0]), float(y[0]), float(z[0]), flush=True)


if len(sys.argv) > 1:
    child(sys.argv[1])
else:
    for v in VARIANTS:
        r = subprocess.run([sys.executable, __file__, v], capture_output=True, text=True, timeout=120)
        tail = (r.stdout.strip().splitlines() or [""])[-1]
        err = [ln for ln in r.stderr.splitlines() if "Fatal" in ln or "Error" in ln or "capture_end" in ln]
        print(f"{v:11s} rc={r.returncode} {tail} {' | '.join(err[:3])}", flush=True)
