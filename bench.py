#!/usr/bin/env python
"""Headline benchmark: labeled+unlabeled images/sec/node of the Mean-Teacher U-Net step (BASELINE.json configs[1]):
U-Net(1->4 classes) at 224x224, 8 labelled + 8 unlabelled images per GPU, student forward+backward, train-mode teacher forward,
CE+Dice+MSE loss, SGD, EMA -- every step of 2017_03_NIPS_Mean-Teacher_ACDC.py:82-113 inside the timed region, synthetic data.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-graph] [--no-cpu-baseline] [--math bf16x3|f32] [--sync-bn]

`python bench.py --gpus N` with N > 1 and no launcher environment starts `python -m torch.distributed.run --nproc-per-node N`
on this file as a CHILD process (before anything here touches the GPU) and relays its JSON line; under a launcher (RANK /
WORLD_SIZE set) it is one rank of the job.  One rank per GPU over RCCL, weak scaling (per-GPU batch fixed).
Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (the longest launch of the step -- the fused backward kernel of
decoder.up4's first conv; its forward conv under `forward_conv_of_the_same_layer` -- timed with HIP events both alone and inside eager steps), `step_roofline`, `f32_math` (the same step with exact-fp32 MFMA products, N=1 only) and
`cpu_baseline` (the CPU oracle timed on this host's cores on a bounded sample).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import faulthandler

faulthandler.enable()      # a fault inside the HIP runtime still leaves the Python stack on stderr

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic traffic / flop constants (derived from model/unet.py:61-117 at 224x224, 1ch -> 4 classes)
IN_MB, OUT_MB, W_MB, LOSS_MB = 32.21, 26.79, 7.24, 2.5
GFLOP_FWD_IMG = 4.517
GFLOP_TRAIN_IMG = 13.54
DTYPE = {"bf16x3": "bf16x3 (fp32 storage; products as split-bf16 MFMA hi*hi+hi*lo+lo*hi, fp32 accumulate)",
         "f32": "f32 (fp32 storage, exact fp32-input MFMA)"}


def algorithmic_bytes_mt(n_img):
    train = n_img * (3 * IN_MB + 5 * OUT_MB) + 7 * W_MB
    teacher = n_img * (IN_MB + OUT_MB) + W_MB
    return (train + teacher + n_img * LOSS_MB) * 1e6


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32-line", action="store_true", help="skip the secondary exact-fp32 measurement")
    ap.add_argument("--workload", default="mt", choices=["mt"])
    ap.add_argument("--math", default=os.environ.get("HPFG_MATH", "bf16x3"), choices=["bf16x3", "f32"])
    ap.add_argument("--lab", type=int, default=8)
    ap.add_argument("--unlab", type=int, default=8)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--force-sync", action="store_true", help="diagnostics: run the data-parallel code path (RCCL collectives) on one rank")
    ap.add_argument("--sync-bn", action="store_true", help="N > 1: all-reduce every BatchNorm statistic and the loss sums (R ranks == one process on the "
                    "global batch) instead of the default per-rank BatchNorm + averaged gradients (DDP semantics)")
    ap.add_argument("--overlap", action="store_true",
                    help="N > 1: bucketed gradient all-reduce overlapped with the encoder half of backward (a chain of three hipGraphs around two eager "
                         "RCCL calls).  Default is ONE all-reduce between two hipGraphs: measured on one rank (nccl, this flag vs none) the chain costs "
                         "130-250 us per step -- it cannot queue the decoder's weight gradients onto the side stream and pays two more graph "
                         "launches -- which is more than the 7.26 MB all-reduce it would hide takes over xGMI")
    ap.add_argument("--no-overlap", action="store_true", help="(the default; kept for older command lines)")
    return ap.parse_args()


def relaunch(a):
    """--gpus N > 1 outside a launcher: run the N ranks as a child job and relay its output (no exec: this process may not have
    touched the GPU yet, but a child keeps that true by construction)."""
    port = os.environ.get("MASTER_PORT") or str(29500 + (os.getpid() % 2000))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("MASTER_PORT", None)
    r = subprocess.run(cmd, env=env)
    raise SystemExit(r.returncode)


def build_step(dev, a, math, dp):
    from copy import deepcopy
    from hpfg_amd.model import build_model
    from hpfg_amd.train import MeanTeacherStep
    from hpfg_amd.utils import loadyaml
    import torch
    args = loadyaml(os.path.join(ROOT, "config", "mean_teacher_unet_30k_224x224_ACDC.yaml"))
    args.batch_size, args.unlabel_batch_size = a.lab, a.unlab
    torch.manual_seed(args.seed)
    model = build_model(args).to(dev)
    model.math = math
    ema = deepcopy(model)
    for p in ema.parameters():
        p.requires_grad = False
    model.train()
    ema.train()
    return model, ema, MeanTeacherStep(model, ema, args, dp)


def timed_run(step, inputs, a, dev, dp, use_graph, steps, warmup):
    """W untimed + K timed steps bracketed by barrier + synchronize; returns (seconds, graph actually used, last iteration)."""
    import torch
    from hpfg_amd.train import GraphedStep
    runner, it = None, 0
    if use_graph:
        try:
            runner = GraphedStep(step, list(inputs), warmup=3, alias_inputs=True)   # the batch is resident in HBM at fixed addresses
            it = 3
        except Exception as e:  # capture unsupported: fall back to eager launches (still the HIP path)
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            runner, use_graph = None, False

    def one(i):
        if runner is not None:
            runner.step(list(inputs), i)
        else:
            step.step(*inputs, i)

    def barrier():
        if dp is not None:
            dp.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(warmup):
        it += 1
        one(it)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        it += 1
        one(it)
    barrier()
    dt = time.perf_counter() - t0
    if dp is not None:
        dt = dp.max_float(dt)
    return dt, use_graph, it


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and "RANK" not in os.environ:
        relaunch(a)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    if os.environ.get("HPFG_BENCH_ONE_DEVICE", "0") == "1":      # rehearsal: several ranks on one GPU (with HPFG_DP_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from hpfg_amd import parallel
    from hpfg_amd.datasets.synthetic import synth_batch

    dp = None
    if world > 1 or a.force_sync:
        if a.force_sync and world == 1:      # documented single-process diagnostic: supply the rendezvous a launcher would
            for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29511"), ("RANK", "0"), ("WORLD_SIZE", "1")):
                os.environ.setdefault(k, v)
        dp = parallel.init_from_env(dev, backend=os.environ.get("HPFG_DP_BACKEND") or None)
        dp.force_sync = bool(a.force_sync)
        dp.sync_bn = bool(a.sync_bn or (a.force_sync and os.environ.get("HPFG_BENCH_LOCAL_BN", "0") != "1"))
        dp.overlap = bool(a.overlap) and not a.no_overlap
    model, ema, step = build_step(dev, a, a.math, dp)
    xl, yl = synth_batch(1234 + rank, a.lab, a.size, a.size, 1, 4, 32)
    xu, _ = synth_batch(91234 + rank, a.unlab, a.size, a.size, 1, 4, 32)
    xl, yl, xu = xl.to(dev), yl.to(dev), xu.to(dev)
    from hpfg_amd.train import batch_pair
    xl, xu = batch_pair(xl, xu)          # labelled and unlabelled images back to back in HBM: the step's batch is a view, not a concat copy

    # N > 1, default mode: two graphs around the gradient all-reduce (no RCCL node inside a hipGraph).  --sync-bn (collectives
    # between the kernels of forward and backward) runs eager unless HPFG_DP_GRAPH=1 asks for a capture with RCCL nodes.
    sync_mode = dp is not None and dp.sync_bn and (world > 1 or a.force_sync)
    use_graph = (not a.no_graph) and (not sync_mode or os.environ.get("HPFG_DP_GRAPH", "0") == "1")
    dt, use_graph, it = timed_run(step, (xl, yl, xu), a, dev, dp, use_graph, a.steps, a.warmup)
    n_img = a.lab + a.unlab
    ms = dt / a.steps * 1e3
    value = n_img * world / (dt / a.steps)

    roof = f32 = cpu = None
    if rank == 0:
        fwd = dominant_kernel_roofline(model, step, (xl, yl, xu), it, dev, eager_ok=dp is None)
        roof = fused_bwd_roofline(model, step, (xl, yl, xu), it, dev, eager_ok=dp is None)
        if roof is None:
            roof = fwd
        else:
            roof["forward_conv_of_the_same_layer"] = fwd
    if world == 1 and dp is None and not a.no_f32_line and a.math != "f32":
        del step, model, ema
        torch.cuda.empty_cache()
        m2, e2, s2 = build_step(dev, a, "f32", None)
        k2 = max(5, a.steps // 2)
        dt2, g2, _ = timed_run(s2, (xl, yl, xu), a, dev, None, use_graph, k2, max(3, a.warmup // 2))
        f32 = {"dtype": DTYPE["f32"], "value": round(n_img / (dt2 / k2), 2), "unit": "images/s", "ms_per_step": round(dt2 / k2 * 1e3, 4), "steps": k2,
               "hipgraph": bool(g2), "note": "same step, HPFG_MATH=f32: every product exact fp32 (v_mfma_f32_16x16x4_f32); meets 1e-3 on every fixture"}
        del m2, e2, s2
    if rank == 0 and world == 1 and not a.no_cpu_baseline:      # reported baseline: rank 0 at N=1 only
        cpu = cpu_baseline(a.lab, a.unlab, a.size)
    if rank == 0:
        step_bytes = algorithmic_bytes_mt(n_img)
        par = f"dp{world}"
        if world > 1:
            par += " (sync BatchNorm + loss sums: == one process on the global batch)" if (dp is not None and dp.sync_bn) else \
                (" (per-rank BatchNorm, gradients averaged by bucketed all-reduces overlapped with backward)" if dp is not None and dp.overlap else
                 " (per-rank BatchNorm, gradients averaged by one all-reduce between two hipGraphs)")
        out = {
            "metric": "labeled+unlabeled images/sec/node, U-Net 224x224 ACDC-shaped (Mean-Teacher step)", "value": round(value, 2),
            "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE[a.math], "data": "synthetic",
            "config": {"workload": "mean_teacher_unet_224x224 (BASELINE configs[1]): U-Net 1ch->4cls, 8 labelled + 8 unlabelled per GPU, "
                                   "student fwd+bwd + train-mode teacher fwd + CE/Dice/MSE + SGD + EMA",
                       "per_gpu_batch": [a.lab, a.unlab], "size": a.size, "hipgraph": bool(use_graph), "sync_bn": bool(dp is not None and dp.sync_bn),
                       "parallelism": par, "math": a.math},
            "step_roofline": {"algorithmic_GB_per_step": round(step_bytes / 1e9, 3), "achieved_GBps": round(step_bytes / (dt / a.steps) / 1e9, 1),
                              "frac_of_8TBps": round(step_bytes / (dt / a.steps) / 8e12, 4),
                              "algorithmic_GFLOP_per_step": round(n_img * (GFLOP_TRAIN_IMG + GFLOP_FWD_IMG), 1),
                              "achieved_TFLOPs": round(n_img * (GFLOP_TRAIN_IMG + GFLOP_FWD_IMG) / (dt / a.steps) / 1e3, 2)},
            "roofline": roof, "f32_math": f32, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dp is not None:
        dp.shutdown()


def fused_bwd_roofline(model, step, inputs, it, dev, eager_ok=True):
    """The longest launch of the step since round 2: the fused backward kernel of decoder.up4's first conv (32 -> 16 channels at 224x224):
    input gradient + weight-gradient slabs from one staging of dZ.  Timed like dominant_kernel_roofline(): HIP events around the launch
    inside eager steps (teacher stream idle during backward, so in-step == what the kernel trace shows), and 20 launches alone.
    Algorithmic bytes per pixel: dA + z of the layer (2 x 16 ch), the skip tensor (16 ch), the low-res 1x1 output (16 ch at a quarter of
    the pixels), dX written (32 ch) = 336 B.  None when the layer is not on the fused path (f32 math, HPFG_FUSED_BWD=0)."""
    import ctypes as C
    import torch
    from hpfg_amd import _lib as L
    eng = next(iter(model._engines.values()))[0]
    name = "decoder.up4.conv.conv_conv.0"
    if name not in eng.fused_grid or name not in eng._last_fused:
        return None
    s = eng.specs[name]
    in_step_us = raw_us = bracket_us = None
    if eager_ok:
        eng.probe = ("fused_bwd:" + name, [], [])
        for k in range(6):
            step.step(*inputs, it + 1 + k)
        torch.cuda.synchronize(dev)
        ts = [e0.elapsed_time(e1) * 1e3 for (e0, e1) in eng.probe[1][len(eng.probe[1]) // 3:]]
        tb = [e0.elapsed_time(e1) * 1e3 for (e0, e1) in eng.probe[2][len(eng.probe[2]) // 3:]]
        eng.probe = None
        if ts:
            # an event pair costs a few microseconds by itself (two marker packets): the same bracket recorded around NOTHING, right in
            # front of the launch, is subtracted -- what is left agrees with the kernel's duration in the rocprofv3 trace of the captured step
            raw_us = sum(ts) / len(ts)
            bracket_us = sum(tb) / len(tb) if tb else 0.0
            in_step_us = raw_us - bracket_us
    fa = eng._last_fused[name]
    st = torch.cuda.current_stream(dev)
    lib = L.load()
    for _ in range(3):
        L.check(lib.hpfg_fused_bwd(C.byref(fa), st.cuda_stream), "fused_bwd")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record(st)
    for _ in range(reps):
        L.check(lib.hpfg_fused_bwd(C.byref(fa), st.cuda_stream), "fused_bwd")
    e1.record(st)
    e1.synchronize()
    solo_us = e0.elapsed_time(e1) / reps * 1e3
    us = in_step_us if in_step_us is not None else solo_us
    n = eng.N
    bytes_alg = n * s.h * s.w * (2 * 16 + 16 + 32) * 4 + n * (s.h // 2) * (s.w // 2) * 16 * 4
    flops = n * s.h * s.w * 9 * 32 * 16 * 2 * 2
    ach = bytes_alg / (us * 1e-6) / 1e9
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_fused_bwd_kernel.json")))
        if n == 16 and s.h == 224:
            traffic = int(tj["hbm_bytes_per_launch"])
    except Exception:
        traffic = None
    return {"kernel": "fused_bwd_kernel<2 input x 1 output channel tiles, concat input, dZ source, 8 waves> @ decoder.up4.conv.conv_conv.0 (32->16ch, "
                      "224x224): input gradient + weight-gradient slabs from one staging of dZ = k1*g + k2*z + k3 (BatchNorm / LeakyReLU backward on load)",
            "bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": traffic,
            "avg_launch_us": round(us, 2), "timing": ("in-step (HIP events around the launch inside eager steps, minus the same event bracket around nothing: "
                                                      f"{raw_us:.2f} - {bracket_us:.2f} us)") if in_step_us is not None else "solo",
            "solo_launch_us": round(solo_us, 2), "frac_solo": round(bytes_alg / (solo_us * 1e-6) / 8e12, 4),
            "algorithmic_bytes_per_launch": bytes_alg, "flops_per_launch": flops, "achieved_TFLOPs": round(flops / (us * 1e-6) / 1e12, 2),
            "note": "the longest launch of the step (backward of the layer whose forward was round 1's roofline kernel); no single kernel dominates "
                    "(largest family 19 % of kernel time): step_roofline is the figure that matters"}


def dominant_kernel_roofline(model, step, inputs, it, dev, eager_ok=True):
    """The heaviest-traffic conv launch of the step (decoder.up4 first conv: 32ch->16ch at 224x224, reads the skip tensor + the
    upsampled 1x1 output, writes 16ch), timed with HIP events on the stream it is launched on, two ways: (a) inside real steps --
    the engine records events around that launch of the student's and of the teacher's forward while a few EAGER steps run (the
    other network's kernels run concurrently on the second stream, as in the timed region) -- and (b) alone, 20 back-to-back
    launches.  `achieved` / `frac` use (a): the in-step figure is the one the rocprofv3 kernel trace under profiles/ reproduces."""
    import ctypes as C
    import torch
    from hpfg_amd import _lib as L
    eng = next(iter(model._engines.values()))[0]
    name = "decoder.up4.conv.conv_conv.0"
    s = eng.specs[name]
    in_step_us = None
    if eager_ok:
        engines = [e for m in (model, step.ema_model) for pool in m._engines.values() for e in pool]
        for e in engines:
            e.probe = (name, [], [])
        for k in range(6):
            step.step(*inputs, it + 1 + k)
        torch.cuda.synchronize(dev)
        ts = [e0.elapsed_time(e1) * 1e3 for e in engines for (e0, e1) in e.probe[1][len(e.probe[1]) // 3:]]     # first third = warm-up
        tb = [e0.elapsed_time(e1) * 1e3 for e in engines for (e0, e1) in e.probe[2][len(e.probe[2]) // 3:]]     # the event bracket around nothing
        for e in engines:
            e.probe = None
        if ts:
            in_step_us = sum(ts) / len(ts) - (sum(tb) / len(tb) if tb else 0.0)
    a0, a1 = eng.input_acts(name)
    ca = L.ConvArgs()
    ca.a0, ca.a1 = a0, a1
    ca.math = eng.math
    ca.wpk = L.ptr(eng.wpk16_f[name]) if eng.math == L.MATH_BF16X3 else L.ptr(eng.wpk_f[name])
    ca.bias, ca.out = L.ptr(eng.bias_pad[name]), L.ptr(eng.z[name])
    ca.stat_partials = L.ptr(eng.partials)
    ca.out_pstride, ca.Cout, ca.CoutPad, ca.N, ca.H, ca.W, ca.taps = s.cout, s.cout, s.cout_pad, eng.N, s.h, s.w, 9
    st = torch.cuda.current_stream(dev)
    lib = L.load()
    for _ in range(3):
        L.check(lib.hpfg_conv_fwd(C.byref(ca), st.cuda_stream), "conv")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record(st)
    for _ in range(reps):
        L.check(lib.hpfg_conv_fwd(C.byref(ca), st.cuda_stream), "conv")
    e1.record(st)
    e1.synchronize()
    solo_us = e0.elapsed_time(e1) / reps * 1e3
    us = in_step_us if in_step_us is not None else solo_us
    n = eng.N
    # algorithmic bytes of this launch: skip 16ch@224 + 1x1 output 16ch@112 read, 16ch@224 written, weights
    bytes_alg = n * (s.h * s.w * 16 + (s.h // 2) * (s.w // 2) * 16 + s.h * s.w * 16) * 4 + 9 * 32 * 16 * 4
    flops = n * s.h * s.w * 9 * 32 * 16 * 2
    ach = bytes_alg / (us * 1e-6) / 1e9
    traffic = None
    try:      # HBM bytes of this launch from the PMC passes committed under profiles/ (FETCH_SIZE x2 + WRITE_SIZE, see the file)
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_dominant_kernel.json")))
        if n == 16 and s.h == 224 and eng.math == L.MATH_BF16X3:
            traffic = int(tj["hbm_bytes_per_launch"])
    except Exception:
        traffic = None
    kname = "conv_thin_kernel" if eng.math == L.MATH_BF16X3 else "conv_mfma_kernel"
    return {"kernel": kname + "<16x16 tile, 16 output channels, 3x3, CAT loader> @ decoder.up4.conv.conv_conv.0 (32->16ch, 224x224, skip concat + "
                      "bilinear upsample + BN + LeakyReLU fused on load, BN partial sums in the epilogue)",
            "bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": traffic,
            "avg_launch_us": round(us, 2),
            "timing": "in-step (HIP events around the launch inside eager steps, net of the same event bracket around nothing)" if in_step_us is not None else "solo",
            "solo_launch_us": round(solo_us, 2), "frac_solo": round(bytes_alg / (solo_us * 1e-6) / 8e12, 4),
            "algorithmic_bytes_per_launch": bytes_alg, "flops_per_launch": flops, "achieved_TFLOPs": round(flops / (us * 1e-6) / 1e12, 2),
            "note": "no single kernel dominates the step (largest template instance ~7 % of kernel time): step_roofline is the figure that matters"}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(n_lab, n_unlab, size):
    """CPU oracle (plain PyTorch restatement of the reference step, oracle/steps_ref.py) on this host's cores, SURVEY.md section 8(d)
    protocol: 3 warm-up + 10 timed Mean-Teacher steps, median."""
    import torch
    from hpfg_amd.datasets.synthetic import synth_batch
    from oracle import laws_ref, steps_ref, unet_ref
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))      # a 1-GPU box shares its host: 16 cores is this job's CPU share
    torch.set_num_threads(cores)
    st = unet_ref.init_state(1337, 1, 4)
    ema = unet_ref.clone_state(st)
    xl, yl = synth_batch(1234, n_lab, size, size, 1, 4, 32)
    xu, _ = synth_batch(91234, n_unlab, size, size, 1, 4, 32)
    bufs = {}
    times = []
    warm, timed = 3, 10
    for k in range(1, warm + timed + 1):
        t0 = time.perf_counter()
        steps_ref.mean_teacher_step(st, ema, bufs, xl, yl.long(), xu, laws_ref.medical_lr(k, 0.01, 30000), 0.0, laws_ref.ema_alpha(k, 0.99))
        times.append(time.perf_counter() - t0)
        if k > warm and sum(times[warm:]) > 40.0:      # bound the sample on a slow host
            break
    ts = sorted(times[warm:])
    t = ts[len(ts) // 2]
    return {"value": round((n_lab + n_unlab) / t, 2), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": _cpu_model(), "os_cpu_count": os.cpu_count(), "affinity_cores": avail,
            "sample": f"{warm} warm-up + {len(ts)} timed Mean-Teacher steps of {n_lab}+{n_unlab} images at {size}x{size}, median, torch CPU fp32, "
                      f"{torch.get_num_threads()} threads"}


if __name__ == "__main__":
    main()
