#!/usr/bin/env python
"""Headline benchmark: labeled+unlabeled images/sec/node of the Mean-Teacher U-Net step (BASELINE.json configs[1]):
U-Net(1->4 classes) at 224x224, 8 labelled + 8 unlabelled images per GPU, student forward+backward, train-mode teacher forward,
CE+Dice+MSE loss, SGD, EMA -- every step of 2017_03_NIPS_Mean-Teacher_ACDC.py:82-113 inside the timed region, fp32, synthetic data.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-graph] [--no-cpu-baseline] [--workload mt|hpfg|sup]

N>1 is launched by torch.distributed.run (one rank per GPU, RCCL): weak scaling, per-GPU batch fixed; gradients, BatchNorm
statistics and Dice/CE sums are all-reduced so the arithmetic equals a single process on the global batch.
Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (dominant kernel, measured with HIP events) and
`cpu_baseline` (the CPU oracle timed on this host's cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

# algorithmic traffic / flop constants (derived from model/unet.py:61-117 at 224x224, 1ch -> 4 classes)
IN_MB, OUT_MB, W_MB, LOSS_MB = 32.21, 26.79, 7.24, 2.5
GFLOP_FWD_IMG = 4.517
GFLOP_TRAIN_IMG = 13.54


def algorithmic_bytes_mt(n_img):
    train = n_img * (3 * IN_MB + 5 * OUT_MB) + 7 * W_MB
    teacher = n_img * (IN_MB + OUT_MB) + W_MB
    return (train + teacher + n_img * LOSS_MB) * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="mt", choices=["mt"])
    ap.add_argument("--lab", type=int, default=8)
    ap.add_argument("--unlab", type=int, default=8)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--force-sync", action="store_true", help="diagnostics: run the data-parallel code path (RCCL collectives) on one rank")
    ap.add_argument("--sync-bn", action="store_true", help="N > 1: all-reduce every BatchNorm statistic and the loss sums (R ranks == one process on the "
                    "global batch) instead of the default per-rank BatchNorm + averaged gradients (DDP semantics)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N>1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if os.environ.get("HPFG_BENCH_ONE_DEVICE", "0") == "1":      # rehearsal: several ranks on one GPU (with HPFG_DP_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from copy import deepcopy
    from hpfg_amd import parallel
    from hpfg_amd.datasets.synthetic import synth_batch
    from hpfg_amd.model import build_model
    from hpfg_amd.train import GraphedStep, MeanTeacherStep
    from hpfg_amd.utils import loadyaml

    dp = parallel.init_from_env(dev, backend=os.environ.get("HPFG_DP_BACKEND") or None) if (world > 1 or a.force_sync) else None
    if a.force_sync and dp is not None:
        os.environ.setdefault("MASTER_PORT", "29511")
        dp.force_sync = True
    if dp is not None:
        dp.sync_bn = bool(a.sync_bn or (a.force_sync and os.environ.get("HPFG_BENCH_LOCAL_BN", "0") != "1"))
    args = loadyaml(os.path.join(ROOT, "config", "mean_teacher_unet_30k_224x224_ACDC.yaml"))
    args.batch_size, args.unlabel_batch_size = a.lab, a.unlab
    torch.manual_seed(args.seed)
    model = build_model(args).to(dev)
    ema = deepcopy(model)
    for p in ema.parameters():
        p.requires_grad = False
    model.train()
    ema.train()
    step = MeanTeacherStep(model, ema, args, dp)
    xl, yl = synth_batch(1234 + rank, a.lab, a.size, a.size, 1, 4, 32)
    xu, _ = synth_batch(91234 + rank, a.unlab, a.size, a.size, 1, 4, 32)
    xl, yl, xu = xl.to(dev), yl.to(dev), xu.to(dev)

    # N > 1 runs eager (the RCCL watchdog rejects stream capture here; HPFG_DP_GRAPH=1 forces the attempt) with per-rank BatchNorm
    # and one gradient all-reduce per step; --sync-bn adds the ~90 small BatchNorm / loss collectives of the exact global-batch mode
    # N > 1, default mode: two graphs around one eager gradient all-reduce (no RCCL node inside a hipGraph).  --sync-bn (collectives
    # between the kernels of forward and backward) runs eager unless HPFG_DP_GRAPH=1 asks for a capture with RCCL nodes.
    sync_mode = dp is not None and dp.sync_bn and (world > 1 or a.force_sync)
    use_graph = (not a.no_graph) and (not sync_mode or os.environ.get("HPFG_DP_GRAPH", "0") == "1")
    runner = None
    it = 0
    if use_graph:
        try:
            runner = GraphedStep(step, [xl, yl, xu], warmup=3, alias_inputs=True)   # the batch is resident in HBM at fixed addresses
            it = 3
        except Exception as e:  # capture unsupported: fall back to eager launches (still the HIP path)
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            runner, use_graph = None, False

    def one(i):
        if runner is not None:
            runner.step([xl, yl, xu], i)
        else:
            step.step(xl, yl, xu, i)

    for _ in range(a.warmup):
        it += 1
        one(it)

    def barrier():
        if dp is not None:
            dp.barrier()
        torch.cuda.synchronize(dev)

    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        it += 1
        one(it)
    barrier()
    dt = time.perf_counter() - t0
    if dp is not None:
        dt = dp.max_float(dt)
    n_img = a.lab + a.unlab
    ms = dt / a.steps * 1e3
    value = n_img * world / (dt / a.steps)

    # ---- roofline of the dominant kernel, timed live with HIP events on the launch stream ------------------------------
    roof = None
    if rank == 0:
        roof = dominant_kernel_roofline(model, xl, xu, dev)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:      # reported baseline: rank 0 at N=1 only
        cpu = cpu_baseline(a.lab, a.unlab, a.size)
    if rank == 0:
        step_bytes = algorithmic_bytes_mt(n_img)
        out = {
            "metric": "labeled+unlabeled images/sec/node, U-Net 224x224 ACDC-shaped (Mean-Teacher step)", "value": round(value, 2),
            "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "mean_teacher_unet_224x224 (BASELINE configs[1]): U-Net 1ch->4cls, 8 labelled + 8 unlabelled per GPU, "
                                   "student fwd+bwd + train-mode teacher fwd + CE/Dice/MSE + SGD + EMA",
                       "per_gpu_batch": [a.lab, a.unlab], "size": a.size, "hipgraph": bool(use_graph), "sync_bn": bool(dp is not None and dp.sync_bn), "parallelism": f"dp{world}" + (" (per-rank BatchNorm, averaged gradients)" if world > 1 and not (dp is not None and dp.sync_bn) else ""),
                       "math": model.math + (" (split-bf16 MFMA products hi*hi+hi*lo+lo*hi, fp32 accumulate; parity 1e-3 verified by "
                                             "tests/test_gpu_steps.py::test_mean_teacher_trace_224_vs_oracle)" if model.math == "bf16x3" else " (exact fp32 MFMA)")},
            "step_roofline": {"algorithmic_GB_per_step": round(step_bytes / 1e9, 3), "achieved_GBps": round(step_bytes / (dt / a.steps) / 1e9, 1),
                              "frac_of_8TBps": round(step_bytes / (dt / a.steps) / 8e12, 4),
                              "algorithmic_GFLOP_per_step": round(n_img * (GFLOP_TRAIN_IMG + GFLOP_FWD_IMG), 1),
                              "achieved_TFLOPs": round(n_img * (GFLOP_TRAIN_IMG + GFLOP_FWD_IMG) / (dt / a.steps) / 1e3, 2)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dp is not None:
        dp.shutdown()


def dominant_kernel_roofline(model, xl, xu, dev):
    """Times the heaviest-traffic conv launch of the step (decoder.up4 first conv: 32ch->16ch at 224x224, reads the skip
    tensor + upsampled 1x1 output, writes 16ch) with HIP events on the stream it is launched on."""
    import ctypes as C
    from hpfg_amd import _lib as L
    eng = next(iter(model._engines.values()))[0]
    name = "decoder.up4.conv.conv_conv.0"
    s = eng.specs[name]
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
    us = e0.elapsed_time(e1) / reps * 1e3
    n = eng.N
    # algorithmic bytes of this launch: skip 16ch@224 + 1x1 output 16ch@112 read, 16ch@224 written, weights
    bytes_alg = n * (224 * 224 * 16 + 112 * 112 * 16 + 224 * 224 * 16) * 4 + 9 * 32 * 16 * 4
    flops = n * 224 * 224 * 9 * 32 * 16 * 2
    ach = bytes_alg / (us * 1e-6) / 1e9
    traffic = None
    try:      # HBM bytes of this launch from the PMC passes committed under profiles/ (FETCH_SIZE x2 + WRITE_SIZE, see the file)
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_dominant_kernel.json")))
        if n == 16 and eng.math == L.MATH_BF16X3:
            traffic = int(tj["hbm_bytes_per_launch"])
    except Exception:
        traffic = None
    kname = "conv_bf16x3_kernel" if eng.math == L.MATH_BF16X3 else "conv_mfma_kernel"
    return {"kernel": kname + "<16x16 tile, 16 output channels, 3x3, CAT loader> @ decoder.up4.conv.conv_conv.0 (32->16ch, 224x224, skip concat + "
                      "bilinear upsample + BN + LeakyReLU fused on load, BN partial sums in the epilogue)",
            "bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": traffic,
            "avg_launch_us": round(us, 2), "algorithmic_bytes_per_launch": bytes_alg,
            "flops_per_launch": flops, "achieved_TFLOPs": round(flops / (us * 1e-6) / 1e12, 2)}


def cpu_baseline(n_lab, n_unlab, size):
    """CPU oracle (plain PyTorch restatement of the reference step, oracle/steps_ref.py) on this host's cores: bounded sample."""
    from hpfg_amd.datasets.synthetic import synth_batch
    from oracle import laws_ref, steps_ref, unet_ref
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))      # a 1-GPU box shares its host: 16 cores is this job's CPU share
    torch.set_num_threads(cores)
    st = unet_ref.init_state(1337, 1, 4)
    ema = unet_ref.clone_state(st)
    xl, yl = synth_batch(1234, n_lab, size, size, 1, 4, 32)
    xu, _ = synth_batch(91234, n_unlab, size, size, 1, 4, 32)
    bufs = {}
    times = []
    for k in range(1, 4):
        t0 = time.perf_counter()
        steps_ref.mean_teacher_step(st, ema, bufs, xl, yl.long(), xu, laws_ref.medical_lr(k, 0.01, 30000), 0.0, laws_ref.ema_alpha(k, 0.99))
        times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[0] if len(times) > 1 else times[0]
    return {"value": round((n_lab + n_unlab) / t, 2), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"3 Mean-Teacher steps of {n_lab}+{n_unlab} images at {size}x{size} (1 warm-up, best of 2 timed), torch CPU fp32"}


if __name__ == "__main__":
    main()
