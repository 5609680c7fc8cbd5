#!/usr/bin/env python
"""Headline benchmark: labeled+unlabeled images/sec/node of one step of the reference's drivers on the MI355X hot path, synthetic data.

    python bench.py [--workload mt|sup|hpfg|cps|ctct] [--gpus N] [--steps K] [--warmup W] [--math bf16x3|f32] [--no-graph]
                    [--local-bn] [--no-overlap] [--rccl] [--no-cpu-baseline] [--no-f32-line] [--no-probe]

Workloads = BASELINE.json configs (the default, `mt`, is configs[1], the one the metric is quoted on):
  sup   configs[0]  sup_ACDC.py:83-93                          U-Net 1->4, 8 x 224^2, SGD + cosine schedule
  mt    configs[1]  2017_03_NIPS_Mean-Teacher_ACDC.py:82-113   U-Net 1->4, 8 + 8 x 224^2: student fwd+bwd, train-mode teacher fwd, CE/Dice/MSE, SGD, EMA
  hpfg  configs[2]  main.py:125-212                            U-Net+ x 2 + EMA teacher, 16 + 16 x 224^2 per GPU, CutMix, Dense_Loss
  cps   configs[3]  2021_06_CVPR_CPS_ACDC.py:95-120            U-Net 3->2 x 2, 32 + 32 x 96^2
  ctct  configs[4]  2021_12_MIDL_CTCT_ACDC.py:117-151          U-Net + SegFormer-B0, 8 + 24 x 224^2
Every statement of the loop body is inside the timed region; inputs are resident in HBM; one hipGraph per step at N = 1.

`python bench.py --gpus N` with N > 1 and no launcher environment starts `python -m torch.distributed.run --nproc-per-node N` on this
file as a CHILD process (before anything here touches the GPU) and relays its JSON line; under a launcher (RANK / WORLD_SIZE set) it is
one rank of the job.  One rank per GPU (process group over RCCL), weak scaling (per-GPU batch fixed).  N > 1 runs the mode BASELINE.json's
north_star names: gradients AND BatchNorm / loss sums cross the ranks (R ranks == one process on the global batch; the sums through peer
mailboxes inside the finalize kernels), the gradient exchange bucketed and overlapped with the encoder half of backward, everything through
IPC-mapped peer windows over xGMI inside the step's hipGraph (after a self-test on the node; `--rccl`: RCCL all-reduces between graphs).  The
same job then times the DistributedDataParallel-semantics mode (per-rank BatchNorm, averaged gradients: `--local-bn` makes it the headline)
and reports it beside the headline as `other_bn_mode`.

Prints ONE JSON line (rank 0): the driver's contract plus
  step_roofline   the whole step against its algorithmic bytes / FLOPs,
  roofline        the kernel family with the largest share of the step's kernel time, measured live: every conv / dgrad / wgrad /
                  BatchNorm launch of the U-Net engines is bracketed by device time stamps (one-wave kernels storing s_memrealtime) inside
                  a second captured hipGraph of the same step -- no host events, no eager steps -- and the families' times are summed per
                  step; `longest_launch` is kept beside it,
  blocks          the contract's K timed steps are block 0; four more blocks of K steps follow: ms/step of each, median, min, max,
  f32_math        the same step with exact-fp32 MFMA products (N = 1, `mt` only),
  cpu_baseline    the CPU oracle of the same step on this host's cores (bounded sample; all physical cores),
  exchange_path   N > 1: what every rank's sums actually travelled through (peer windows / mailboxes or the RCCL fallback and why), its
                  window mapping and self-test results and its peer error word.
"""
import argparse
import faulthandler
import json
import os
import subprocess
import sys
import time

faulthandler.enable()      # a fault inside the HIP runtime still leaves the Python stacks on stderr

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# ---- algorithmic traffic / flop constants (derived from model/unet.py:61-117) -----------------------------------
# per image: IN = sum of conv-input bytes, OUT = sum of conv-output bytes (MB, fp32); W = parameters (MB); GF = forward GFLOP
UNET = {224: dict(IN=32.21, OUT=26.79, GF=4.517), 96: dict(IN=5.99, OUT=4.85, GF=0.830)}
W_UNET, W_PLUS = 7.24, 14.65
DTYPE = {"bf16x3": "bf16x3 (fp32 storage; products as split-bf16 MFMA hi*hi+hi*lo+lo*hi, fp32 accumulate)",
         "f32": "f32 (fp32 storage, exact fp32-input MFMA)"}
HBM_PEAK, MFMA_PEAK_TF = 8000.0, 2500.0      # GB/s; dense bf16 TFLOP/s (MI355X_MICROARCH.md)


def unet_bytes(size, n_train, n_nograd, n_loss, ncls, w_mb, trainable_nets=1, nograd_nets=0):
    """train = N(3 IN + 5 OUT) + 7 W per trainable network, no-grad forward = N(IN + OUT) + W, loss = (12 C + 1) B / pixel / image."""
    u = UNET[size]
    loss_mb = (12 * ncls + 1) * size * size / 1e6
    mb = trainable_nets * (n_train * (3 * u["IN"] + 5 * u["OUT"]) + 7 * w_mb) + nograd_nets * (n_nograd * (u["IN"] + u["OUT"]) + w_mb) + n_loss * loss_mb
    return mb * 1e6


def segformer_b0_bytes(n_img, size=224):
    """Byte model of the SegFormer-B0 branch (no SURVEY figure exists): per transformer block and token the forward reads / writes about 26 C
    floats (x, q, kv, attention out, projection + residual, fc1 4C, depthwise-GELU 4C + 4C, fc2 + residual), dims (32, 64, 160, 256) at
    (H/4)^2 ... (H/32)^2 tokens, two blocks per stage; the head adds about 10 MB per image; training = 3 x forward; weights 3.7 M x 4 B x 7."""
    dims, fwd = (32, 64, 160, 256), 0.0
    for i, c in enumerate(dims):
        tokens = (size // (4 << i)) ** 2
        fwd += 2 * tokens * 26 * c * 4
    fwd += 10e6
    return n_img * 3 * fwd + 7 * 3.7e6 * 4


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="mt", choices=["mt", "sup", "hpfg", "cps", "ctct"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32-line", action="store_true", help="skip the secondary exact-fp32 measurement")
    ap.add_argument("--no-probe", action="store_true", help="skip the per-kernel time stamps (roofline = step figures only)")
    ap.add_argument("--math", default=os.environ.get("HPFG_MATH", "bf16x3"), choices=["bf16x3", "f32"])
    ap.add_argument("--lab", type=int, default=None, help="labelled images per GPU (default: the workload's BASELINE figure)")
    ap.add_argument("--unlab", type=int, default=None)
    ap.add_argument("--force-sync", action="store_true", help="diagnostics: run the data-parallel code path (RCCL collectives) on one rank")
    ap.add_argument("--sync-bn", action="store_true", help="(the default for N > 1; kept for older command lines) exchange every BatchNorm statistic and the "
                    "loss sums: R ranks == one process on the global batch")
    ap.add_argument("--local-bn", action="store_true", help="N > 1: per-rank BatchNorm + averaged gradients (DistributedDataParallel semantics) as the headline "
                    "mode; the global-batch mode is then the `other_bn_mode` line")
    ap.add_argument("--overlap", action="store_true", help="(the default for N > 1) bucketed gradient exchange overlapped with the encoder half of backward "
                    "(peer windows: inside the one graph; --rccl: a chain of hipGraphs around eager RCCL calls)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: ONE gradient exchange after backward instead")
    ap.add_argument("--blocks", type=int, default=5, help="timed blocks of --steps steps (block 0 is the contract's measurement)")
    ap.add_argument("--rccl", action="store_true", help="N > 1: the gradient all-reduce as an RCCL call between two hipGraphs instead of the peer-window "
                    "exchange captured inside the step's one graph (the fallback bench.py takes by itself when the peer windows fail their self-test)")
    ap.add_argument("--no-p2p", action="store_true", help="--sync-bn: exchange the BatchNorm / loss sums with host-launched RCCL all-reduces (eager step) instead "
                    "of the peer mailboxes the finalize kernels write themselves (hipIpc over xGMI; the step stays a chain of hipGraphs)")
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


# ---- workloads ----------------------------------------------------------------------------------------------------------------------
class Workload:
    """One BASELINE config: models + step object + resident inputs + the algorithmic byte / flop figures of one step on one rank."""

    def __init__(self, name, a, dev, math, dp, rank):
        import numpy as np
        import torch
        from copy import deepcopy
        from hpfg_amd.datasets.synthetic import synth_batch
        from hpfg_amd.model import build_model
        from hpfg_amd.train import CPSStep, CTCTStep, HPFGStep, MeanTeacherStep, SupervisedStep, batch_pair
        from hpfg_amd.utils import loadyaml
        self.name, self.dev = name, dev
        cfg = {"mt": "mean_teacher_unet_30k_224x224_ACDC.yaml", "sup": "unet_30k_224x224_ACDC.yaml", "hpfg": "hpfg_unet_plus_30k_224x224_ACDC.yaml",
               "cps": "cps_unet_30k_96x96_LIDC.yaml", "ctct": "ctct_unet_segformer_30k_224x224_ACDC.yaml"}[name]
        args = loadyaml(os.path.join(ROOT, "config", cfg))
        args.device = dev
        lab0, unlab0 = {"mt": (8, 8), "sup": (8, 0), "hpfg": (16, 16), "cps": (32, 32), "ctct": (8, 24)}[name]
        lab = lab0 if a.lab is None else a.lab
        unlab = unlab0 if a.unlab is None else a.unlab
        args.batch_size, args.unlabel_batch_size = lab, unlab
        self.lab, self.unlab, self.n_img = lab, unlab, lab + unlab
        torch.manual_seed(args.seed)

        def mk(sub):
            m = build_model(sub).to(dev)
            if hasattr(m, "math"):
                m.math = math
            return m

        def frozen(m):
            e = deepcopy(m)
            for p in e.parameters():
                p.requires_grad = False
            e.train()
            return e

        s1, s2 = 1234 + rank, 91234 + rank
        if name == "sup":
            self.size = 224
            m = mk(args)
            m.train()
            self.models, self.step = [m], SupervisedStep(m, args, dp)
            x, y = synth_batch(s1, lab, 224, 224, 1, 4, 32)
            self.inputs = (x.to(dev), y.to(dev))
            self.bytes = unet_bytes(224, lab, 0, lab, 4, W_UNET)
            self.gflop = lab * 3 * UNET[224]["GF"]
            self.desc = f"supervised_unet_224x224 (BASELINE configs[0]): U-Net 1ch->4cls, {lab} images per GPU, fwd+bwd + 0.5 CE + 0.5 Dice + SGD"
        elif name == "mt":
            self.size = 224
            m = mk(args)
            ema = frozen(m)
            m.train()
            self.models, self.step = [m, ema], MeanTeacherStep(m, ema, args, dp)
            xl, yl = synth_batch(s1, lab, 224, 224, 1, 4, 32)
            xu, _ = synth_batch(s2, unlab, 224, 224, 1, 4, 32)
            xl, xu = batch_pair(xl.to(dev), xu.to(dev))      # back to back in HBM: the step's batch is a view, not a concat copy
            self.inputs = (xl, yl.to(dev), xu)
            n = self.n_img
            self.bytes = unet_bytes(224, n, n, n, 4, W_UNET, 1, 1)
            self.gflop = n * (3 * UNET[224]["GF"] + UNET[224]["GF"])
            self.desc = (f"mean_teacher_unet_224x224 (BASELINE configs[1]): U-Net 1ch->4cls, {lab} labelled + {unlab} unlabelled per GPU, "
                         "student fwd+bwd + train-mode teacher fwd + CE/Dice/MSE + SGD + EMA")
        elif name == "hpfg":
            self.size = 224
            m1, m2 = mk(args.model1), mk(args.model2)
            ema = frozen(m2)
            m1.train(), m2.train()
            st = HPFGStep(m1, m2, ema, args, dp)
            self.models, self.step = [m1, m2, ema], st
            xl, yl = synth_batch(s1, lab, 224, 224, 1, 4, 32)
            xl1, yl1 = synth_batch(s1 + 7, lab, 224, 224, 1, 4, 32)
            xu, _ = synth_batch(s2, unlab, 224, 224, 1, 4, 32)
            rep = max(1, unlab // lab)
            cm = st.make_cutmix_mask(unlab, (224, 224), rng=np.random.RandomState(1 + rank))
            self.inputs = tuple(t.to(dev) for t in (xl, yl, xl1.repeat(rep, 1, 1, 1), yl1.repeat(rep, 1, 1), xu, cm))
            n = self.n_img
            self.bytes = unet_bytes(224, n, n, 2 * n, 4, W_PLUS, 2, 1)
            self.gflop = n * (2 * 3 * UNET[224]["GF"] + UNET[224]["GF"]) + 1.0
            self.desc = (f"hpfg_unet_plus_224x224 (BASELINE configs[2]): two U-Net+ students + EMA teacher, {lab} + {unlab} per GPU (every network sees "
                         f"{n} images), CutMix pseudo-labels, Dense_Loss necks, SGD x 2, backbone EMA + teacher EMA")
        elif name == "cps":
            self.size = 96
            m1, m2 = mk(args.model1), mk(args.model2)
            m1.train(), m2.train()
            self.models, self.step = [m1, m2], CPSStep(m1, m2, args, dp)
            xl, yl = synth_batch(s1, lab, 96, 96, 3, 2, 12)
            xu, _ = synth_batch(s2, unlab, 96, 96, 3, 2, 12)
            xl, xu = batch_pair(xl.to(dev), xu.to(dev))
            self.inputs = (xl, yl.to(dev), xu)
            n = self.n_img
            self.bytes = unet_bytes(96, n, 0, 2 * n, 2, W_UNET, 2, 0)
            self.gflop = 2 * n * 3 * UNET[96]["GF"]
            self.desc = f"cps_unet_96x96 (BASELINE configs[3]): twin U-Net 3ch->2cls cross pseudo supervision, {lab} + {unlab} per GPU, SGD x 2"
        else:
            self.size = 224
            m1, m2 = mk(args.model1), mk(args.model2)
            m1.train(), m2.train()
            self.models, self.step = [m1, m2], CTCTStep(m1, m2, args, dp)
            xl, yl = synth_batch(s1, lab, 224, 224, 1, 4, 32)
            xu, _ = synth_batch(s2, unlab, 224, 224, 1, 4, 32)
            xl, xu = batch_pair(xl.to(dev), xu.to(dev))
            self.inputs = (xl, yl.to(dev), xu)
            n = self.n_img
            self.bytes = unet_bytes(224, n, 0, 2 * n, 4, W_UNET) + segformer_b0_bytes(n)
            self.gflop = n * 3 * UNET[224]["GF"] + n * 3 * 1.6      # (SegFormer-B0 forward ~1.6 GFLOP at 224^2: approximate)
            self.desc = (f"ctct_unet_segformer_224x224 (BASELINE configs[4]): U-Net + SegFormer-B0 cross teaching, {lab} + {unlab} per GPU, both networks "
                         "fwd+bwd on all images, SGD + AdamW")

    def unet_engines(self):
        return [e for m in self.models for pool in getattr(m, "_engines", {}).values() for e in pool]


def timed_run(wl, dp, use_graph, steps, warmup, dev, blocks=1):
    """W untimed + K timed steps bracketed by barrier + synchronize; returns (seconds, graph actually used, last iteration, [seconds of every
    block]): block 0 is the contract's measurement, `blocks - 1` further blocks of K steps are timed the same way (spread of the figure)."""
    import torch
    from hpfg_amd.train import GraphedStep
    step, inputs = wl.step, wl.inputs
    runner, it = None, 0
    if use_graph:
        try:
            runner = GraphedStep(step, list(inputs), warmup=3, alias_inputs=True)   # the batch is resident in HBM at fixed addresses
            it = 3
        except Exception as e:  # capture unsupported: fall back to eager launches (still the HIP path)
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            runner, use_graph = None, False
        if dp is not None:
            dp.barrier()      # every rank has finished capturing before any rank replays (the in-graph exchange polls its peers with a bound)

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
    dts = []
    for _ in range(max(1, blocks)):
        t0 = time.perf_counter()
        for _ in range(steps):
            it += 1
            one(it)
        barrier()
        dt = time.perf_counter() - t0
        if dp is not None:
            dt = dp.max_float(dt)
        dts.append(dt)
    if dp is not None:
        # a peer exchange whose poll expired carried on with partial sums (it must not hang the GPU): such a run is not a measurement
        try:
            dp.check_peer_errors()
        except RuntimeError as e:
            print(f"[bench] rank {dp.rank}: {e}", file=sys.stderr, flush=True)
            raise SystemExit(3)
    return dts[0], use_graph, it, dts


# ---- per-kernel probe ------------------------------------------------------------------------------------------------------------------
def layer_costs(eng, tag):
    """(family, algorithmic bytes, flops) of one bracketed launch `tag` = "<pass>:<conv name>" of engine `eng`.
    Bytes: the tensors the pass must move given what is stored: fwd = stored input + raw output + weights (a pooled
    input is the full-resolution producer, a concat input the skip tensor + the quarter-resolution 1x1 output); dgrad = dA + z + dX;
    wgrad = dA + z + stored input; fused backward = dA + z + stored input + dX.  Layers with side tensors (HpfgConvArgs.stage_out): the
    forward conv also writes its staged input, the dgrad its dZ, and the weight gradient reads those two tensors instead.  FLOPs: 2 x taps x Cin x Cout per output pixel and pass."""
    kind, name = tag.split(":", 1) if ":" in tag else (tag, "")
    s = eng.specs.get(name)
    if s is None or kind not in ("fwd", "dgrad", "wgrad", "fused_bwd"):
        return {"bn_fin": "BatchNorm forward finalize", "bn_red": "BatchNorm backward reduction (+ max-pool backward)", "bn_bfin": "BatchNorm backward finalize",
                "upbwd": "bilinear upsample backward", "slab_reduce": "weight-gradient slab reduction", "pack_weights": "weight packing",
                "csum": "bias-gradient channel sums"}.get(kind, kind), 0.0, 0.0, 0.0
    N = eng.N
    px = N * s.h * s.w
    if name.startswith("decoder.up") and name.endswith("conv.conv_conv.0"):
        x_b = px * (s.cin // 2) * 4 + px // 4 * (s.cin // 2) * 4           # skip + low-resolution 1x1 output
    elif name.startswith("encoder.down") and name.endswith(".0"):
        x_b = 4 * px * s.cin * 4                                           # the full-resolution producer behind MaxPool2d(2)
    else:
        x_b = px * s.cin * 4
    o_b, w_b = px * s.cout * 4, s.taps * s.cin * s.cout * 4
    fl = 2.0 * px * s.taps * s.cin * s.cout
    if s.idx == 0:
        cls = "first conv (1 / 3 input channels)"
    elif s.taps == 1:
        cls = "1x1 conv"
    elif min(s.cin, s.cout) >= 32:
        cls = "channel-rich 3x3 (>= 32 channels in and out)"
    else:
        cls = "thin 3x3 (< 32 channels in or out: the 224 x 224 / 112 x 112 layers and out_conv)"
    # The ALGORITHMIC bytes are SURVEY.md section 8(d)'s model and nothing else: a conv reads its logical input once and writes its raw output
    # once; backward per conv reads (dA, z) [+ the input for the weight gradient] and writes dX.  What this build moves BESIDE that -- the side
    # tensors of HpfgConvArgs.stage_out (a forward conv / dgrad also stores what it staged), the max-pool backward riding in a dgrad epilogue,
    # the weight-gradient slabs written for the fixed-order reduction -- is returned separately as overhead bytes.
    act_st = px * s.cin * 4 if name in getattr(eng, "_act_live", ()) else 0
    dz_st = o_b if name in getattr(eng, "dzbuf", {}) else 0
    slab = 0
    if hasattr(eng, "slab_of") and name in getattr(eng, "slab_of", {}):
        S = eng.fused_grid.get(name) or eng.lib.hpfg_wgrad_splits(N, s.h, s.w, s.cin_pad, s.cout_pad, s.taps)
        slab = S * s.taps * s.cin_pad * s.cout_pad * 4
    if kind == "fwd":
        return f"forward conv, {cls}", x_b + o_b + w_b, fl, act_st
    if kind == "dgrad":
        from hpfg_amd.engine import enc_prefix
        pool_of = next((enc_prefix(lv - 1) + ".4" for lv in range(1, 5) if name == enc_prefix(lv) + ".0"), None)
        # max-pool backward in the epilogue: z and the gradient so far (read + written) at 2H x 2W instead of writing dP
        pool_x = (3 * 4 - 1) * px * s.cin * 4 if pool_of in getattr(eng, "_pool_done", ()) else 0
        return f"input gradient (separate dgrad), {cls}", 2 * o_b + px * s.cin * 4 + w_b, fl, dz_st + pool_x
    if kind == "wgrad":
        return f"weight gradient (separate wgrad), {cls}", 2 * o_b + x_b, fl, slab
    dx = 0 if s.idx == 0 else px * s.cin * 4
    return f"fused dgrad + wgrad, {cls}", 2 * o_b + x_b + dx + w_b, fl * (1 if s.idx == 0 else 2), slab


def probe_families(wl, it, reps=6):
    """Times of every bracketed launch of the step inside a captured hipGraph (see module docstring); returns (families, longest, calib_us)."""
    import numpy as np
    import torch
    from hpfg_amd.engine import MarkLog
    from hpfg_amd.train import GraphedStep
    engines = wl.unet_engines()
    if not engines:
        return None
    logs = {id(e): MarkLog(wl.dev) for e in engines}
    for e in engines:
        e.marks = logs[id(e)]

    def reset():
        for lg in logs.values():
            lg.n, lg.spans = 0, []

    try:
        runner = GraphedStep(wl.step, list(wl.inputs), warmup=1, alias_inputs=True, before_capture=reset)
        acc = {}
        for r in range(reps):
            runner.step(list(wl.inputs), it + 1 + r)
            torch.cuda.synchronize(wl.dev)
            if r < 2:
                continue      # warm-up replays
            for e in engines:
                for i, (tag, us) in enumerate(logs[id(e)].read_us()):
                    acc.setdefault((id(e), i, tag), []).append(us)
    finally:
        for e in engines:
            e.marks = None
    calib = float(np.median([np.mean(v) for (_, _, tag), v in acc.items() if tag == "calib"] or [0.0]))
    eng_of = {id(e): e for e in engines}
    fams, longest = {}, None
    for (eid, _, tag), v in acc.items():
        if tag == "calib":
            continue
        us = max(float(np.mean(v)) - calib, 0.0)
        fam, b, fl, ov = layer_costs(eng_of[eid], tag)
        f = fams.setdefault(fam, dict(us=0.0, launches=0, bytes=0.0, flops=0.0, overhead=0.0))
        f["us"] += us
        f["launches"] += 1
        f["bytes"] += b
        f["flops"] += fl
        f["overhead"] += ov
        if b > 0 and (longest is None or us > longest["us"]):
            longest = dict(tag=tag, us=us, bytes=b, flops=fl, family=fam)
    return fams, longest, calib


def probe_solo_forward(wl, family, calib, reps=4):
    """The dominant forward family WITHOUT a second network beside it: train-mode forwards of the step's first network alone on the stream (the
    teacher's / second student's launches of the same layers run side by side with these in the step, and each then sees half the chip), bracketed
    by the same device time stamps.  Returns (us per forward, launches, flops, bytes) of `family`, or None where the workload has no such network."""
    import numpy as np
    import torch
    from hpfg_amd.engine import MarkLog
    from hpfg_amd.train import cat_batch
    if wl.name not in ("mt", "sup") or not family.startswith("forward conv"):
        return None
    m = wl.models[0]
    x = wl.inputs[0] if wl.name == "sup" else cat_batch(wl.inputs[0], wl.inputs[2])
    engs = [e for pool in m._engines.values() for e in pool]
    logs = {id(e): MarkLog(wl.dev) for e in engs}
    acc = {}
    try:
        for r in range(reps):
            for e in engs:
                e.marks = logs[id(e)] if r >= 1 else None
                logs[id(e)].n, logs[id(e)].spans = 0, []
            with torch.no_grad():
                m(x)
            torch.cuda.synchronize(wl.dev)
            if r >= 1:
                for e in engs:
                    for i, (tag, us) in enumerate(logs[id(e)].read_us()):
                        acc.setdefault((id(e), i, tag), []).append(us)
    finally:
        for e in engs:
            e.marks = None
    eng_of = {id(e): e for e in engs}
    us = fl = by = 0.0
    n = 0
    for (eid, _, tag), v in acc.items():
        fam, b, f, _ = layer_costs(eng_of[eid], tag)
        if fam == family:
            us += max(float(np.mean(v)) - calib, 0.0)
            fl += f
            by += b
            n += 1
    return (us, n, fl, by) if n else None


def roofline_objects(wl, fams, longest, calib, math):
    passes = 3.0 if math == "bf16x3" else 16.0      # bf16 MFMA-equivalents per product (exact-fp32 MFMA runs at 1/16 of the bf16 rate)
    total = sum(f["us"] for f in fams.values())
    conv = {k: f for k, f in fams.items() if f["bytes"] > 0}
    name, top = max(conv.items(), key=lambda kv: kv[1]["us"])

    def obj(label, us, b, fl, n=1):
        gbps, tf = b / (us * 1e-6) / 1e9, fl * passes / (us * 1e-6) / 1e12
        hbm_frac, mfma_frac = gbps / HBM_PEAK, tf / MFMA_PEAK_TF
        bound = "hbm" if hbm_frac >= mfma_frac else "mfma"
        return {"kernel": label, "bound": bound, "achieved": round(gbps if bound == "hbm" else tf, 1), "peak": HBM_PEAK if bound == "hbm" else MFMA_PEAK_TF,
                "unit": "GB/s" if bound == "hbm" else "TFLOP/s", "frac": round(max(hbm_frac, mfma_frac), 4), "traffic": None,
                "avg_launch_us": round(us / n, 2), "launches_per_step": n, "algorithmic_bytes_per_launch": int(b / n), "flops_per_launch": int(fl / n),
                "hbm_frac": round(hbm_frac, 4), "mfma_frac": round(mfma_frac, 4)}
    roof = obj(f"{name}: all {top['launches']} launches of the family in one step (every network of the step)", top["us"], top["bytes"], top["flops"],
               top["launches"])
    roof["share_of_bracketed_kernel_time"] = round(top["us"] / total, 4)
    roof["timing"] = (f"device time stamps (s_memrealtime) around each launch inside a captured hipGraph of the step, minus the same bracket around nothing "
                      f"({calib:.2f} us); bf16-MFMA passes per product: {passes:g}")
    # HBM bytes per launch of this family: NOT measured by this run -- the figure of the PMC passes (FETCH_SIZE x 2 / WRITE_SIZE, separate
    # rocprofv3 --pmc runs of the same step) committed under profiles/ by tools/family_traffic.py; `traffic_source` names the file
    for tf in ("r05_family_traffic.json", "r04_family_traffic.json", "r03_family_traffic.json"):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", tf)))
            if tj.get("workload") == wl.name and name in tj.get("families", {}):
                roof["traffic"] = int(tj["families"][name]["hbm_bytes_per_launch"])
                roof["traffic_source"] = f"profiles/{tf} (committed PMC passes of `{tj.get('command', 'bench.py --no-graph')}`, {tj.get('commit', 'commit n/a')}); not collected by this run"
                break
        except Exception:
            pass
    roof["overhead_bytes_per_launch"] = int(top["overhead"] / top["launches"])
    roof["bytes_model"] = ("algorithmic_bytes_per_launch = SURVEY.md section 8(d) only (logical input + raw output + weights; backward: dA, z [, input] + dX); "
                           "overhead_bytes_per_launch = what this build moves beside it (side tensors, max-pool backward in a dgrad epilogue, "
                           "weight-gradient slabs)")
    roof["families"] = {k: {"us_per_step": round(f["us"], 1), "launches": f["launches"], "share": round(f["us"] / total, 4),
                            **({"overhead_MB_per_step": round(f["overhead"] / 1e6, 1)} if f.get("overhead") else {}),
                            **({"GBps": round(f["bytes"] / (f["us"] * 1e-6) / 1e9, 1), "TFLOPs_bf16_passes": round(f["flops"] * passes / (f["us"] * 1e-6) / 1e12, 1)}
                               if f["bytes"] > 0 and f["us"] > 0 else {})}
                        for k, f in sorted(fams.items(), key=lambda kv: -kv[1]["us"])}
    if longest is not None:
        roof["longest_launch"] = obj(f"{longest['tag']} ({longest['family']})", longest["us"], longest["bytes"], longest["flops"])
    try:
        solo = probe_solo_forward(wl, name, calib)
    except Exception as e:
        print(f"[bench] solo probe failed ({type(e).__name__}: {e})", file=sys.stderr)
        solo = None
    if solo is not None:      # the same family with nothing beside it: what the kernels do with the whole chip (NOT the step's schedule)
        us, n, fl, by = solo
        roof["alone_on_the_chip"] = {"launches": n, "avg_launch_us": round(us / n, 2), "mfma_frac": round(fl * passes / (us * 1e-6) / 2.5e15, 4),
                                     "hbm_frac": round(by / (us * 1e-6) / 8e12, 4),
                                     "note": "train-mode forwards of the step's first network alone on one stream, same brackets; `frac` above is measured "
                                             "in the step, where the other network's launches of the same layers share the chip"}
    return roof


# ---- CPU baseline ----------------------------------------------------------------------------------------------------------------------
def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores():
    """(physical cores, logical CPUs) among the CPUs this process may run on (SMT siblings counted once)."""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        return os.cpu_count() or 1, os.cpu_count() or 1
    cores, cur = set(), {}

    def flush():
        if cur and int(cur.get("processor", -1)) in allowed:
            cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))

    try:
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [t.strip() for t in line.split(":", 1)]
                cur[k] = v
            else:
                flush()
                cur = {}
        flush()
    except OSError:
        pass
    return (len(cores) or len(allowed)), len(allowed)


def cpu_baseline(name, lab, unlab, size):
    """The CPU oracle (plain PyTorch restatement of the reference step, oracle/steps_ref.py) of the same workload on this host's cores,
    SURVEY.md section 8(d): k = all physical cores, 1-3 warm-up and up to 10 timed steps within ~30 s, median."""
    import numpy as np
    import torch
    from hpfg_amd.datasets.synthetic import synth_batch
    from oracle import laws_ref, steps_ref, unet_ref
    cores, avail = physical_cores()
    torch.set_num_threads(cores)
    lr = laws_ref.medical_lr(1, 0.01, 30000)
    if name == "sup":
        st, bufs = unet_ref.init_state(1, 1, 4), {}
        x, y = synth_batch(1234, lab, size, size, 1, 4, 32)
        fn = lambda k: steps_ref.supervised_step(st, bufs, x, y.long(), lr, 0.9, 5e-4)
    elif name == "mt":
        st = unet_ref.init_state(1337, 1, 4)
        ema, bufs = unet_ref.clone_state(st), {}
        xl, yl = synth_batch(1234, lab, size, size, 1, 4, 32)
        xu, _ = synth_batch(91234, unlab, size, size, 1, 4, 32)
        fn = lambda k: steps_ref.mean_teacher_step(st, ema, bufs, xl, yl.long(), xu, laws_ref.medical_lr(k, 0.01, 30000), 0.0, laws_ref.ema_alpha(k, 0.99))
    elif name == "cps":
        torch.manual_seed(1337)
        sa, sb, ba, bb = unet_ref.init_state(None, 3, 2), unet_ref.init_state(None, 3, 2), {}, {}
        xl, yl = synth_batch(1234, lab, size, size, 3, 2, 12)
        xu, _ = synth_batch(91234, unlab, size, size, 3, 2, 12)
        fn = lambda k: steps_ref.cps_step(sa, sb, ba, bb, xl, yl.long(), xu, lr, lr, 0.004)
    elif name == "hpfg":
        torch.manual_seed(1)
        sa, sb = unet_ref.init_state(None, 1, 4, True), unet_ref.init_state(None, 1, 4, True)
        se, ba, bb = unet_ref.clone_state(sb), {}, {}
        xl, yl = synth_batch(1234, lab, size, size, 1, 4, 32)
        xl1, yl1 = synth_batch(1241, lab, size, size, 1, 4, 32)
        xu, _ = synth_batch(91234, unlab, size, size, 1, 4, 32)
        rep = max(1, unlab // lab)
        cm = torch.tensor(laws_ref.box_masks(unlab, (size, size), np.random.RandomState(1)), dtype=torch.float)
        fn = lambda k: steps_ref.hpfg_step(sa, sb, se, ba, bb, xl, yl.long(), xl1.repeat(rep, 1, 1, 1), yl1.long().repeat(rep, 1, 1), xu, cm, 1000 + k, lr, lr,
                                           0.1, 200.0, 0.99)
    else:
        from oracle import segformer_ref
        s1, s2, b1, ad = unet_ref.init_state(1, 1, 4), segformer_ref.init_state(2, 1, 4), {}, {}
        xl, yl = synth_batch(1234, lab, size, size, 1, 4, 32)
        xu, _ = synth_batch(91234, unlab, size, size, 1, 4, 32)
        fn = lambda k: steps_ref.ctct_step(s1, s2, b1, ad, xl, yl.long(), xu, 0.01, 8e-4, 0.004)
    def sample(threads, budget):
        torch.set_num_threads(threads)
        times, t_start, warm = [], time.perf_counter(), 0
        for k in range(1, 14):
            t0 = time.perf_counter()
            fn(k)
            dt = time.perf_counter() - t0
            if k == 1 or (warm < 3 and time.perf_counter() - t_start < budget / 4):
                warm += 1          # (a 20-second step -- HPFG -- gets one warm-up, a 1-second step three)
            else:
                times.append(dt)
            if times and time.perf_counter() - t_start > budget:
                break
        ts = sorted(times or [dt])
        return ts[len(ts) // 2], warm, len(ts)

    # SURVEY.md 8(d) asks for k = all physical cores; torch's CPU kernels on a 224 x 224 U-Net stop scaling long before 128 threads (the
    # synchronisation of each small op dominates), so the same sample is also timed on 16 threads (one GPU's share of this host) and the
    # FASTER of the two is the reported baseline -- both figures are in `sample`.
    n = lab + unlab
    runs = [(cores,) + sample(cores, 15.0)]
    if cores > 16:
        runs.append((16,) + sample(16, 15.0))
    best = min(runs, key=lambda r: r[1])
    return {"value": round(n / best[1], 2), "unit": "images/s", "cores": best[0], "kind": "port", "cpu_model": _cpu_model(), "os_cpu_count": os.cpu_count(),
            "affinity_cpus": avail, "physical_cores": cores,
            "sample": f"`{name}` steps of {lab}+{unlab} images at {size}x{size}, CPU oracle, torch CPU fp32, median step time: " +
                      "; ".join(f"{r[0]} threads{' (all physical cores)' if r[0] == cores else ''}: {n / r[1]:.2f} images/s ({r[2]} warm-up + {r[3]} timed)" for r in runs)}


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

    dp = None
    if world > 1 or a.force_sync:
        if a.force_sync and world == 1:      # documented single-process diagnostic: supply the rendezvous a launcher would
            for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29511"), ("RANK", "0"), ("WORLD_SIZE", "1")):
                os.environ.setdefault(k, v)
        dp = parallel.init_from_env(dev, backend=os.environ.get("HPFG_DP_BACKEND") or None)
        dp.force_sync = bool(a.force_sync)
        # N > 1 default = the mode north_star names: BatchNorm / loss sums exchanged (global-batch equivalence), gradient buckets overlapped with backward
        dp.sync_bn = bool((world > 1 and not a.local_bn) or a.sync_bn or (a.force_sync and not a.local_bn))
        dp.overlap = bool(world > 1 or a.overlap) and not a.no_overlap
        if dp.sync_bn and not a.no_p2p:
            dp.enable_peer_exchange()
    wl = Workload(a.workload, a, dev, a.math, dp, rank)
    exchange = None
    if dp is not None and world > 1:
        # N > 1 default: the flat gradients cross the ranks through IPC-mapped peer windows (xGMI pushes: csrc/peer.hip) as kernels of the
        # captured step -- one hipGraph per step, as at N = 1.  The path is tried on the actual devices first; RCCL between two graphs otherwise.
        exchange = "RCCL all-reduce between two hipGraphs"
        if not a.rccl:
            nmax = max([int(m.flat_grads.numel()) if hasattr(m, "flat_grads") and not hasattr(m, "_hpfg_generic_flat") else
                        sum(p.numel() for p in m.parameters() if p.requires_grad) for m in wl.models if any(p.requires_grad for p in m.parameters())] or [0])
            if nmax and dp.enable_peer_grads(nmax):
                exchange = "peer-window all-reduce over xGMI (push / reduce / gather kernels inside the step's hipGraph" + \
                    (", decoder bucket on a side stream beside the encoder half of backward)" if dp.overlap else ")")
            elif rank == 0:
                print("[bench] peer-window gradient exchange unavailable on this node; using RCCL", file=sys.stderr)
        if dp.overlap and not dp.p2p_grads:
            exchange = "bucketed RCCL all-reduces overlapped with backward (chain of hipGraphs)"

    # N > 1: graphs around the gradient all-reduce (no RCCL node inside a hipGraph).  --sync-bn exchanges the BatchNorm / loss sums inside the
    # kernels (peer mailboxes) and captures as well; with --no-p2p (collectives between the kernels), and for the HPFG step (its Dense_Loss
    # all-gathers the neck features with a host-launched collective), the global-batch mode runs eager.
    sync_mode = dp is not None and dp.sync_bn and (world > 1 or a.force_sync)
    use_graph = (not a.no_graph) and (not sync_mode or (dp.p2p and (a.workload != "hpfg" or dp.p2p_grads)))
    dt, use_graph, it, dts = timed_run(wl, dp, use_graph, a.steps, a.warmup, dev, blocks=a.blocks)
    ms = dt / a.steps * 1e3
    value = wl.n_img * world / (dt / a.steps)
    bl = sorted(d / a.steps * 1e3 for d in dts)
    blocks = {"ms_per_step": [round(d / a.steps * 1e3, 4) for d in dts], "median_ms": round(bl[len(bl) // 2], 4), "min_ms": round(bl[0], 4),
              "max_ms": round(bl[-1], 4), "note": "block 0 = the contract's K timed steps (value / ms_per_step); each block bracketed by barrier + synchronize"}
    headline_sync = bool(dp is not None and dp.sync_bn)

    other = None
    if dp is not None and world > 1:
        # the other BatchNorm mode on the same ranks (fresh models and step objects): DistributedDataParallel semantics when the headline is
        # the global-batch mode, and vice versa
        dp.sync_bn = not headline_sync
        if dp.sync_bn and not a.no_p2p:
            dp.enable_peer_exchange()
        try:
            wl2 = Workload(a.workload, a, dev, a.math, dp, rank)
            sync2 = dp.sync_bn
            ug2 = (not a.no_graph) and (not sync2 or (dp.p2p and (a.workload != "hpfg" or dp.p2p_grads)))
            k2 = max(5, a.steps // 2)
            dt2, ug2, _, _ = timed_run(wl2, dp, ug2, k2, max(3, a.warmup // 2), dev)
            other = {"sync_bn": bool(sync2), "value": round(wl2.n_img * world / (dt2 / k2), 2), "unit": "images/s", "ms_per_step": round(dt2 / k2 * 1e3, 4),
                     "steps": k2, "hipgraph": bool(ug2),
                     "mode": "global-batch equivalence (BatchNorm + loss sums exchanged)" if sync2 else "per-rank BatchNorm, gradients averaged (DDP semantics)"}
            del wl2
        finally:
            dp.sync_bn = headline_sync

    xpath = None
    if dp is not None:          # every rank's exchange path + peer error word (collective: all ranks call it), for the line rank 0 prints
        xpath = dp.exchange_report(rccl_requested=a.rccl)
    roof = f32 = cpu = None
    if rank == 0 and not a.no_probe and use_graph:
        pw = wl
        if dp is not None:      # the per-rank work is the same at every N (weak scaling): probe a private copy of the step without a process group
            pw = Workload(a.workload, a, dev, a.math, None, rank)
        try:
            got = probe_families(pw, it)
            if got is not None:
                roof = roofline_objects(pw, *got, a.math)
        except Exception as e:
            print(f"[bench] per-kernel probe failed ({type(e).__name__}: {e})", file=sys.stderr)
        if pw is not wl:
            del pw
    if world == 1 and dp is None and not a.no_f32_line and a.math != "f32" and a.workload == "mt":
        del wl
        torch.cuda.empty_cache()
        wl = Workload(a.workload, a, dev, "f32", None, rank)
        k2 = max(5, a.steps // 2)
        dt2, g2, _, _ = timed_run(wl, None, use_graph, k2, max(3, a.warmup // 2), dev)
        f32 = {"dtype": DTYPE["f32"], "value": round(wl.n_img / (dt2 / k2), 2), "unit": "images/s", "ms_per_step": round(dt2 / k2 * 1e3, 4), "steps": k2,
               "hipgraph": bool(g2), "note": "same step, HPFG_MATH=f32: every product exact fp32 (v_mfma_f32_16x16x4_f32); meets 1e-3 on every fixture"}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:      # reported baseline: rank 0 at N=1 only
        cpu = cpu_baseline(a.workload, wl.lab, wl.unlab, wl.size)
    if rank == 0:
        par = f"dp{world}"
        if world > 1:
            par += (" (global-batch mode: BatchNorm + loss sums exchanged " + ("by the finalize kernels through peer mailboxes over xGMI" if dp.p2p else
                    "by RCCL all-reduces between the kernels") + "; == one process on the global batch)") if (dp is not None and dp.sync_bn) else \
                " (per-rank BatchNorm, gradients averaged)"
            par += f"; gradient exchange: {exchange}"
        step_roof = {"algorithmic_GB_per_step": round(wl.bytes / 1e9, 3), "achieved_GBps": round(wl.bytes / (dt / a.steps) / 1e9, 1),
                     "frac_of_8TBps": round(wl.bytes / (dt / a.steps) / 8e12, 4), "algorithmic_GFLOP_per_step": round(wl.gflop, 1),
                     "achieved_TFLOPs": round(wl.gflop / (dt / a.steps) / 1e3, 2)}
        if roof is None:      # probe skipped / unavailable: the step as a whole is the only measured aggregate
            roof = {"kernel": "whole step (per-kernel probe not run)", "bound": "hbm", "achieved": step_roof["achieved_GBps"], "peak": HBM_PEAK, "unit": "GB/s",
                    "frac": step_roof["frac_of_8TBps"], "traffic": None}
        metric = "labeled+unlabeled images/sec/node, U-Net 224x224 ACDC-shaped (Mean-Teacher step)" if a.workload == "mt" else \
            f"labeled+unlabeled images/sec/node ({a.workload} step)"
        out = {
            "metric": metric, "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE[a.math], "data": "synthetic",
            "config": {"workload": wl.desc, "per_gpu_batch": [wl.lab, wl.unlab], "size": wl.size, "hipgraph": bool(use_graph),
                       "sync_bn": bool(dp is not None and dp.sync_bn), "parallelism": par, "math": a.math},
            "step_roofline": step_roof, "roofline": roof, "blocks": blocks, "other_bn_mode": other, "f32_math": f32, "cpu_baseline": cpu,
            "exchange_path": xpath,      # N > 1: per rank {grad_path: peer-window | rccl-fallback (why), bn_loss_path, windows_mapped, self_test, peer_err}
        }
        print(json.dumps(out), flush=True)
    if dp is not None:
        dp.shutdown()


if __name__ == "__main__":
    main()
