"""One data-parallel rank of tests/test_gpu_dp_two_ranks.py (started as a child process: RANK / WORLD_SIZE / MASTER_* in the environment).
Runs Mean-Teacher steps of the REAL engine on this rank's shard of a fixed global batch and saves losses + final parameters."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from hpfg_amd import engine as E  # noqa: E402
from hpfg_amd import parallel  # noqa: E402
from hpfg_amd.datasets.synthetic import synth_batch  # noqa: E402
from hpfg_amd.model import UNet, reset_dropout_streams  # noqa: E402
from hpfg_amd.train import MeanTeacherStep  # noqa: E402
from hpfg_amd.utils import AttrDict  # noqa: E402

N_LAB = N_UNL = 4
SIZE = 64


def opt_args():
    return AttrDict(opt="sgd", lr=0.01, momentum=0.9, weight_decay=1e-4, sched="medical", total_itrs=30000, step_size=200, warmup_epochs=0,
                    warmup_lr=1e-4, min_lr=1e-6, consistency=0.1, consistency_rampup=200.0, ema_decay=0.99)


def global_batch():
    xl, yl = synth_batch(11, N_LAB, SIZE, SIZE, 1, 4, 8)
    xu, _ = synth_batch(12, N_UNL, SIZE, SIZE, 1, 4, 8)
    return xl, yl, xu


def image_masks(step, who, n_images):
    """Per-IMAGE dropout keep masks of the five encoder dropout sites (explicit masks: the device RNG is indexed by the position inside
    the local batch, so shards could not reproduce a global batch's draws).  Returns {conv name: uint8 [n_images, h, w, C]}."""
    g = torch.Generator().manual_seed(1000 * step + who)
    out = {}
    for lvl in range(5):
        c, h = E.WIDTHS[lvl], SIZE >> lvl
        out[E.enc_prefix(lvl) + ".0"] = (torch.rand(n_images, h, h, c, generator=g) >= E.ENC_DROPOUT[lvl]).to(torch.uint8)
    return out


def take(masks, idx, dev):
    return {k: v[idx].contiguous().to(dev) for k, v in masks.items()}


def run(dev, dp, rank, world, steps=2, overlap=True):
    torch.manual_seed(5)
    reset_dropout_streams()
    m = UNet(1, 4).to(dev)
    m.math = "f32"
    from copy import deepcopy
    ema = deepcopy(m)
    for p in ema.parameters():
        p.requires_grad = False
    m.train()
    ema.train()
    if dp is not None:
        dp.sync_bn, dp.overlap = True, overlap
    st = MeanTeacherStep(m, ema, opt_args(), dp)
    xl, yl, xu = global_batch()
    kl, ku = N_LAB // world, N_UNL // world
    il = list(range(rank * kl, (rank + 1) * kl))
    iu = list(range(rank * ku, (rank + 1) * ku))
    idx = il + [N_LAB + i for i in iu]          # this rank's images in the order of ITS batch [labelled shard ; unlabelled shard]
    losses = []
    for k in range(1, steps + 1):
        m.external_dropout_masks = take(image_masks(k, 0, N_LAB + N_UNL), idx, dev)
        ema.external_dropout_masks = take(image_masks(k, 1, N_LAB + N_UNL), idx, dev)
        r = st.step(xl[il].to(dev), yl[il].to(dev), xu[iu].to(dev), k, cons_w=0.05)
        losses.append(r["parts"].cpu())
    torch.cuda.synchronize()
    return torch.stack(losses), m.flat_params.cpu(), ema.flat_params.cpu(), dict(ema.named_buffers())["encoder.down2.maxpool_conv.1.conv_conv.1.running_var"].cpu()


def main():
    out = sys.argv[1]
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dp = parallel.init_from_env(dev, backend="gloo")
    try:
        res = run(dev, dp, dp.rank, dp.world_size, overlap=os.environ.get("HPFG_TEST_OVERLAP", "1") == "1")
        torch.save(res, f"{out}.rank{dp.rank}")
    finally:
        dp.shutdown()


if __name__ == "__main__":
    main()
