"""GPU: device-side slice augmentation (hpfg_amd.datasets.device_pool) is bit-identical to the host pipeline of the reference
(oracle/augment_ref.py = datasets/utils.py RandomGenerator) when both replay the same random draws."""
import random

import numpy as np
import pytest
import torch

from hpfg_amd.datasets.device_pool import DeviceSlicePool, RandomGeneratorDevice
from oracle import augment_ref

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _slices(seed, n):
    g = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        h, w = int(g.integers(150, 260)), int(g.integers(150, 260))
        lab = np.kron(g.integers(0, 4, (8, 8)), np.ones((h // 8 + 1, w // 8 + 1), dtype=np.int64))[:h, :w].astype(np.uint8)
        img = (lab / 3.0 + 0.1 * g.standard_normal((h, w))).astype(np.float32)
        out.append((img, lab))
    return out


@pytest.mark.parametrize("size", [(224, 224), (96, 128)])
def test_device_augmentation_equals_host_pipeline(size):
    slices = _slices(7, 12)
    pool = DeviceSlicePool(slices, DEV)
    gen = RandomGeneratorDevice(size)
    idx = [3, 0, 11, 5, 5, 7, 1, 9, 2, 10, 4, 6, 8, 3, 0, 11, 7, 7, 2, 1, 6, 9, 10, 4]
    img_d, lab_d = gen(pool, idx, py_rng=random.Random(123), np_rng=np.random.RandomState(456))
    py, nr = random.Random(123), np.random.RandomState(456)
    modes = set()
    for b, i in enumerate(idx):
        st = (py.getstate(), nr.get_state())
        ri, rl = augment_ref.random_generator(slices[i][0], slices[i][1], size, py, nr)
        assert np.array_equal(img_d[b].cpu().numpy(), ri), (b, i)
        assert np.array_equal(lab_d[b].cpu().numpy(), rl), (b, i)
        py2, nr2 = random.Random(), np.random.RandomState()
        py2.setstate(st[0]); nr2.set_state(st[1])
        modes.add(gen.draw(*slices[i][0].shape, py2, nr2)[0])
    assert modes == {0, 1, 2}          # the batch exercised: no-op, rot90+flip, rotation


def test_device_augmentation_equals_the_reference_fixture(golden_dir):
    """The device pipeline against outputs of the REFERENCE's own RandomGenerator (datasets/utils.py:99-117; fixture written by
    oracle/make_golden_r2.py from seeded `random` / `np.random`): images and masks bit for bit, all three branches."""
    from tests.trace_replay import unpack_labels2
    d = np.load(f"{golden_dir}/augment.npz")
    n = int(d["n"])
    slices = [(d[f"src_img{i}"], d[f"src_lab{i}"]) for i in range(n)]
    pool = DeviceSlicePool(slices, DEV)
    gen = RandomGeneratorDevice((224, 224))
    for k, seed in enumerate(d["seeds"]):
        img, lab = gen(pool, [k % n], py_rng=random.Random(int(seed)), np_rng=np.random.RandomState(int(seed)))
        got = img[0].cpu().numpy()
        assert np.array_equal(got.astype(np.float16), d[f"img{k}"]), (k, int(d["branch"][k]))
        assert abs(float(got.astype(np.float64).sum()) - float(d[f"img{k}_sum"])) < 1e-6, k          # full-precision checksum
        assert np.array_equal(lab[0].cpu().numpy().reshape(-1), unpack_labels2(d[f"lab{k}"], 224 * 224)), k
    assert set(d["branch"].tolist()) == {0, 1, 2}


def test_device_pool_loader_contract_and_one_training_step():
    """build_loader("device_synthetic"): DataLoader-shaped device loaders feeding a Mean-Teacher step."""
    from copy import deepcopy
    from hpfg_amd.datasets import build_loader
    from hpfg_amd.model import UNet
    from hpfg_amd.train import MeanTeacherStep
    from hpfg_amd.utils import AttrDict
    args = AttrDict(datasets="device_synthetic", in_channels=1, num_classes=4, batch_size=4, unlabel_batch_size=4, train_crop_size=(64, 64),
                    num_labeled=12, num_unlabeled=20, device="cuda:0", opt="sgd", lr=0.01, momentum=0.9, weight_decay=1e-4, sched="medical",
                    total_itrs=100, step_size=200, warmup_epochs=0, warmup_lr=1e-4, min_lr=1e-6, consistency=0.1, consistency_rampup=200.0,
                    ema_decay=0.99)
    lab, unl, _ = build_loader(args)
    assert len(lab) == 3 and len(unl) == 5 and len(lab.dataset) == 32
    batches = list(lab)
    assert len(batches) == 3
    x, y = batches[0]
    assert x.shape == (4, 1, 64, 64) and x.dtype == torch.float32 and x.is_cuda and y.shape == (4, 64, 64) and y.dtype == torch.uint8
    assert int(y.max()) <= 3
    xu, _ = next(iter(unl))
    torch.manual_seed(0)
    m = UNet(1, 4).to(DEV)
    e = deepcopy(m)
    for p in e.parameters():
        p.requires_grad = False
    m.train(); e.train()
    r = MeanTeacherStep(m, e, args).step(x, y, xu, 1, cons_w=0.05)
    assert torch.isfinite(r["loss"]).all()
