"""CPU: host-side logic of the product (config, factories, schedulers, ramp-ups, CutMix masks, synthetic data, module plumbing,
dropout RNG law) against the oracle / the reference's recorded behaviour.  No GPU, no HIP compute calls."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from hpfg_amd.datasets import build_loader
from hpfg_amd.datasets.synthetic import synth_batch
from hpfg_amd.model import UNet, UNet_Plus, build_model
from hpfg_amd.utils import (AttrDict, BoxMaskGenerator, CosineWarmupLR_Scheduler, Medical_LR, build_lr_scheduler, ema_alpha,
                            get_current_consistency_weight, linear_rampup, loadyaml, sigmoid_rampup)
from oracle import laws_ref, rng_ref, unet_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_yaml_configs_parse_with_reference_keys():
    for name in os.listdir(os.path.join(ROOT, "config")):
        a = loadyaml(os.path.join(ROOT, "config", name))
        for k in ("datasets", "num_classes", "train_crop_size", "batch_size", "seed", "total_itrs", "step_size"):
            assert k in a, (name, k)
        assert a.ckpt == "None"            # YAML `ckpt: None` is the STRING "None", as in the reference
        if "model1" in a:
            assert a.model1.opt == "sgd" and isinstance(a.model1, AttrDict)


def test_build_model_keys_and_state_dict_layout():
    torch.manual_seed(1)
    m = build_model(AttrDict(model="unet", in_channels=1, num_classes=4))
    st = unet_ref.init_state(1, 1, 4)
    sd = m.state_dict()
    assert list(sd.keys()) == list(st.keys()) and len(sd) == 136 and len(list(m.parameters())) == 82
    assert all(torch.equal(sd[k], st[k]) for k in st)           # same RNG consumption as the reference constructor
    torch.manual_seed(1)
    mp = build_model(AttrDict(model="unet_plus", in_channels=1, num_classes=4))
    stp = unet_ref.init_state(1, 1, 4, True)
    assert list(mp.state_dict().keys()) == list(stp.keys()) and len(stp) == 152 and len(list(mp.parameters())) == 98
    assert all(torch.equal(mp.state_dict()[k], stp[k]) for k in stp)
    assert hasattr(mp, "encoder") and hasattr(mp, "decoder") and callable(mp.val)
    with pytest.raises(NotImplementedError):
        build_model(AttrDict(model="swinunet", in_channels=1, num_classes=4, train_crop_size=[224, 224]))


def test_flat_parameter_views_deepcopy_and_state_dict_roundtrip():
    torch.manual_seed(3)
    m = UNet(1, 4)
    assert m._is_flat() and m.flat_params.numel() == 1813764 == m.backbone_numel()
    e = copy.deepcopy(m)
    assert e._is_flat() and e.flat_params.data_ptr() != m.flat_params.data_ptr()
    assert torch.equal(e.flat_params, m.flat_params)
    with torch.no_grad():
        m.flat_params.mul_(0.5)
    p = next(m.parameters())
    assert torch.equal(p.detach().reshape(-1), m.flat_params[: p.numel()])          # parameters are views of the flat buffer
    e.load_state_dict(m.state_dict())
    assert e._is_flat() and torch.equal(e.flat_params, m.flat_params)
    mp = UNet_Plus(1, 4)
    assert mp.backbone_numel() == 1813764 and mp.flat_params.numel() == 3663620


def test_cpu_input_is_refused_loudly():
    m = UNet(1, 4)
    with pytest.raises(RuntimeError, match="HIP"):
        m(torch.zeros(1, 1, 32, 32))


def test_schedulers_reproduce_reference_quirks(golden_dir):
    a = json.load(open(f"{golden_dir}/anchors.json"))
    lin = torch.nn.Linear(1, 1)
    opt = torch.optim.SGD(lin.parameters(), lr=0.01, momentum=0.9)
    sch = Medical_LR(opt, 0.01, 30000)
    got = []
    for _ in range(6):
        got.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    assert np.allclose(got, a["medical_lr_first6"], rtol=0, atol=1e-12) and got[0] > 0.01
    opt = torch.optim.SGD(lin.parameters(), lr=0.01, momentum=0.9)
    args = AttrDict(sched="cosine", lr=0.01, warmup_epochs=0, warmup_lr=1e-4, min_lr=1e-6, step_size=200, total_itrs=30000)
    sch = build_lr_scheduler(args, opt)
    assert isinstance(sch, CosineWarmupLR_Scheduler)
    got = []
    for _ in range(6):
        got.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    assert np.allclose(got, a["cosine_lr_first6"], rtol=0, atol=1e-12)


def test_rampups_ema_alpha_and_box_masks(golden_dir):
    for e in (0, 1, 50, 199, 200, 500):
        assert sigmoid_rampup(e, 200.0) == laws_ref.sigmoid_rampup(e, 200.0)
        assert linear_rampup(e, 200.0) == laws_ref.linear_rampup(e, 200.0)
    assert get_current_consistency_weight(40, AttrDict(consistency=0.1, consistency_rampup=200.0)) == 0.1 * laws_ref.sigmoid_rampup(40, 200.0)
    assert [ema_alpha(s, 0.99) for s in (1, 2, 3, 200)] == [laws_ref.ema_alpha(s, 0.99) for s in (1, 2, 3, 200)]
    d = np.load(f"{golden_dir}/box_masks.npz")
    ref = np.unpackbits(d["masks"])[: int(np.prod(d["shape"]))].reshape(d["shape"])
    gen = BoxMaskGenerator(prop_range=(0.25, 0.5), n_boxes=4, random_aspect_ratio=True, prop_by_area=True, within_bounds=True, invert=True)
    np.random.seed(1)
    got = gen.generate_params(5, (64, 64))
    assert got.shape == (5, 1, 64, 64) and np.array_equal(got.astype(np.uint8), ref)       # the reference's own masks (fixture)


def test_synthetic_loaders_keep_the_batch_contract():
    a = AttrDict(datasets="synthetic", num_classes=4, train_crop_size=[64, 64], batch_size=2, unlabel_batch_size=4, in_channels=1,
                 synthetic_labeled=8, synthetic_unlabeled=16)
    lab, unl, test = build_loader(a)
    x, y = next(iter(lab))
    assert x.shape == (2, 1, 64, 64) and x.dtype == torch.float32 and y.shape == (2, 64, 64) and y.dtype == torch.uint8
    xu, _ = next(iter(unl))
    assert xu.shape == (4, 1, 64, 64) and len(unl) == 4 and len(lab.dataset) == 8
    v, l = next(iter(test))
    assert v.dim() == 4 and v.shape[0] == 1 and l.shape == v.shape
    a2 = AttrDict(datasets="sup_synthetic", num_classes=4, train_crop_size=[32, 32], batch_size=8, in_channels=1)
    tr, te = build_loader(a2)
    assert next(iter(tr))[0].shape == (8, 1, 32, 32)
    with pytest.raises(NotImplementedError):
        build_loader(AttrDict(datasets="lidc"))          # real-data keys outside the hot-path build raise like an unknown key (builder.py:76-77)
    with pytest.raises(NotImplementedError):
        build_loader(AttrDict(datasets="no-such-dataset"))
    x1, y1 = synth_batch(5, 2, 32, 32, 3, 2, 8)
    x2, y2 = synth_batch(5, 2, 32, 32, 3, 2, 8)
    assert torch.equal(x1, x2) and torch.equal(y1, y2) and x1.shape == (2, 3, 32, 32) and int(y1.max()) <= 1


def test_dropout_rng_law_statistics_and_determinism():
    for p in (0.05, 0.1, 0.2, 0.3, 0.5):
        m = rng_ref.keep_mask_nhwc(400000, p, 12345)
        assert abs((1 - m.mean()) - p) < 0.004, p
        assert np.array_equal(m, rng_ref.keep_mask_nhwc(400000, p, 12345))
        assert not np.array_equal(m, rng_ref.keep_mask_nhwc(400000, p, 12346))
    m = rng_ref.keep_mask_nchw(2, 16, 8, 8, 0.3, 7)
    assert m.shape == (2, 16, 8, 8)


def test_local_bn_mode_averages_the_summed_gradient():
    """sync_bn=False: ranks keep their own BatchNorm statistics and loss; the SUM all-reduce of the gradients is turned into the
    mean by the SGD kernel's gradient scale (no extra pass)."""
    from hpfg_amd.train import _StepBase

    class _DP:
        world_size, force_sync, sync_bn = 4, False, False

    class _Opt:
        grad_scale = 1.0

    s = _StepBase.__new__(_StepBase)
    s.dp = _DP()
    o1, o2 = _Opt(), _Opt()
    s._set_grad_scale(o1, o2)
    assert o1.grad_scale == 0.25 and o2.grad_scale == 0.25
    s.dp.sync_bn = True
    o3 = _Opt()
    s._set_grad_scale(o3)
    assert o3.grad_scale == 1.0


def test_eval_zoom_index_is_scipys_mapping():
    """hpfg_amd.val resizes through an index map taken from scipy.ndimage.zoom(order=0) itself (val.py:274,280 semantics)."""
    import numpy as np
    from scipy.ndimage import zoom
    from hpfg_amd import val as V

    g = np.random.default_rng(0)
    for (h, w), (H, W) in (((40, 36), (32, 32)), ((32, 32), (50, 44)), ((17, 23), (64, 48)), ((32, 32), (32, 32)), ((230, 232), (224, 224))):
        a = g.standard_normal((h, w)).astype(np.float32)
        idx = V._zoom_index((h, w), (H, W))
        got = np.where(idx >= 0, a.reshape(-1)[np.maximum(idx, 0)], 0.0).reshape(H, W)
        assert np.array_equal(got, zoom(a, (H / h, W / w), order=0))
    cm = np.array([[5, 1, 0], [2, 4, 0], [1, 1, 0]])
    assert V.dice_from_counts(cm, 1) == 2 * 4 / (6 + 6) and V.dice_from_counts(cm, 2) == 0.0
