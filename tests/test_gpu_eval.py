"""GPU: device evaluation path (hpfg_amd.val) against the CPU oracle restatement of the reference's val.py."""
import numpy as np
import pytest
import torch

from hpfg_amd import val as V
from hpfg_amd.model import UNet
from oracle import eval_ref, losses_ref

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _volume(seed, s, h, w, ncls):
    g = np.random.default_rng(seed)
    coarse = g.integers(0, ncls, (s, 5, 5))
    lab = np.kron(coarse, np.ones((h // 5 + 1, w // 5 + 1), dtype=np.int64))[:, :h, :w].astype(np.uint8)
    img = (lab / (ncls - 1) + 0.1 * g.standard_normal((s, h, w))).astype(np.float32)
    return img, lab


def _trained_like_model(seed):
    """A U-Net whose BatchNorm running statistics are not the initial (0, 1): a few train-mode forwards on random data."""
    torch.manual_seed(seed)
    m = UNet(1, 4).to(DEV)
    m.math = "f32"
    m.train()
    with torch.no_grad():
        for k in range(3):
            m(torch.randn(8, 1, 32, 32, device=DEV, generator=None) * (1 + k))
    return m


@pytest.mark.parametrize("shape,patch", [((11, 40, 36), (32, 32)), ((5, 32, 32), (32, 32)), ((9, 50, 44), (48, 48))])
def test_single_volume_matches_oracle(shape, patch):
    s, h, w = shape
    m = _trained_like_model(5)
    img, lab = _volume(3, s, h, w, 4)
    got = V.test_single_volume(torch.from_numpy(img)[None], torch.from_numpy(lab)[None], m, classes=4, patch_size=patch)
    state = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    ref_dice, ref_pred = eval_ref.test_single_volume(img, lab, state, 4, patch)
    pred = V.predict_volume(torch.from_numpy(img), m, patch).cpu().numpy()
    # identical label maps except where two logits tie to < 1e-4 (fp32 summation order); Dice within 1e-3 (BASELINE tolerance)
    assert (pred != ref_pred).mean() < 2e-3
    for (d, hd), r in zip(got, ref_dice):
        assert abs(d - r) < 1e-3 and hd == 0.0


def test_confusion_counts_and_dice_rule():
    g = torch.Generator().manual_seed(0)
    pred = torch.randint(0, 4, (7, 33, 29), generator=g, dtype=torch.uint8)
    gt = torch.randint(0, 4, (7, 33, 29), generator=g, dtype=torch.uint8)
    pred[pred == 3] = 0                      # class 3 never predicted -> dice 0 by the reference's rule
    cm = V.confusion_counts(pred.to(DEV), gt.to(DEV), 4)
    ref = np.zeros((4, 4), dtype=np.int64)
    np.add.at(ref, (gt.numpy().ravel(), pred.numpy().ravel()), 1)
    assert (cm == ref).all()
    for c in range(1, 4):
        p, t = pred.numpy() == c, gt.numpy() == c
        want = losses_ref.binary_dice(p, t) if p.sum() > 0 else 0.0
        assert abs(V.dice_from_counts(cm, c) - want) < 1e-12
    assert V.dice_from_counts(cm, 3) == 0.0


def test_eval_restores_training_mode_and_hd95_host():
    m = _trained_like_model(2)
    img, lab = _volume(1, 3, 32, 32, 4)
    out = V.test_single_volume(torch.from_numpy(img)[None], torch.from_numpy(lab)[None], m, classes=4, patch_size=(32, 32), with_hd95=True)
    assert m.training and len(out) == 3 and all(hd >= 0.0 for _, hd in out)
    a = np.zeros((8, 8), bool); b = np.zeros((8, 8), bool)
    a[2:5, 2:5] = True; b[2:5, 3:6] = True
    assert abs(V.hd95_host(a, b) - 1.0) < 1e-9


def test_eval_image_hooks_when_a_writer_is_supplied():
    """main.py:309-325: with a TensorBoard-like writer the first volume's first slice is logged as <name>/Image, <name>/label_pred and
    <name>/label_true (palette images); the prediction shown is slice 0 of the scored volume prediction."""
    from hpfg_amd.datasets.synthetic import SyntheticVolumes
    from hpfg_amd.utils import AttrDict

    class Writer:
        def __init__(self):
            self.images = {}

        def add_image(self, tag, img, step, dataformats="CHW"):
            self.images[tag] = (np.asarray(img), step, dataformats)

        def add_scalar(self, *a, **k):
            pass

    m = _trained_like_model(4)
    loader = torch.utils.data.DataLoader(SyntheticVolumes(2, 3, (40, 36)), batch_size=1)
    w = Writer()
    args = AttrDict(num_classes=4, test_crop_size=(32, 32), writer=w, device=DEV)
    dice, _ = V.test_acdc(m, loader, args, cur_itrs=200, name="model1")
    assert 0.0 <= dice <= 1.0 and m.training
    assert set(w.images) == {"model1/Image", "model1/label_pred", "model1/label_true"}
    img, step, fmt = w.images["model1/Image"]
    assert img.shape == (1, 32, 32) and step == 200 and fmt == "CHW"
    for tag in ("model1/label_pred", "model1/label_true"):
        im, step, fmt = w.images[tag]
        assert im.shape == (40, 36, 3) and im.dtype == np.uint8 and fmt == "HWC" and step == 200
    image0, label0 = loader.dataset[0]
    pred0 = V.predict_volume(image0, m, (32, 32))[0].cpu().numpy()
    assert np.array_equal(w.images["model1/label_pred"][0], loader.dataset.label_to_img(pred0))
    assert np.array_equal(w.images["model1/label_true"][0], loader.dataset.label_to_img(label0[0].numpy()))
    # no writer: nothing is logged and nothing fails
    V.test_acdc(m, loader, AttrDict(num_classes=4, test_crop_size=(32, 32), device=DEV), cur_itrs=1)
