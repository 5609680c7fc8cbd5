"""Evaluation of the hot path's models on the device (reference val.py:154-193, 268-287, 376-387).

``test_single_volume`` keeps the reference's signature and arithmetic -- every slice is resized to ``patch_size`` with
``scipy.ndimage.zoom(order=0)``, run through ``net`` in eval mode, arg-maxed, resized back, and scored per foreground class with
medpy's binary Dice ``2 n(A&B) / (n(A) + n(B))`` -- but without the per-slice host round trips: the nearest-neighbour resize
becomes ONE device gather through an index map that scipy itself produces for the (shape, patch) pair (so the mapping is scipy's
by construction), all slices go through the HIP engine in fixed-size batches, arg-max and the class-confusion counts are HIP
kernels, and only C*C integers per volume cross to the host.
HD95 (medpy ``hd95``, CPU distance transforms; out of scope of the hot path) is reported as 0.0 unless
``with_hd95=True``, which runs the scipy restatement below on the host.
"""
from __future__ import annotations

import ctypes as C
from functools import lru_cache
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from .train import argmax_labels

EVAL_BATCH = 8      # slices per engine launch (one engine shape for every volume; the last batch is zero padded)


@lru_cache(maxsize=64)
def _zoom_index(src_hw: Tuple[int, int], dst_hw: Tuple[int, int]) -> np.ndarray:
    """Flat source index of every destination pixel under scipy.ndimage.zoom(order=0) with the reference's factors
    (val.py:274,280: zoom(slice, (dst/src, dst/src), order=0))."""
    from scipy.ndimage import zoom
    h, w = src_hw
    idx = np.arange(1, h * w + 1, dtype=np.float64).reshape(h, w)
    out = zoom(idx, (dst_hw[0] / h, dst_hw[1] / w), order=0)
    assert out.shape == tuple(dst_hw), (out.shape, dst_hw)
    return np.rint(out).astype(np.int64).reshape(-1) - 1      # -1: scipy wrote its constant 0 (coordinate a rounding error past the edge)


def _resize_nearest(t: torch.Tensor, dst_hw: Sequence[int]) -> torch.Tensor:
    """[S,h,w] -> [S,H,W] with scipy's order-0 mapping, one gather on the device."""
    s, h, w = t.shape
    if (h, w) == tuple(dst_hw):
        return t
    idx = torch.from_numpy(_zoom_index((h, w), (int(dst_hw[0]), int(dst_hw[1])))).to(t.device)
    out = t.reshape(s, h * w).index_select(1, idx.clamp(min=0))
    out = torch.where(idx.unsqueeze(0) >= 0, out, torch.zeros((), dtype=t.dtype, device=t.device))
    return out.reshape(s, int(dst_hw[0]), int(dst_hw[1]))


def predict_volume(image: torch.Tensor, net, patch_size: Sequence[int] = (256, 256)) -> torch.Tensor:
    """image [S,h,w] (float, any device) -> predicted labels uint8 [S,h,w] on the model's device."""
    dev = next(net.parameters()).device
    if dev.type != "cuda":
        raise RuntimeError("hpfg_amd.val runs on the HIP library only (no CPU fallback)")
    vol = image.to(dev, torch.float32)
    s, h, w = vol.shape
    x = _resize_nearest(vol, patch_size)
    was_training = net.training
    net.eval()
    fwd = net.val if hasattr(net, "val") else net
    preds: List[torch.Tensor] = []
    with torch.no_grad():
        for i in range(0, s, EVAL_BATCH):
            chunk = x[i:i + EVAL_BATCH]
            n = chunk.shape[0]
            if n < EVAL_BATCH:
                chunk = torch.cat([chunk, chunk.new_zeros(EVAL_BATCH - n, *chunk.shape[1:])], 0)
            logits = fwd(chunk.unsqueeze(1).contiguous())
            preds.append(argmax_labels(logits)[:n])       # argmax(softmax(z)) == argmax(z)
    net.train(was_training)
    return _resize_nearest(torch.cat(preds, 0), (h, w)).contiguous()


def confusion_counts(pred: torch.Tensor, gt: torch.Tensor, classes: int) -> np.ndarray:
    """counts[g, p] over all voxels (uint8 label tensors on the device) -> int64 [classes, classes] on the host."""
    assert pred.shape == gt.shape and pred.is_cuda and gt.is_cuda
    p8, g8 = pred.to(torch.uint8).contiguous(), gt.to(torch.uint8).contiguous()
    out = torch.zeros(classes * classes, dtype=torch.int64, device=pred.device)
    L.check(L.load().hpfg_confusion_counts(L.ptr(p8), L.ptr(g8), p8.numel(), classes, L.ptr(out),
                                           torch.cuda.current_stream(pred.device).cuda_stream), "confusion_counts")
    return out.cpu().numpy().reshape(classes, classes)


def dice_from_counts(cm: np.ndarray, cls: int) -> float:
    """The reference's per-class rule (val.py:376-387): 0 if the class is never predicted, else medpy dc."""
    n_pred, n_gt, inter = int(cm[:, cls].sum()), int(cm[cls, :].sum()), int(cm[cls, cls])
    if n_pred == 0:
        return 0.0
    return 2.0 * inter / float(n_pred + n_gt)


def hd95_host(pred: np.ndarray, gt: np.ndarray) -> float:
    """medpy.metric.binary.hd95 restated with scipy (voxel spacing 1, connectivity 1): 95th percentile of the symmetric surface
    distances.  Host only; raises like medpy when one of the objects is empty."""
    from scipy.ndimage import binary_erosion, distance_transform_edt, generate_binary_structure

    def surf_dist(a, b):
        a, b = np.atleast_1d(a.astype(bool)), np.atleast_1d(b.astype(bool))
        if not a.any():
            raise RuntimeError("The first supplied array does not contain any binary object.")
        if not b.any():
            raise RuntimeError("The second supplied array does not contain any binary object.")
        fp = generate_binary_structure(a.ndim, 1)
        ab = a ^ binary_erosion(a, structure=fp, iterations=1)
        bb = b ^ binary_erosion(b, structure=fp, iterations=1)
        dt = distance_transform_edt(~bb)
        return dt[ab]

    return float(np.percentile(np.hstack((surf_dist(pred, gt), surf_dist(gt, pred))), 95))


def test_single_volume(image, label, net, classes, patch_size=(256, 256), with_hd95: bool = False, _pred_out: list = None):
    """Reference signature (val.py:268).  image, label: [1,S,h,w].  Returns [(dice, hd95)] for classes 1..classes-1.
    _pred_out: optional list that receives the predicted label volume (uint8 [S,h,w], device) -- test_acdc's TensorBoard image hook."""
    dev = next(net.parameters()).device
    img = image.squeeze(0)
    lab = label.squeeze(0).to(dev)
    pred = predict_volume(img, net, patch_size)
    if _pred_out is not None:
        _pred_out.append(pred)
    cm = confusion_counts(pred, lab.to(torch.uint8), classes)
    out = []
    pred_h = lab_h = None
    for c in range(1, classes):
        d = dice_from_counts(cm, c)
        hd = 0.0
        if with_hd95 and cm[:, c].sum() > 0:
            if pred_h is None:
                pred_h, lab_h = pred.cpu().numpy(), lab.cpu().numpy()
            hd = hd95_host(pred_h == c, lab_h == c)
        out.append((d, hd))
    return out


test_single_volume.__test__ = False      # reference name, not a pytest case


def test_acdc(model, test_loader, args, cur_itrs=0, name="test", with_hd95: bool = False):
    """Reference signature (val.py:154): mean foreground Dice and mean HD95 over the volumes of ``test_loader`` (bs=1 volumes
    ``(image [1,S,h,w], label [1,S,h,w])``).  With ``args.writer`` (anything with TensorBoard's ``add_image``) the first volume's first slice
    is logged the way main.py:309-325 does: ``<name>/Image`` (the slice resized to ``test_crop_size``, [1,H,W]), ``<name>/label_pred`` and
    ``<name>/label_true`` (the dataset's ``label_to_img`` palette images, HWC) -- the prediction is slice 0 of the volume prediction above
    (resize -> eval forward -> arg-max -> resize back: the arithmetic of the reference's separate forward of that slice)."""
    metric_list = 0.0
    n = 0
    writer = getattr(args, "writer", None)
    to_img = getattr(getattr(test_loader, "dataset", None), "label_to_img", None)
    for image, label in test_loader:
        hook = n == 0 and writer is not None and hasattr(writer, "add_image") and to_img is not None
        keep = [] if hook else None
        metric_list = metric_list + np.array(test_single_volume(image, label, model, classes=args.num_classes,
                                                               patch_size=args.test_crop_size, with_hd95=with_hd95, _pred_out=keep))
        if hook:
            first = image[0, 0].to(keep[0].device, torch.float32)
            shown = _resize_nearest(first.unsqueeze(0), args.test_crop_size)          # [1,H,W]: what the network saw
            writer.add_image("{}/Image".format(name), shown.cpu(), cur_itrs)
            writer.add_image("{}/label_pred".format(name), to_img(keep[0][0].cpu().numpy()), cur_itrs, dataformats="HWC")
            writer.add_image("{}/label_true".format(name), to_img(label[0, 0].cpu().numpy()), cur_itrs, dataformats="HWC")
        n += 1
    metric_list = metric_list / max(n, 1)
    logger = getattr(args, "logger", None)
    if logger is not None:
        logger.info("class dice:{}".format(metric_list[:, 0]))
    return float(np.mean(metric_list, axis=0)[0]), float(np.mean(metric_list, axis=0)[1])


test_acdc.__test__ = False
