"""Host-side helpers of the hot loop with the reference's names and argument meaning (utils/utils.py).

loadyaml (:33-42), mk_path (:22-30), get_current_consistency_weight (:67-69), sigmoid_rampup (:72-79),
update_ema_variables (:82-86), linear_rampup (:89-95), BoxMaskGenerator (:98-176).
"""
from __future__ import annotations

import math
import os
import shutil

import numpy as np
import torch
import yaml


class AttrDict(dict):
    """Attribute-style dict (the reference uses easydict.EasyDict; nested dicts become AttrDicts too)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    __setattr__ = __setitem__


def loadyaml(file_path):
    if file_path is None:
        print("file path is empty")
        return None
    try:
        with open(file_path, "r", encoding="utf-8") as f:
            return AttrDict(yaml.safe_load(f))
    except IOError as e:
        print(e)
        return None


def mk_path(path, remove=False):
    try:
        if not os.path.exists(path):
            os.makedirs(path)
        elif remove:
            shutil.rmtree(path, ignore_errors=True)
    except Exception as e:
        print(e)


def sigmoid_rampup(current, rampup_length):
    if rampup_length == 0:
        return 1.0
    current = float(np.clip(current, 0.0, rampup_length))
    phase = 1.0 - current / rampup_length
    return float(math.exp(-5.0 * phase * phase))


def linear_rampup(current, rampup_length):
    assert current >= 0 and rampup_length >= 0
    return 1.0 if current >= rampup_length else current / rampup_length


def get_current_consistency_weight(epoch, args):
    return args.consistency * sigmoid_rampup(epoch, args.consistency_rampup)


def ema_alpha(global_step: int, alpha: float) -> float:
    return min(1 - 1 / (global_step + 1), alpha)


def _flat_pair(model, ema_model):
    fp = getattr(model, "flat_params", None)
    fe = getattr(ema_model, "flat_params", None)
    if fp is None or fe is None or fp.numel() != fe.numel() or not fp.is_cuda:
        return None
    return fp, fe


_alpha_cache = {}


def _alpha_dev(dev, alpha: float) -> torch.Tensor:
    t = _alpha_cache.get(dev)
    if t is None:
        t = _alpha_cache[dev] = torch.zeros(1, dtype=torch.float32, device=dev)
    t.fill_(alpha)
    return t


def update_ema_variables(model, ema_model, alpha, global_step, alpha_dev: torch.Tensor = None, numel: int = None):
    """ema = a*ema + (1-a)*param over PARAMETERS only (never BN buffers), a = min(1-1/(step+1), alpha).
    One HIP kernel over the flat parameter buffers; ``alpha_dev`` lets a captured graph read a from the device."""
    from .. import _lib as L
    pair = _flat_pair(model, ema_model)
    if pair is None:
        raise RuntimeError("update_ema_variables needs two hpfg_amd U-Nets on the GPU (no CPU fallback)")
    src, dst = pair
    n = src.numel() if numel is None else numel
    a = alpha_dev if alpha_dev is not None else _alpha_dev(src.device, ema_alpha(global_step, alpha))
    L.check(L.load().hpfg_ema_update(L.ptr(dst), L.ptr(src), n, L.ptr(a), torch.cuda.current_stream(src.device).cuda_stream), "ema_update")


def update_ema_variables_backbone(model, ema_model, alpha, global_step, alpha_dev: torch.Tensor = None):
    """Same, restricted to encoder+decoder parameters (main.py:68-76): they are the leading part of the flat buffers."""
    update_ema_variables(model, ema_model, alpha, global_step, alpha_dev, numel=model.backbone_numel())


class BoxMaskGenerator(object):
    """CutMix box masks, generated on the host with numpy in the reference's draw order (utils/utils.py:98-176)."""

    def __init__(self, prop_range, n_boxes=1, random_aspect_ratio=True, prop_by_area=True, within_bounds=True, invert=False):
        if isinstance(prop_range, float):
            prop_range = (prop_range, prop_range)
        self.prop_range, self.n_boxes = prop_range, n_boxes
        self.random_aspect_ratio, self.prop_by_area = random_aspect_ratio, prop_by_area
        self.within_bounds, self.invert = within_bounds, invert

    def draw_rects(self, n_masks, mask_shape, rng=None):
        """The random part: box corners [n_masks, n_boxes, (y0, x0, y1, x1)] in the reference's draw order."""
        rng = np.random if rng is None else rng
        lo, hi = self.prop_range
        nb = self.n_boxes
        if self.prop_by_area:
            props = rng.uniform(lo, hi, size=(n_masks, nb))
            zero = props == 0.0
            if self.random_aspect_ratio:
                y_props = np.exp(rng.uniform(low=0.0, high=1.0, size=(n_masks, nb)) * np.log(props))
                x_props = props / y_props
            else:
                y_props = x_props = np.sqrt(props)
            fac = np.sqrt(1.0 / nb)
            y_props, x_props = y_props * fac, x_props * fac
            y_props[zero] = 0
            x_props[zero] = 0
        else:
            if self.random_aspect_ratio:
                y_props = rng.uniform(lo, hi, size=(n_masks, nb))
                x_props = rng.uniform(lo, hi, size=(n_masks, nb))
            else:
                x_props = y_props = rng.uniform(lo, hi, size=(n_masks, nb))
            fac = np.sqrt(1.0 / nb)
            y_props, x_props = y_props * fac, x_props * fac
        shape = np.array(mask_shape)
        sizes = np.round(np.stack([y_props, x_props], axis=2) * shape[None, None, :])
        if self.within_bounds:
            pos = np.round((shape - sizes) * rng.uniform(low=0.0, high=1.0, size=sizes.shape))
            rects = np.append(pos, pos + sizes, axis=2)
        else:
            cen = np.round(shape * rng.uniform(low=0.0, high=1.0, size=sizes.shape))
            rects = np.append(cen - sizes * 0.5, cen + sizes * 0.5, axis=2)
        return rects

    def generate_params(self, n_masks, mask_shape, rng=None):
        rects = self.draw_rects(n_masks, mask_shape, rng)
        masks = np.zeros((n_masks, 1) + tuple(mask_shape)) if self.invert else np.ones((n_masks, 1) + tuple(mask_shape))
        for i, sample in enumerate(rects):
            for y0, x0, y1, x1 in sample:
                sl = (i, 0, slice(int(y0), int(y1)), slice(int(x0), int(x1)))
                masks[sl] = 1 - masks[sl]
        return masks

    def generate_params_device(self, n_masks, mask_shape, device, rng=None):
        """Same draws, same masks, rasterised on the device: 16 integers per mask cross the bus instead of
        H*W floats.  Box bounds go through Python's slice normalisation on the host so that the kernel reproduces numpy's
        slicing exactly (negative / out-of-range bounds when within_bounds=False)."""
        import ctypes as C

        import torch

        from .. import _lib as L
        rects = self.draw_rects(n_masks, mask_shape, rng)
        H, W = int(mask_shape[0]), int(mask_shape[1])
        norm = np.zeros((n_masks, self.n_boxes, 4), dtype=np.int32)
        for i, sample in enumerate(rects):
            for b, (y0, x0, y1, x1) in enumerate(sample):
                ys, ye, _ = slice(int(y0), int(y1)).indices(H)
                xs, xe, _ = slice(int(x0), int(x1)).indices(W)
                norm[i, b] = (ys, max(ye, ys), xs, max(xe, xs))
        r = torch.from_numpy(norm).to(device, non_blocking=True)
        out = torch.empty(n_masks, 1, H, W, dtype=torch.float32, device=device)
        L.check(L.load().hpfg_box_masks(L.ptr(r), n_masks, self.n_boxes, H, W, 1 if self.invert else 0, L.ptr(out),
                                        torch.cuda.current_stream(device).cuda_stream), "box_masks")
        return out

    def torch_masks_from_params(self, t_params, mask_shape, torch_device):
        return t_params
