"""Oracle (test infrastructure): functional CPU restatement of the reference U-Net.

Follows the graph of /root/reference/model/unet.py:
  * two-conv block  conv3x3+b -> BN(train) -> LeakyReLU(0.01) -> Dropout(p)
                    -> conv3x3+b -> BN -> LeakyReLU              (unet.py:12-28)
  * encoder level k>0 = MaxPool2d(2) then block                  (unet.py:31-42)
  * decoder level = conv1x1 -> bilinear x2 (align_corners=True)
                    -> cat([skip, up]) -> block(p=0)             (unet.py:45-58)
  * encoder channels [16,32,64,128,256], dropout [.05,.1,.2,.3,.5]
                                                                 (unet.py:159-165)
  * logits = conv3x3(16 -> n_class)                              (unet.py:99,114-117)
  * UNet_Plus adds two projection necks (GAP->fc->ReLU->fc and
    AdaptiveAvgPool(4)->1x1->ReLU->1x1)                          (unet.py:120-152,178-206)

State is a flat ``dict[name] -> tensor`` that uses the reference's
``state_dict()`` key names so reference checkpoints can be compared key by key.
Everything is fp32, NCHW, CPU.  Dropout masks are explicit inputs (or drawn from
the torch CPU generator in the reference's order) so that the HIP path, which
uses its own counter-based RNG, can be checked with identical masks.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from . import bf16x3_ref

WIDTHS = (16, 32, 64, 128, 256)
ENC_DROPOUT = (0.05, 0.1, 0.2, 0.3, 0.5)
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LEAKY = 0.01


def enc_block_prefix(level: int) -> str:
    return "encoder.in_conv.conv_conv" if level == 0 else f"encoder.down{level}.maxpool_conv.1.conv_conv"


def dec_block_prefix(k: int) -> str:
    return f"decoder.up{k}.conv.conv_conv"


def conv_bn_layers() -> List[Tuple[str, str, int]]:
    """(conv key prefix, bn key prefix, dropout-site index or -1) for all 18 conv+BN layers
    in forward order.  Dropout sites are the first conv of each encoder block."""
    out = []
    for lvl in range(5):
        p = enc_block_prefix(lvl)
        out.append((f"{p}.0", f"{p}.1", lvl))
        out.append((f"{p}.4", f"{p}.5", -1))
    for k in range(1, 5):
        p = dec_block_prefix(k)
        out.append((f"{p}.0", f"{p}.1", -1))
        out.append((f"{p}.4", f"{p}.5", -1))
    return out


def init_state(seed: Optional[int], in_channels: int = 1, num_classes: int = 4, plus: bool = False) -> "OrderedDict[str, torch.Tensor]":
    """Create parameters + BN buffers with torch's default layer initialisers, consuming
    the CPU generator in the reference's constructor order (encoder blocks, then per
    decoder level conv1x1 + block, then out_conv, then the two necks)."""
    if seed is not None:
        torch.manual_seed(seed)
    st: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def add_conv(name, cin, cout, k):
        m = torch.nn.Conv2d(cin, cout, kernel_size=k, padding=k // 2)
        st[f"{name}.weight"] = m.weight.detach().clone()
        st[f"{name}.bias"] = m.bias.detach().clone()

    def add_linear(name, cin, cout):
        m = torch.nn.Linear(cin, cout)
        st[f"{name}.weight"] = m.weight.detach().clone()
        st[f"{name}.bias"] = m.bias.detach().clone()

    def add_bn(name, c):
        st[f"{name}.weight"] = torch.ones(c)
        st[f"{name}.bias"] = torch.zeros(c)
        st[f"{name}.running_mean"] = torch.zeros(c)
        st[f"{name}.running_var"] = torch.ones(c)
        st[f"{name}.num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    def add_block(prefix, cin, cout):
        add_conv(f"{prefix}.0", cin, cout, 3)
        add_bn(f"{prefix}.1", cout)
        add_conv(f"{prefix}.4", cout, cout, 3)
        add_bn(f"{prefix}.5", cout)

    cprev = in_channels
    for lvl, c in enumerate(WIDTHS):
        add_block(enc_block_prefix(lvl), cprev, c)
        cprev = c
    for k in range(1, 5):
        c1, c2 = WIDTHS[5 - k], WIDTHS[4 - k]
        add_conv(f"decoder.up{k}.conv1x1", c1, c2, 1)
        add_block(dec_block_prefix(k), 2 * c2, c2)
    add_conv("decoder.out_conv", WIDTHS[0], num_classes, 3)
    if plus:
        for name, cin, hid in (("dense_projection_high", WIDTHS[-1], 2048), ("dense_projection_head", num_classes, 1024)):
            add_linear(f"{name}.mlp.0", cin, hid)
            add_linear(f"{name}.mlp.2", hid, 128)
            add_conv(f"{name}.mlp_conv.0", cin, hid, 1)
            add_conv(f"{name}.mlp_conv.2", hid, 128, 1)
    return st


def param_names(state: Dict[str, torch.Tensor]) -> List[str]:
    return [k for k in state if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))]


def clone_state(state, requires_grad: bool = False):
    out = OrderedDict()
    for k, v in state.items():
        t = v.detach().clone()
        if requires_grad and t.is_floating_point() and k in set(param_names(state)):
            t.requires_grad_(True)
        out[k] = t
    return out


def dropout_site_shapes(n: int, h: int, w: int) -> List[Tuple[int, int, int, int]]:
    """Shapes (NCHW) of the five active dropout sites for an n x * x h x w input."""
    return [(n, WIDTHS[l], h >> l, w >> l) for l in range(5)]


def draw_dropout_masks(n: int, h: int, w: int) -> List[torch.Tensor]:
    """Keep-masks (0/1 floats) drawn like torch's CPU dropout does: one
    ``bernoulli_(1-p)`` over a tensor of the activation's shape per site, in forward order."""
    return [torch.empty(s).bernoulli_(1.0 - p) for s, p in zip(dropout_site_shapes(n, h, w), ENC_DROPOUT)]


def _conv_bn_act(x, st, conv, bn, train, track):
    z = bf16x3_ref.conv2d(x, st[f"{conv}.weight"], st[f"{conv}.bias"], padding=1, first_layer=conv == "encoder.in_conv.conv_conv.0")
    if train:
        rm = st[f"{bn}.running_mean"] if track else None
        rv = st[f"{bn}.running_var"] if track else None
        y = F.batch_norm(z, rm, rv, st[f"{bn}.weight"], st[f"{bn}.bias"], True, BN_MOMENTUM, BN_EPS)
        if track:
            st[f"{bn}.num_batches_tracked"] += 1
    else:
        y = F.batch_norm(z, st[f"{bn}.running_mean"], st[f"{bn}.running_var"], st[f"{bn}.weight"], st[f"{bn}.bias"], False, BN_MOMENTUM, BN_EPS)
    return F.leaky_relu(y, LEAKY), z


def _block(x, st, prefix, train, track, keep_mask, p, taps):
    a, z1 = _conv_bn_act(x, st, f"{prefix}.0", f"{prefix}.1", train, track)
    if train and p > 0.0:
        if keep_mask is None:
            keep_mask = torch.empty_like(a).bernoulli_(1.0 - p)
        a = a * (keep_mask / (1.0 - p))
    a2, z2 = _conv_bn_act(a, st, f"{prefix}.4", f"{prefix}.5", train, track)
    if taps is not None:
        taps[f"{prefix}.0"] = z1
        taps[f"{prefix}.4"] = z2
    return a2


def unet_forward(st, x, train: bool = True, drop_masks: Optional[List[Optional[torch.Tensor]]] = None,
                 track_running: bool = True, taps: Optional[dict] = None, plus: bool = False):
    """Returns logits [N,ncls,H,W]; with ``plus`` also ((g_high,d_high),(g_head,d_head)).
    ``drop_masks``: five keep-masks (see draw_dropout_masks); None draws them from torch's
    generator site by site like nn.Dropout does; entries ignored when ``train`` is False.
    ``taps``: optional dict filled with the raw (pre-BN) conv outputs keyed by conv prefix."""
    feats = []
    h = x
    for lvl in range(5):
        if lvl > 0:
            h = F.max_pool2d(h, 2)
        km = None if drop_masks is None else drop_masks[lvl]
        h = _block(h, st, enc_block_prefix(lvl), train, track_running, km, ENC_DROPOUT[lvl], taps)
        feats.append(h)
    h = feats[4]
    for k in range(1, 5):
        u = bf16x3_ref.conv2d(h, st[f"decoder.up{k}.conv1x1.weight"], st[f"decoder.up{k}.conv1x1.bias"])
        if taps is not None:
            taps[f"decoder.up{k}.conv1x1"] = u
        u = F.interpolate(u, scale_factor=2, mode="bilinear", align_corners=True)
        h = torch.cat([feats[4 - k], u], dim=1)
        h = _block(h, st, dec_block_prefix(k), train, track_running, None, 0.0, taps)
    logits = bf16x3_ref.conv2d(h, st["decoder.out_conv.weight"], st["decoder.out_conv.bias"], padding=1)
    if not plus:
        return logits
    return logits, projection_neck(st, "dense_projection_high", feats[4]), projection_neck(st, "dense_projection_head", logits)


def projection_neck(st, name, x, s: int = 4):
    """GAP -> fc -> ReLU -> fc  and  AdaptiveAvgPool(s) -> 1x1 -> ReLU -> 1x1 (unet.py:139-152)."""
    g = F.adaptive_avg_pool2d(x, 1).flatten(1)
    g = F.linear(F.relu(F.linear(g, st[f"{name}.mlp.0.weight"], st[f"{name}.mlp.0.bias"])),
                 st[f"{name}.mlp.2.weight"], st[f"{name}.mlp.2.bias"])
    d = F.adaptive_avg_pool2d(x, s)
    d = F.conv2d(F.relu(F.conv2d(d, st[f"{name}.mlp_conv.0.weight"], st[f"{name}.mlp_conv.0.bias"])),
                 st[f"{name}.mlp_conv.2.weight"], st[f"{name}.mlp_conv.2.bias"])
    return g, d.flatten(2)
