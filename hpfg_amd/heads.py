"""Projection necks of UNet_Plus (reference model/unet.py:120-152).

GAP -> Linear -> ReLU -> Linear and AdaptiveAvgPool(4x4) -> 1x1 conv -> ReLU -> 1x1 conv on [N,256,14,14] features and on
the [N,4,224,224] logits: ~1 GFLOP per HPFG step of plain dense GEMMs.  They are issued as library GEMMs through
PyTorch-ROCm (rocBLAS / hipBLASLt), with torch autograd for their backward; the gradient w.r.t. their inputs flows back into
the HIP U-Net backward (engine.backward's ``dfeat4`` / ``dlogits``).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def projection_neck(m, x: torch.Tensor, s: int = 4):
    """x: [N,C,H,W] (any strides).  Returns (g [N,128], d [N,128,s*s]) like projection_conv.forward (unet.py:139-152)."""
    g = F.adaptive_avg_pool2d(x, 1).flatten(1)
    g = F.linear(F.relu(F.linear(g, m.mlp["0"].weight, m.mlp["0"].bias)), m.mlp["2"].weight, m.mlp["2"].bias)
    d = F.adaptive_avg_pool2d(x, s)
    d = F.conv2d(F.relu(F.conv2d(d, m.mlp_conv["0"].weight, m.mlp_conv["0"].bias)), m.mlp_conv["2"].weight, m.mlp_conv["2"].bias)
    return g, d.flatten(2)
