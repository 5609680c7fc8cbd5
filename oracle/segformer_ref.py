"""CPU restatement of the reference's SegFormer (MiT-B0 backbone + all-MLP head) -- TEST INFRASTRUCTURE ONLY, like the rest of oracle/.

Groundwork for SURVEY.md section 8f row 1 (the CTCT cross-teaching branch, BASELINE.json configs[4]); no HIP kernels consume it yet.
Follows /root/reference/model/segformer.py: Attention with spatial reduction :92-128, DWConv / MLP :131-156, PatchEmbed :159-177,
Block :180-200, MiT :213-272 (B0: dims 32/64/160/256, depths 2/2/2/2, heads 1/2/5/8, sr 8/4/2/1, drop-path linspace(0, 0.1, 8)),
FFN / ConvModule / SegFormerHead :275-320, SegFormer :397-411.  Functional over a state dict with the reference's state_dict keys.
Stochastic parts take explicit draws so that a HIP implementation can replay them: drop-path keep factors (one [B,1,1] tensor per
residual branch whose rate is > 0, in call order) and the Dropout2d(0.1) channel mask of the head.
Pinned by tests/golden/segformer_b0.npz (written by oracle/make_golden_segformer.py from the reference module itself).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

DIMS, DEPTHS, HEADS, SR = [32, 64, 160, 256], [2, 2, 2, 2], [1, 2, 5, 8], [8, 4, 2, 1]
PATCH = [(7, 4), (3, 2), (3, 2), (3, 2)]          # (kernel, stride) of the four overlap patch embeddings; padding = kernel // 2
DROP_PATH_RATE, HEAD_DROPOUT, EMBED = 0.1, 0.1, 256


def drop_path_rates() -> List[float]:
    return [x.item() for x in torch.linspace(0, DROP_PATH_RATE, sum(DEPTHS))]          # segformer.py:227


def init_state(seed: Optional[int], in_channels: int = 3, num_classes: int = 4) -> "OrderedDict[str, torch.Tensor]":
    """Parameters and buffers with torch's default initialisers, consuming the CPU generator in the reference's constructor order
    (MiT: the four patch embeddings, then per stage its blocks and final norm; head: linear_c1..4, linear_fuse, linear_pred)."""
    if seed is not None:
        torch.manual_seed(seed)
    st: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def conv(name, cin, cout, k, s=1, p=0, groups=1, bias=True):
        m = torch.nn.Conv2d(cin, cout, k, s, p, groups=groups, bias=bias)
        st[f"{name}.weight"] = m.weight.detach().clone()
        if bias:
            st[f"{name}.bias"] = m.bias.detach().clone()

    def linear(name, cin, cout):
        m = torch.nn.Linear(cin, cout)
        st[f"{name}.weight"], st[f"{name}.bias"] = m.weight.detach().clone(), m.bias.detach().clone()

    def ln(name, c):
        st[f"{name}.weight"], st[f"{name}.bias"] = torch.ones(c), torch.zeros(c)

    cin = in_channels
    for i, (c, (k, s)) in enumerate(zip(DIMS, PATCH)):
        conv(f"encoder.patch_embed{i + 1}.proj", cin, c, k, s, k // 2)
        ln(f"encoder.patch_embed{i + 1}.norm", c)
        cin = c
    for i, c in enumerate(DIMS):
        for d in range(DEPTHS[i]):
            b = f"encoder.block{i + 1}.{d}"
            ln(f"{b}.norm1", c)
            linear(f"{b}.attn.q", c, c)
            linear(f"{b}.attn.kv", c, 2 * c)
            linear(f"{b}.attn.proj", c, c)
            if SR[i] > 1:
                conv(f"{b}.attn.sr", c, c, SR[i], SR[i])
                ln(f"{b}.attn.norm", c)
            ln(f"{b}.norm2", c)
            linear(f"{b}.mlp.fc1", c, 4 * c)
            conv(f"{b}.mlp.dwconv.dwconv", 4 * c, 4 * c, 3, 1, 1, groups=4 * c)
            linear(f"{b}.mlp.fc2", 4 * c, c)
        ln(f"encoder.norm{i + 1}", c)
    for i, c in enumerate(DIMS):
        linear(f"decoder.linear_c{i + 1}.proj", c, EMBED)
    conv("decoder.linear_fuse.conv", 4 * EMBED, EMBED, 1, bias=False)
    st["decoder.linear_fuse.bn.weight"], st["decoder.linear_fuse.bn.bias"] = torch.ones(EMBED), torch.zeros(EMBED)
    st["decoder.linear_fuse.bn.running_mean"], st["decoder.linear_fuse.bn.running_var"] = torch.zeros(EMBED), torch.ones(EMBED)
    st["decoder.linear_fuse.bn.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    conv("decoder.linear_pred", EMBED, num_classes, 1)
    return st


def draw_randomness(batch: int):
    """The draws of one train-mode forward in the reference's order: torch.rand([B,1,1]) per residual branch with rate > 0
    (segformer.py:23-30; the first block has rate 0 and draws nothing), then the Dropout2d channel mask of the head (:307,318)."""
    dp = []
    for r in drop_path_rates():
        for _ in range(2):                      # attention branch, then MLP branch (segformer.py:197-198)
            dp.append(None if r == 0.0 else torch.rand((batch, 1, 1)))
    mask = torch.empty(batch, EMBED, 1, 1).bernoulli_(1.0 - HEAD_DROPOUT)
    return dp, mask


def _attention(st, b, x, H, W, heads, sr):
    B, N, C = x.shape
    q = F.linear(x, st[f"{b}.q.weight"], st[f"{b}.q.bias"]).reshape(B, N, heads, C // heads).permute(0, 2, 1, 3)
    if sr > 1:
        x = x.permute(0, 2, 1).reshape(B, C, H, W)
        x = F.conv2d(x, st[f"{b}.sr.weight"], st[f"{b}.sr.bias"], stride=sr).reshape(B, C, -1).permute(0, 2, 1)
        x = F.layer_norm(x, (C,), st[f"{b}.norm.weight"], st[f"{b}.norm.bias"])
    kv = F.linear(x, st[f"{b}.kv.weight"], st[f"{b}.kv.bias"]).reshape(B, -1, 2, heads, C // heads).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    attn = ((q @ k.transpose(-2, -1)) * (C // heads) ** -0.5).softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(x, st[f"{b}.proj.weight"], st[f"{b}.proj.bias"])


def _mlp(st, b, x, H, W):
    x = F.linear(x, st[f"{b}.fc1.weight"], st[f"{b}.fc1.bias"])
    B, _, C = x.shape
    y = F.conv2d(x.transpose(1, 2).reshape(B, C, H, W), st[f"{b}.dwconv.dwconv.weight"], st[f"{b}.dwconv.dwconv.bias"], padding=1, groups=C)
    return F.linear(F.gelu(y.flatten(2).transpose(1, 2)), st[f"{b}.fc2.weight"], st[f"{b}.fc2.bias"])


def _drop_path(x, rate, draw, train):
    if rate == 0.0 or not train:
        return x
    kp = 1.0 - rate
    return x.div(kp) * (kp + draw).floor()


def segformer_forward(st: Dict[str, torch.Tensor], x: torch.Tensor, train: bool = True, drop_path_draws=None, dropout_mask=None,
                      track_running: bool = True, taps: Optional[dict] = None) -> torch.Tensor:
    """Logits [B,num_classes,H,W] (the head's output is resized to the input size, segformer.py:319)."""
    B, _, Hin, Win = x.shape
    rates = drop_path_rates()
    feats, bi = [], 0
    for i, c in enumerate(DIMS):
        k, s = PATCH[i]
        p = f"encoder.patch_embed{i + 1}"
        x = F.conv2d(x, st[f"{p}.proj.weight"], st[f"{p}.proj.bias"], stride=s, padding=k // 2)
        H, W = x.shape[-2:]
        x = F.layer_norm(x.flatten(2).transpose(1, 2), (c,), st[f"{p}.norm.weight"], st[f"{p}.norm.bias"])
        for d in range(DEPTHS[i]):
            b = f"encoder.block{i + 1}.{d}"
            r = rates[bi]
            d0 = drop_path_draws[2 * bi] if (train and drop_path_draws is not None) else None
            d1 = drop_path_draws[2 * bi + 1] if (train and drop_path_draws is not None) else None
            a = _attention(st, f"{b}.attn", F.layer_norm(x, (c,), st[f"{b}.norm1.weight"], st[f"{b}.norm1.bias"]), H, W, HEADS[i], SR[i])
            x = x + _drop_path(a, r, d0, train and d0 is not None)
            m = _mlp(st, f"{b}.mlp", F.layer_norm(x, (c,), st[f"{b}.norm2.weight"], st[f"{b}.norm2.bias"]), H, W)
            x = x + _drop_path(m, r, d1, train and d1 is not None)
            bi += 1
        x = F.layer_norm(x, (c,), st[f"encoder.norm{i + 1}.weight"], st[f"encoder.norm{i + 1}.bias"]).reshape(B, H, W, c).permute(0, 3, 1, 2)
        feats.append(x)
        if taps is not None:
            taps[f"stage{i + 1}"] = x
    H, W = feats[0].shape[-2:]
    outs = []
    for i, f in enumerate(feats):
        y = F.linear(f.flatten(2).transpose(1, 2), st[f"decoder.linear_c{i + 1}.proj.weight"], st[f"decoder.linear_c{i + 1}.proj.bias"])
        y = y.permute(0, 2, 1).reshape(B, EMBED, *f.shape[-2:])
        outs.append(y if i == 0 else F.interpolate(y, size=(H, W), mode="bilinear", align_corners=False))
    y = F.conv2d(torch.cat(outs[::-1], 1), st["decoder.linear_fuse.conv.weight"])
    bn = "decoder.linear_fuse.bn"
    if train:
        rm = st[f"{bn}.running_mean"] if track_running else None
        rv = st[f"{bn}.running_var"] if track_running else None
        y = F.batch_norm(y, rm, rv, st[f"{bn}.weight"], st[f"{bn}.bias"], True, 0.1, 1e-5)
        if track_running:
            st[f"{bn}.num_batches_tracked"] += 1
    else:
        y = F.batch_norm(y, st[f"{bn}.running_mean"], st[f"{bn}.running_var"], st[f"{bn}.weight"], st[f"{bn}.bias"], False, 0.1, 1e-5)
    y = F.relu(y)
    if train and dropout_mask is not None:
        y = y * dropout_mask / (1.0 - HEAD_DROPOUT)
    y = F.conv2d(y, st["decoder.linear_pred.weight"], st["decoder.linear_pred.bias"])
    return F.interpolate(y, size=(Hin, Win), mode="bilinear", align_corners=False)
