"""SegFormer-B0 for the CTCT cross-teaching branch.

Same module tree, parameter names and constructor order as the reference's ``model/segformer.py`` (MiT :213-272, SegFormerHead
:298-320, SegFormer :397-411), so ``state_dict()`` interchanges and a seed gives the same initial weights.  Where the work runs:

* hand-written HIP through ``hpfg_amd.ops_tokens``, forward and backward:
  - every dense product on the matrix cores in split-bf16 arithmetic (``csrc/gemm.hip``: ``hpfg_gemm_bf16x3`` for Y = X W^T + b and
    dX = dY W, ``hpfg_gemm_tn_bf16x3`` for dW = dY^T X together with db): q / kv / proj / fc1 / fc2, the spatial-reduction conv (kernel ==
    stride, so it is a GEMM over non-overlapping patches), the patch embeddings (after a HIP im2col), the head's per-stage projections and
    its 1x1 fuse / prediction convs.  No rocBLAS / hipBLASLt call is left (HPFG_MATH=f32 switches them to the exact-fp32 MFMA GEMM);
  - the attention core softmax(q k^T / sqrt(d)) v on the matrix cores (``csrc/attn.hip``: at most 64 keys after the spatial reduction,
    head dim 32 = one MFMA k-step; forward, dQ and dK / dV kernels without LDS transposes of the probabilities);
  - every LayerNorm, depthwise 3x3 + GELU of the Mix-FFN, the head's bilinear resizes and its BatchNorm(train) + ReLU + Dropout2d,
    im2col / col2im of the overlap patch embeddings, residual adds with their drop-path factor (``csrc/tokens.hip``);
* the head never builds the 4E-channel concat: ``linear_fuse`` (bias-free 1x1 conv) is applied per stage at the stage's resolution and the four
  E-channel maps are resized and added in one pass (``hpfg_resize_sum_fwd``; reference model/segformer.py:309-315 up to fp32 association); each
  stage's ``linear_c`` projection is composed with its slice of the fuse weights first (one C_i -> E GEMM per stage);
* still plain PyTorch-ROCm ops: the token <-> image reshape copies and the slices of the fuse weight (memory movement only).
No MIOpen call is left in the module: with MIOpen convolutions / BatchNorm the forward was not bit-reproducible between identical runs
(logits differing by ~4e-7), and one ReLU gate of the head flipping on such noise moves every gradient upstream by ~1e-3; without it the
forward is bit-identical run to run.

Tokens are kept as [B, N, C] == NHWC throughout.  Stochastic depth and the head's Dropout2d draw from the torch device generator;
``external_draws = (drop_path_draws, dropout_mask)`` replays given draws (parity tests against oracle/segformer_ref.py).
"""
from __future__ import annotations


import torch
import torch.nn as nn

from ..ops_tokens import attention, bn_relu_dropout, dwconv_gelu, im2col, layer_norm, linear, residual_scale, resize_bilinear, resize_sum

MIT_SETTINGS = {"B0": [[32, 64, 160, 256], [2, 2, 2, 2]]}
HEADS, SR = [1, 2, 5, 8], [8, 4, 2, 1]


class Attention(nn.Module):
    def __init__(self, dim, head, sr_ratio):
        super().__init__()
        self.head, self.sr_ratio, self.scale = head, sr_ratio, (dim // head) ** -0.5
        self.q = nn.Linear(dim, dim)
        self.kv = nn.Linear(dim, dim * 2)
        self.proj = nn.Linear(dim, dim)
        if sr_ratio > 1:
            self.sr = nn.Conv2d(dim, dim, sr_ratio, sr_ratio)
            self.norm = nn.LayerNorm(dim)

    def forward(self, x, H, W):
        B, N, C = x.shape
        q = linear(x, self.q.weight, self.q.bias)
        if self.sr_ratio > 1:
            s = self.sr_ratio
            # kernel == stride: the conv is a GEMM over non-overlapping s x s patches, rows ordered (u, v, channel)
            p = x.view(B, H // s, s, W // s, s, C).permute(0, 1, 3, 2, 4, 5).reshape(B, (H // s) * (W // s), s * s * C)
            x = linear(p, self.sr.weight.permute(0, 2, 3, 1).reshape(C, s * s * C), self.sr.bias)
            x = layer_norm(x, self.norm.weight, self.norm.bias)
        kv = linear(x, self.kv.weight, self.kv.bias)
        return linear(attention(q, kv, self.head, self.scale), self.proj.weight, self.proj.bias)


class DWConv(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, groups=dim)


class MLP(nn.Module):
    def __init__(self, c1, c2):
        super().__init__()
        self.fc1 = nn.Linear(c1, c2)
        self.dwconv = DWConv(c2)
        self.fc2 = nn.Linear(c2, c1)

    def forward(self, x, H, W):
        B, N, _ = x.shape
        h = linear(x, self.fc1.weight, self.fc1.bias)
        h = dwconv_gelu(h.view(B, H, W, -1), self.dwconv.dwconv.weight, self.dwconv.dwconv.bias).view(B, N, -1)
        return linear(h, self.fc2.weight, self.fc2.bias)


class PatchEmbed(nn.Module):
    def __init__(self, c1=3, c2=32, patch_size=7, stride=4):
        super().__init__()
        self.proj = nn.Conv2d(c1, c2, patch_size, stride, patch_size // 2)       # parameter holder (reference names / initialiser)
        self.norm = nn.LayerNorm(c2)
        self.k, self.s = patch_size, stride

    def forward(self, x):
        """x NHWC [B,H,W,C] -> tokens [B, H'*W', c2]: HIP im2col, split-bf16 MFMA GEMM (csrc/gemm.hip), HIP LayerNorm."""
        B, H, W, _ = x.shape
        Ho, Wo = (H + 2 * (self.k // 2) - self.k) // self.s + 1, (W + 2 * (self.k // 2) - self.k) // self.s + 1
        wt = self.proj.weight.permute(0, 2, 3, 1).reshape(self.proj.weight.shape[0], -1)          # rows ordered (u, v, c) like the patches
        t = linear(im2col(x, self.k, self.s), wt, self.proj.bias)
        return layer_norm(t, self.norm.weight, self.norm.bias), Ho, Wo


class Block(nn.Module):
    def __init__(self, dim, head, sr_ratio=1, dpr=0.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = Attention(dim, head, sr_ratio)
        self.dpr = float(dpr)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = MLP(dim, int(dim * 4))

    def _keep_scale(self, batch, draw, device):
        """Per-sample factor of a residual branch (DropPath, segformer.py:23-30): floor(keep + U) / keep, or None when nothing is dropped."""
        if self.dpr == 0.0 or not self.training:
            return None
        kp = 1.0 - self.dpr
        r = torch.rand((batch, 1, 1), dtype=torch.float32, device=device) if draw is None else draw.to(device)
        return (kp + r).floor() / kp

    def forward(self, x, H, W, draws=(None, None)):
        B = x.shape[0]
        a = self.attn(layer_norm(x, self.norm1.weight, self.norm1.bias), H, W)
        x = residual_scale(x, a, self._keep_scale(B, draws[0], x.device))
        m = self.mlp(layer_norm(x, self.norm2.weight, self.norm2.bias), H, W)
        return residual_scale(x, m, self._keep_scale(B, draws[1], x.device))


class MiT(nn.Module):
    def __init__(self, model_name: str = "B0", in_channels: int = 3):
        super().__init__()
        if model_name not in MIT_SETTINGS:
            raise NotImplementedError(f"MiT-{model_name}: only B0 (the CTCT configuration) is built")
        embed_dims, depths = MIT_SETTINGS[model_name]
        self.embed_dims, self.depths = embed_dims, depths
        self.patch_embed1 = PatchEmbed(in_channels, embed_dims[0], 7, 4)
        self.patch_embed2 = PatchEmbed(embed_dims[0], embed_dims[1], 3, 2)
        self.patch_embed3 = PatchEmbed(embed_dims[1], embed_dims[2], 3, 2)
        self.patch_embed4 = PatchEmbed(embed_dims[2], embed_dims[3], 3, 2)
        dpr = [x.item() for x in torch.linspace(0, 0.1, sum(depths))]
        cur = 0
        for i in range(4):
            setattr(self, f"block{i + 1}", nn.ModuleList([Block(embed_dims[i], HEADS[i], SR[i], dpr[cur + j]) for j in range(depths[i])]))
            setattr(self, f"norm{i + 1}", nn.LayerNorm(embed_dims[i]))
            cur += depths[i]

    def forward(self, x, draws=None):
        """Returns the four stage outputs as tokens [(tokens [B,N,C], H, W)] (the reference returns them as NCHW images)."""
        B = x.shape[0]
        x = x.permute(0, 2, 3, 1)                                # NCHW input -> NHWC (a view; contiguous already when C == 1)
        feats, bi = [], 0
        for i in range(4):
            x, H, W = getattr(self, f"patch_embed{i + 1}")(x)
            for blk in getattr(self, f"block{i + 1}"):
                d = (None, None) if draws is None else (draws[2 * bi], draws[2 * bi + 1])
                x = blk(x, H, W, d)
                bi += 1
            n = getattr(self, f"norm{i + 1}")
            t = layer_norm(x, n.weight, n.bias)
            feats.append((t, H, W))
            x = t.view(B, H, W, -1)                               # the tokens ARE the NHWC image of the next patch embedding
        return feats


class FFN(nn.Module):
    def __init__(self, dim, embed_dim):
        super().__init__()
        self.proj = nn.Linear(dim, embed_dim)


class ConvModule(nn.Module):
    def __init__(self, c1, c2):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, 1, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.activate = nn.ReLU(True)


class SegFormerHead(nn.Module):
    def __init__(self, dims, image_size=(224, 224), embed_dim: int = 256, num_classes: int = 19):
        super().__init__()
        self.image_size = list(image_size)
        for i, dim in enumerate(dims):
            self.add_module(f"linear_c{i + 1}", FFN(dim, embed_dim))
        self.linear_fuse = ConvModule(embed_dim * 4, embed_dim)
        self.fuse_lowres = True      # False: the reference's literal concat + one GEMM (tests compare both forms with the oracle)
        self.linear_pred = nn.Conv2d(embed_dim, num_classes, 1)
        self.dropout = nn.Dropout2d(0.1)

    def forward(self, feats, dropout_mask=None):
        B = feats[0][0].shape[0]
        H, W = feats[0][1], feats[0][2]
        cv, bn = self.linear_fuse.conv, self.linear_fuse.bn
        wf = cv.weight.reshape(cv.weight.shape[0], -1)                           # [E, 4E]; the concat order is c4, c3, c2, c1
        E = wf.shape[1] // len(feats)
        if self.fuse_lowres:
            # linear_fuse(cat(up(y4), up(y3), up(y2), y1)) = sum_i up(W_i y_i): the bias-free 1x1 conv acts per pixel and the bilinear resize per
            # channel, so they commute -- each stage's slice of the fuse weights is applied at the stage's own resolution (1/4 .. 1/64 of the
            # pixels) and only the E-channel results are resized and added.  No [B, H*W, 4E] concat (411 MB at 32 x 56 x 56 tokens) is written,
            # read by the GEMM and by its two backward GEMMs; the same sum in a different association (fp32, |diff| ~ 1e-6 relative).
            # ... and the stage's projection (linear_c_i: C_i -> E, with bias) composes with its slice of the fuse weights into ONE map C_i -> E:
            # W_eff = W_f,i W_c,i (E x C_i), b_eff = W_f,i b_c,i -- two E x E x C_i products on the weights instead of an E-wide GEMM over every
            # token (100 k tokens at stage 1) forward, for dX and for dW; autograd differentiates the composition through the same HIP ops.
            zs = []
            for i, (t, h, w) in enumerate(feats):
                p = getattr(self, f"linear_c{i + 1}").proj
                slot = len(feats) - 1 - i
                wfi = wf[:, slot * E:(slot + 1) * E]                             # [E, E]
                w_eff = linear(p.weight.t(), wfi).t()                            # (W_c^T W_f^T)^T = W_f W_c: [E, C_i]
                b_eff = linear(p.bias[None, :], wfi)[0]                          # W_f b_c: [E]
                zs.append(linear(t, w_eff, b_eff).view(B, h, w, -1))             # NHWC [B, h, w, E]
            z = resize_sum(zs[0], *zs[1:]).view(B, H * W, -1)                   # one HIP pass: zs[0] + sum of the resized others
        else:
            outs = []
            for i, (t, h, w) in enumerate(feats):
                p = getattr(self, f"linear_c{i + 1}").proj
                y = linear(t, p.weight, p.bias)                                      # [B, h*w, E] tokens
                if i > 0:
                    y = resize_bilinear(y.view(B, h, w, -1), H, W).view(B, H * W, -1)          # HIP, NHWC
                outs.append(y)
            z = linear(torch.cat(outs[::-1], dim=2), wf)      # 1x1 conv without bias == GEMM over tokens
        if self.training:          # nn.BatchNorm2d (train) + ReLU + nn.Dropout2d(0.1) (whole channels per sample): HIP kernels over the tokens
            if dropout_mask is None:
                dropout_mask = torch.empty(B, z.shape[2], 1, 1, device=z.device).bernoulli_(0.9)
            seg, mu, var = bn_relu_dropout(z, bn.weight, bn.bias, dropout_mask.to(z.device), 0.9, bn.eps)
            with torch.no_grad():
                n = z.shape[0] * z.shape[1]
                bn.running_mean.mul_(1 - bn.momentum).add_(mu, alpha=bn.momentum)
                bn.running_var.mul_(1 - bn.momentum).add_(var * (n / max(n - 1, 1)), alpha=bn.momentum)
                bn.num_batches_tracked += 1
        else:                      # eval: a per-channel affine map of the running statistics (elementwise torch ops)
            seg = torch.relu((z - bn.running_mean) * torch.rsqrt(bn.running_var + bn.eps) * bn.weight + bn.bias)
        pr = self.linear_pred
        wp, bp, ncls = pr.weight.reshape(pr.weight.shape[0], -1), pr.bias, pr.weight.shape[0]
        if ncls % 4:          # the HIP resize moves 4 channels per lane: pad the prediction layer with zero rows (2-class LIDC heads), slice them off after
            pad = 4 - ncls % 4
            wp, bp = torch.cat([wp, wp.new_zeros(pad, wp.shape[1])], 0), torch.cat([bp, bp.new_zeros(pad)], 0)
        seg = linear(seg, wp, bp)
        out = resize_bilinear(seg.view(B, H, W, -1), self.image_size[0], self.image_size[1])      # HIP resize on NHWC; the result is viewed as NCHW
        return out[..., :ncls].permute(0, 3, 1, 2)


class SegFormer(nn.Module):
    def __init__(self, image_size=(224, 224), in_channels=3, num_classes=4, model_name: str = "B0"):
        super().__init__()
        self.encoder = MiT(model_name=model_name, in_channels=in_channels)
        self.decoder = SegFormerHead(self.encoder.embed_dims, image_size=image_size, embed_dim=256, num_classes=num_classes)
        self.external_draws = None        # (drop_path_draws, dropout_mask): replay given random draws (tests)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("hpfg_amd.SegFormer runs on the GPU only: its LayerNorm / attention / DWConv kernels have no CPU fallback")
        dp, mask = self.external_draws if self.external_draws is not None else (None, None)
        return self.decoder(self.encoder(x.float(), dp), mask)

    def val(self, x):
        return self.forward(x)

    def bump_graph_seed(self):
        """GraphedStep hook (the U-Net engines advance their dropout seed word here): nothing to do -- drop-path and Dropout2d draw from
        the torch device generator, whose state a captured graph advances by itself at every replay."""
