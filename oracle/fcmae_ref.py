"""TEST INFRASTRUCTURE — CPU restatement (plain torch, fp32) of the reference's FCMAE U-Net:
``viscy_models.unet.fcmae.FullyConvolutionalMAE``, dense (``mask_ratio = 0``: the network behind
``cytoland.engine.FcmaeUNet`` for fine-tuning / inference and the architecture of the published VSCyto3D checkpoint,
/root/reference/applications/cytoland/tests/test_inference_reproducibility.py:55-64) and masked (``mask_ratio > 0``:
self-supervised pre-training with ``MaskedMSELoss``, engine.py:104-125).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package.

Follows /root/reference/packages/viscy-models/src/viscy_models/unet/fcmae.py:
  * ``MaskedAdaptiveProjection`` (:311-385): Conv3d(k = s = stem kernel) -> depth folded into channels -> LayerNorm over
    channels (``masked_patchify`` / ``masked_unpatchify`` with ``unmasked=None`` are a (B,C,H,W) <-> (B,HW,C) round trip);
    the 2-D ``conv2d`` branch (Z == 1 inputs) is kept as parameters for state-dict compatibility;
  * ``MaskedConvNeXtV2Block`` (:144-230): dwconv7 -> LayerNorm -> timm ``GlobalResponseNormMlp`` (Linear) -> + shortcut
    (identity shortcut inside a stage: in == out, stride 1; no layer scale, drop-path 0);
  * ``MaskedConvNeXtV2Stage`` (:233-308): [LayerNorm2d + Conv2d(k = s = stride)] when the stage changes width / stride;
  * ``MaskedMultiscaleEncoder`` (:388-448), ``FullyConvolutionalMAE`` (:451-560): stem -> 4 stages -> features reversed ->
    ``UNeXt2Decoder`` (3 stages, pixel shuffle x2, ConvNeXt conv_mlp blocks) -> ``PixelToVoxelShuffleHead``
    (components/heads.py:656-685: pixel shuffle x xy_scaling + pad-pool, channels -> (C_out, D)) or ``PixelToVoxelHead``.

  * masking (:40-141): ``generate_mask`` (one random low-resolution mask per sample with exactly ``int(n * ratio)`` masked
    cells), ``upsample_mask`` (nearest, integer factor), ``masked_patchify`` / ``masked_unpatchify`` (keep only the
    unmasked tokens, row-major; scatter back into zeros).  With a mask every block computes
    ``x <- x * unmasked`` (in place — the identity shortcut aliases ``x``, so the shortcut is masked as well),
    dwconv, LayerNorm + GRN-MLP **on the unmasked tokens only** (the GRN statistics see L = (1 - ratio)·HW tokens),
    zero-filled scatter, + shortcut.

Pinned by ``oracle/validate_against_reference.py::g9_fcmae`` (the reference's own fcmae.py executed on stubbed timm /
monai modules == this file, exactly, incl. the state-dict key list and the masked path with the reference's own
``generate_mask`` draw) -> ``tests/golden/fcmae_forward.pt``, ``tests/golden/fcmae_masked.pt``.
timm's ``GlobalResponseNormMlp`` / ``create_conv2d`` internals are not in /root/reference: restated as in unext2_ref.py.
"""

from __future__ import annotations

import math
from typing import Sequence

import torch
from torch import Tensor, nn

from .unext2_ref import DropPath, GlobalResponseNormMlp, LayerNorm2d, PixelShuffleUp, PixelToVoxelHead, UNeXt2Decoder


def _init_weights(module: nn.Module) -> None:
    """fcmae.py:26-37"""
    if isinstance(module, nn.Conv2d):
        nn.init.trunc_normal_(module.weight, std=0.02)
        if module.bias is not None:
            nn.init.zeros_(module.bias)
    elif isinstance(module, nn.Linear):
        nn.init.trunc_normal_(module.weight, std=0.02)
        nn.init.zeros_(module.bias)
    elif isinstance(module, nn.LayerNorm):
        nn.init.ones_(module.weight)
        nn.init.zeros_(module.bias)


def generate_mask(target, stride: int, mask_ratio: float, device="cpu") -> Tensor:
    """fcmae.py:40-66: (B, 1, H/stride, W/stride) bool, True = masked; exactly int(n * ratio) cells per sample."""
    mh, mw = target[-2] // stride, target[-1] // stride
    n = mh * mw
    k = int(n * mask_ratio)
    return (torch.rand(target[0], n, device=device).argsort(1) < k).reshape(target[0], 1, mh, mw)


def upsample_mask(mask: Tensor, target) -> Tensor:
    """fcmae.py:69-92: nearest-neighbour integer upsampling of a (B,1,h,w) mask to the (..., H, W) of ``target``."""
    if tuple(target[-2:]) != tuple(mask.shape[-2:]):
        if target[-2] % mask.shape[-2] or target[-1] % mask.shape[-1]:
            raise ValueError(f"feature map shape {tuple(target)} must be divisible by mask shape {tuple(mask.shape)}.")
        mask = mask.repeat_interleave(target[-2] // mask.shape[-2], dim=-2).repeat_interleave(target[-1] // mask.shape[-1], dim=-1)
    return mask


def _tokens(x: Tensor, unmasked: Tensor | None) -> Tensor:
    """masked_patchify, fcmae.py:95-117: (B,C,H,W) -> (B, L, C), only the unmasked positions (row-major) when a mask is given."""
    b, c = x.shape[:2]
    t = x.permute(0, 2, 3, 1)
    if unmasked is None:
        return t.reshape(b, -1, c)
    return t[unmasked[:, 0]].reshape(b, -1, c)


def _untokens(t: Tensor, shape, unmasked: Tensor | None) -> Tensor:
    """masked_unpatchify, fcmae.py:120-141: back to (B,C,H,W); masked positions are zero."""
    b, c, h, w = shape
    if unmasked is None:
        return t.reshape(b, h, w, c).permute(0, 3, 1, 2)
    out = torch.zeros((b, h, w, c), dtype=t.dtype, device=t.device)
    out[unmasked[:, 0]] = t.reshape(-1, c)
    return out.permute(0, 3, 1, 2)


class MaskedMSELoss(nn.Module):
    """cytoland engine.py:104-125: mean over Z of the squared error, summed over masked pixels / number of masked pixels."""

    def forward(self, preds: Tensor, original: Tensor, mask: Tensor) -> Tensor:
        se = (preds - original) ** 2
        return (se.mean(2) * mask).sum() / mask.sum()


class MaskedConvNeXtV2Block(nn.Module):
    def __init__(self, channels: int, kernel_size: int = 7, mlp_ratio: int = 4, drop_path: float = 0.0):
        super().__init__()
        self.dwconv = nn.Conv2d(channels, channels, kernel_size, padding=kernel_size // 2, groups=channels)
        self.layernorm = nn.LayerNorm(channels)
        self.mlp = GlobalResponseNormMlp(channels, mlp_ratio * channels, channels, use_conv=False)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.shortcut = nn.Identity()

    def forward(self, x: Tensor, unmasked: Tensor | None = None) -> Tensor:
        if unmasked is not None:
            x = x * unmasked                        # in place in the reference: the (identity) shortcut is masked too
        shortcut = x
        x = self.dwconv(x)
        t = _tokens(x, unmasked)                    # (B, L, C); L = HW without a mask
        t = self.layernorm(t)
        t = self.mlp(t.unsqueeze(1)).squeeze(1)     # GRN statistics over the (1, L) "spatial" axes = over the kept tokens
        return self.drop_path(_untokens(t, x.shape, unmasked)) + shortcut


class MaskedConvNeXtV2Stage(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, stride: int, num_blocks: int, drop_path: float = 0.0):
        super().__init__()
        if in_channels != out_channels or stride > 1:
            k = stride if stride > 1 else 1
            self.downsample = nn.Sequential(LayerNorm2d(in_channels), nn.Conv2d(in_channels, out_channels, k, stride=stride))
        else:
            self.downsample = nn.Identity()
        self.blocks = nn.ModuleList([MaskedConvNeXtV2Block(out_channels, drop_path=drop_path) for _ in range(num_blocks)])

    def forward(self, x: Tensor, unmasked: Tensor | None = None) -> Tensor:
        x = self.downsample(x)
        if unmasked is not None:
            unmasked = upsample_mask(unmasked, x.shape)
        for blk in self.blocks:
            x = blk(x, unmasked)
        return x


class MaskedAdaptiveProjection(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, kernel_size_2d=(4, 4), kernel_depth: int = 5, in_stack_depth: int = 5):
        super().__init__()
        ratio = in_stack_depth // kernel_depth
        k3 = [kernel_depth, *kernel_size_2d]
        self.conv3d = nn.Conv3d(in_channels, out_channels // ratio, k3, stride=k3)
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size_2d, stride=kernel_size_2d)
        self.norm = nn.LayerNorm(out_channels)

    def forward(self, x: Tensor, unmasked: Tensor | None = None) -> Tensor:
        if x.shape[2] > 1:
            x = self.conv3d(x)
            b, c, d, h, w = x.shape
            x = x.reshape(b, c * d, h, w)
        else:
            x = self.conv2d(x.squeeze(2))
        if unmasked is not None:
            unmasked = upsample_mask(unmasked, x.shape)
        return _untokens(self.norm(_tokens(x, unmasked)), x.shape, unmasked)


class MaskedMultiscaleEncoder(nn.Module):
    def __init__(self, in_channels: int, stage_blocks, dims, stem_kernel_size, in_stack_depth: int, drop_path_rate: float = 0.0):
        super().__init__()
        self.stem = MaskedAdaptiveProjection(in_channels, dims[0], stem_kernel_size[1:], stem_kernel_size[0], in_stack_depth)
        self.stages = nn.ModuleList()
        chs = [dims[0], *dims]
        for i, n in enumerate(stage_blocks):
            # fcmae.py:404-414: drop_path_rates=[drop_path_rate] * num_blocks — the same rate for every block
            self.stages.append(MaskedConvNeXtV2Stage(chs[i], chs[i + 1], 1 if i == 0 else 2, n, drop_path_rate))
        self.total_stride = stem_kernel_size[1] * 2 ** (len(self.stages) - 1)
        self.apply(_init_weights)

    def forward(self, x: Tensor, mask_ratio: float = 0.0, mask: Tensor | None = None):
        """``mask`` (B,1,H/stride,W/stride) injects the draw ``generate_mask`` would make (tests); returns (features, mask at
        input resolution or None)."""
        unmasked = None
        if mask is None and mask_ratio > 0.0:
            mask = generate_mask(x.shape, self.total_stride, mask_ratio, x.device)
        if mask is not None:
            unmasked = ~mask
            mask = upsample_mask(mask, x.shape)
        x = self.stem(x)  # fcmae.py:441: the stem is called WITHOUT the mask (its LayerNorm is per token; stage 0 masks next)
        feats = []
        for st in self.stages:
            x = st(x, unmasked)
            feats.append(x)
        return feats, mask


class PixelToVoxelShuffleHead(nn.Module):
    """components/heads.py:656-685"""

    def __init__(self, in_channels: int, out_channels: int, out_stack_depth: int, xy_scaling: int, pool: bool):
        super().__init__()
        assert in_channels == out_stack_depth * out_channels * xy_scaling**2
        self.out_channels, self.out_stack_depth = out_channels, out_stack_depth
        self.upsample = PixelShuffleUp(xy_scaling, pool)

    def forward(self, x: Tensor) -> Tensor:
        x = self.upsample(x)
        b, _, h, w = x.shape
        return x.reshape(b, self.out_channels, self.out_stack_depth, h, w)


class FullyConvolutionalMAE(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, encoder_blocks: Sequence[int] = (3, 3, 9, 3),
                 dims: Sequence[int] = (96, 192, 384, 768), encoder_drop_path_rate: float = 0.0,
                 stem_kernel_size: Sequence[int] = (5, 4, 4), in_stack_depth: int = 5, decoder_conv_blocks: int = 1,
                 pretraining: bool = True, head_conv: bool = False, head_conv_expansion_ratio: int = 4,
                 head_conv_pool: bool = True):
        super().__init__()
        self.encoder = MaskedMultiscaleEncoder(in_channels, encoder_blocks, dims, stem_kernel_size, in_stack_depth,
                                               encoder_drop_path_rate)
        dec = list(reversed(dims))
        if head_conv:
            dec[-1] = (in_stack_depth + 2) * in_channels * 4 * head_conv_expansion_ratio
        else:
            dec[-1] = out_channels * in_stack_depth * stem_kernel_size[-1] ** 2
        self.decoder = UNeXt2Decoder(dec, decoder_conv_blocks, [2] * (len(dims) - 1) + [stem_kernel_size[-1]])
        if head_conv:
            self.head = PixelToVoxelHead(dec[-1], out_channels, in_stack_depth, head_conv_expansion_ratio, head_conv_pool)
        else:
            self.head = PixelToVoxelShuffleHead(dec[-1], out_channels, in_stack_depth, stem_kernel_size[-1], pool=True)
        self.out_stack_depth = in_stack_depth
        self.num_blocks = len(dims) * int(math.log2(stem_kernel_size[-1]))
        self.pretraining = pretraining

    def forward(self, x: Tensor, mask_ratio: float = 0.0, mask: Tensor | None = None):
        feats, mask = self.encoder(x, mask_ratio, mask)
        feats.reverse()
        x = self.head(self.decoder(feats))
        return (x, mask) if self.pretraining else x
