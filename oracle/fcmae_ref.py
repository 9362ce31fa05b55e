"""TEST INFRASTRUCTURE — CPU restatement (plain torch, fp32) of the DENSE path of the reference's FCMAE U-Net:
``viscy_models.unet.fcmae.FullyConvolutionalMAE`` with ``mask_ratio = 0`` (no sparse masking), i.e. the network behind
``cytoland.engine.FcmaeUNet`` for fine-tuning / inference and the architecture of the published VSCyto3D checkpoint
(/root/reference/applications/cytoland/tests/test_inference_reproducibility.py:55-64).

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

Pinned by ``oracle/validate_against_reference.py::g9_fcmae`` (the reference's own fcmae.py executed on stubbed timm /
monai modules == this file, exactly, incl. the state-dict key list) -> ``tests/golden/fcmae_forward.pt``.
timm's ``GlobalResponseNormMlp`` / ``create_conv2d`` internals are not in /root/reference: restated as in unext2_ref.py.
"""

from __future__ import annotations

import math
from typing import Sequence

import torch
from torch import Tensor, nn

from .unext2_ref import GlobalResponseNormMlp, LayerNorm2d, PixelShuffleUp, PixelToVoxelHead, UNeXt2Decoder


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


class MaskedConvNeXtV2Block(nn.Module):
    def __init__(self, channels: int, kernel_size: int = 7, mlp_ratio: int = 4):
        super().__init__()
        self.dwconv = nn.Conv2d(channels, channels, kernel_size, padding=kernel_size // 2, groups=channels)
        self.layernorm = nn.LayerNorm(channels)
        self.mlp = GlobalResponseNormMlp(channels, mlp_ratio * channels, channels, use_conv=False)
        self.drop_path = nn.Identity()
        self.shortcut = nn.Identity()

    def forward(self, x: Tensor) -> Tensor:
        shortcut = x
        x = self.dwconv(x)
        b, c, h, w = x.shape
        x = x.flatten(2).permute(0, 2, 1)           # masked_patchify(unmasked=None): (B, HW, C)
        x = self.layernorm(x)
        x = self.mlp(x.unsqueeze(1)).squeeze(1)     # GRN statistics over the (1, HW) "spatial" axes = over all pixels
        x = x.permute(0, 2, 1).reshape(b, c, h, w)  # masked_unpatchify
        return x + shortcut


class MaskedConvNeXtV2Stage(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, stride: int, num_blocks: int):
        super().__init__()
        if in_channels != out_channels or stride > 1:
            k = stride if stride > 1 else 1
            self.downsample = nn.Sequential(LayerNorm2d(in_channels), nn.Conv2d(in_channels, out_channels, k, stride=stride))
        else:
            self.downsample = nn.Identity()
        self.blocks = nn.ModuleList([MaskedConvNeXtV2Block(out_channels) for _ in range(num_blocks)])

    def forward(self, x: Tensor) -> Tensor:
        x = self.downsample(x)
        for blk in self.blocks:
            x = blk(x)
        return x


class MaskedAdaptiveProjection(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, kernel_size_2d=(4, 4), kernel_depth: int = 5, in_stack_depth: int = 5):
        super().__init__()
        ratio = in_stack_depth // kernel_depth
        k3 = [kernel_depth, *kernel_size_2d]
        self.conv3d = nn.Conv3d(in_channels, out_channels // ratio, k3, stride=k3)
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size_2d, stride=kernel_size_2d)
        self.norm = nn.LayerNorm(out_channels)

    def forward(self, x: Tensor) -> Tensor:
        if x.shape[2] > 1:
            x = self.conv3d(x)
            b, c, d, h, w = x.shape
            x = x.reshape(b, c * d, h, w)
        else:
            x = self.conv2d(x.squeeze(2))
        b, c, h, w = x.shape
        x = self.norm(x.flatten(2).permute(0, 2, 1))
        return x.permute(0, 2, 1).reshape(b, c, h, w)


class MaskedMultiscaleEncoder(nn.Module):
    def __init__(self, in_channels: int, stage_blocks, dims, stem_kernel_size, in_stack_depth: int):
        super().__init__()
        self.stem = MaskedAdaptiveProjection(in_channels, dims[0], stem_kernel_size[1:], stem_kernel_size[0], in_stack_depth)
        self.stages = nn.ModuleList()
        chs = [dims[0], *dims]
        for i, n in enumerate(stage_blocks):
            self.stages.append(MaskedConvNeXtV2Stage(chs[i], chs[i + 1], 1 if i == 0 else 2, n))
        self.total_stride = stem_kernel_size[1] * 2 ** (len(self.stages) - 1)
        self.apply(_init_weights)

    def forward(self, x: Tensor) -> list[Tensor]:
        x = self.stem(x)
        feats = []
        for st in self.stages:
            x = st(x)
            feats.append(x)
        return feats


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
        assert encoder_drop_path_rate == 0.0
        self.encoder = MaskedMultiscaleEncoder(in_channels, encoder_blocks, dims, stem_kernel_size, in_stack_depth)
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

    def forward(self, x: Tensor, mask_ratio: float = 0.0):
        if mask_ratio > 0.0:
            raise NotImplementedError("the oracle restates the dense path only")
        feats = self.encoder(x)
        feats.reverse()
        x = self.head(self.decoder(feats))
        return (x, None) if self.pretraining else x
