"""ORACLE — test infrastructure, not product code.

Pure-torch fp32 CPU restatement of the reference UNeXt2 forward path.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; nothing under ``viscy_amd/`` does.

What it restates (reference file:line, all under /root/reference/packages/viscy-models/src/viscy_models):
  * ``UNeXt2``            unet/unext2.py:13-82
  * ``UNeXt2Stem``        components/stems.py:8-50
  * ``UNeXt2UpStage`` / ``UNeXt2Decoder`` / ``icnr_init``   components/blocks.py:14-243
  * ``PixelToVoxelHead``  components/heads.py:594-641
  * the ConvNeXt-V2 block / stage / feature-list encoder that the reference obtains from
    **timm 1.0.27** (uv.lock:6269) — not vendored in /root/reference.  The block math
    follows the in-repo restatement unet/fcmae.py:174-221 (dense path) and the stage
    logic unet/fcmae.py:260-274, and is cross-checked against
    ``transformers.models.convnextv2`` (oracle/validate_against_reference.py).
  * MONAI 1.5.2 (uv.lock:3358) pieces: ``UpSample(mode="pixelshuffle", pre_conv=None | "default")``
    (= [3x3 convolution +] pixel shuffle + optional pad-pool) and ``Convolution`` (Conv3d → InstanceNorm3d
    → PReLU, "NDA" ordering).

Parity pinning: see oracle/validate_against_reference.py (run in the build container where
/root/reference exists) and tests/golden/*.  The reference's own tests pin only shapes and
state-dict keys for UNeXt2 (tests/test_state_dict_compat.py:33-55), which
tests/test_oracle.py re-states.
"""

from __future__ import annotations

import math
from typing import Sequence

import torch
import torch.nn.functional as F
from torch import Tensor, nn

# timm convnextv2_* configurations (depths, dims, conv_mlp) — timm/models/convnext.py model
# entrypoints; use_grn=True, ls_init_value=None for every v2 variant.
CONVNEXTV2_CFGS = {
    "convnextv2_atto": ((2, 2, 6, 2), (40, 80, 160, 320), True),
    "convnextv2_femto": ((2, 2, 6, 2), (48, 96, 192, 384), True),
    "convnextv2_pico": ((2, 2, 6, 2), (64, 128, 256, 512), True),
    "convnextv2_nano": ((2, 2, 8, 2), (80, 160, 320, 640), True),
    "convnextv2_tiny": ((3, 3, 9, 3), (96, 192, 384, 768), False),
    "convnextv2_small": ((3, 3, 27, 3), (96, 192, 384, 768), False),
    "convnextv2_base": ((3, 3, 27, 3), (128, 256, 512, 1024), False),
}


class LayerNorm2d(nn.LayerNorm):
    """timm.layers.LayerNorm2d: LN over the channel dim of an NCHW tensor, eps 1e-6."""

    def __init__(self, num_channels: int, eps: float = 1e-6):
        super().__init__(num_channels, eps=eps)

    def forward(self, x: Tensor) -> Tensor:
        x = x.permute(0, 2, 3, 1)
        x = F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
        return x.permute(0, 3, 1, 2)


class GlobalResponseNorm(nn.Module):
    """timm.layers.GlobalResponseNorm."""

    def __init__(self, dim: int, eps: float = 1e-6, channels_last: bool = True):
        super().__init__()
        self.eps = eps
        if channels_last:
            self.spatial_dim, self.channel_dim, self.wb_shape = (1, 2), -1, (1, 1, 1, -1)
        else:
            self.spatial_dim, self.channel_dim, self.wb_shape = (2, 3), 1, (1, -1, 1, 1)
        self.weight = nn.Parameter(torch.zeros(dim))
        self.bias = nn.Parameter(torch.zeros(dim))

    def forward(self, x: Tensor) -> Tensor:
        x_g = x.norm(p=2, dim=self.spatial_dim, keepdim=True)
        x_n = x_g / (x_g.mean(dim=self.channel_dim, keepdim=True) + self.eps)
        return x + torch.addcmul(self.bias.view(self.wb_shape), self.weight.view(self.wb_shape), x * x_n)


class GlobalResponseNormMlp(nn.Module):
    """timm.layers.GlobalResponseNormMlp: fc1 → GELU(erf) → GRN → fc2."""

    def __init__(self, in_features: int, hidden_features: int, out_features: int, use_conv: bool):
        super().__init__()
        # registration order as in timm: fc1, act, drop1, grn, fc2, drop2
        self.fc1 = nn.Conv2d(in_features, hidden_features, 1) if use_conv else nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.grn = GlobalResponseNorm(hidden_features, channels_last=not use_conv)
        self.fc2 = nn.Conv2d(hidden_features, out_features, 1) if use_conv else nn.Linear(hidden_features, out_features)

    def forward(self, x: Tensor) -> Tensor:
        return self.fc2(self.grn(self.act(self.fc1(x))))


class DropPath(nn.Module):
    """timm.layers.DropPath (stochastic depth, scale_by_keep=True): in training mode the whole branch of a sample is
    dropped with probability ``drop_prob`` and the survivors are divided by the keep probability.  ``inject`` ([B] scales)
    replaces the Bernoulli draw (parity tests); the scales actually applied are kept in ``last``."""

    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = drop_prob
        self.inject = None
        self.last = None

    def forward(self, x: Tensor) -> Tensor:
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        if self.inject is not None:
            m = self.inject.to(x.dtype).reshape(shape)
        else:
            m = x.new_empty(shape).bernoulli_(keep)
            if keep > 0.0:
                m.div_(keep)
        self.last = m.reshape(-1).clone()
        return x * m


class ConvNeXtBlock(nn.Module):
    """timm ConvNeXtBlock with use_grn=True, ls_init_value=None."""

    def __init__(self, dim: int, conv_mlp: bool, kernel_size: int = 7, mlp_ratio: int = 4, drop_path: float = 0.0):
        super().__init__()
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.use_conv_mlp = conv_mlp
        self.conv_dw = nn.Conv2d(dim, dim, kernel_size, padding=kernel_size // 2, groups=dim)
        self.norm = LayerNorm2d(dim) if conv_mlp else nn.LayerNorm(dim, eps=1e-6)
        self.mlp = GlobalResponseNormMlp(dim, mlp_ratio * dim, dim, use_conv=conv_mlp)

    def forward(self, x: Tensor) -> Tensor:
        shortcut = x
        x = self.conv_dw(x)
        if self.use_conv_mlp:
            x = self.mlp(self.norm(x))
        else:
            x = x.permute(0, 2, 3, 1)
            x = self.mlp(self.norm(x))
            x = x.permute(0, 3, 1, 2)
        return self.drop_path(x) + shortcut


class ConvNeXtStage(nn.Module):
    """timm ConvNeXtStage: optional (LayerNorm2d → conv k=stride) downsample, then blocks."""

    def __init__(self, in_chs: int, out_chs: int, stride: int, depth: int, conv_mlp: bool, drop_path_rates=None):
        super().__init__()
        drop_path_rates = drop_path_rates or [0.0] * depth
        if in_chs != out_chs or stride > 1:
            ks = 2 if stride > 1 else 1
            self.downsample = nn.Sequential(LayerNorm2d(in_chs), nn.Conv2d(in_chs, out_chs, ks, stride=stride))
        else:
            self.downsample = nn.Identity()
        self.blocks = nn.Sequential(*[ConvNeXtBlock(out_chs, conv_mlp, drop_path=drop_path_rates[i]) for i in range(depth)])

    def forward(self, x: Tensor) -> Tensor:
        return self.blocks(self.downsample(x))


class FeatureInfo:
    def __init__(self, channels: Sequence[int]):
        self._channels = list(channels)

    def channels(self) -> list[int]:
        return list(self._channels)


class ConvNeXtFeatures(nn.Module):
    """What ``timm.create_model(name, features_only=True)`` yields (FeatureListNet):
    flattened children ``stem_0, stem_1, stages_0..3``; returns the 4 stage outputs."""

    def __init__(self, backbone: str, in_chans: int = 3, drop_path_rate: float = 0.0):
        super().__init__()
        depths, dims, conv_mlp = CONVNEXTV2_CFGS[backbone]
        # timm ConvNeXt: dp_rates = [x.tolist() for x in torch.linspace(0, drop_path_rate, sum(depths)).split(depths)]
        dp = [r.tolist() for r in torch.linspace(0, drop_path_rate, sum(depths)).split(list(depths))]
        self.feature_info = FeatureInfo(dims)
        self.stem_0 = nn.Conv2d(in_chans, dims[0], 4, stride=4)
        self.stem_1 = LayerNorm2d(dims[0])
        prev = dims[0]
        for i, (d, c) in enumerate(zip(depths, dims)):
            setattr(self, f"stages_{i}", ConvNeXtStage(prev, c, 2 if i > 0 else 1, d, conv_mlp, dp[i]))
            prev = c
        self.num_stages = len(depths)

    def forward(self, x: Tensor) -> list[Tensor]:
        x = self.stem_1(self.stem_0(x))
        feats = []
        for i in range(self.num_stages):
            x = getattr(self, f"stages_{i}")(x)
            feats.append(x)
        return feats


def timm_init_weights(module: nn.Module) -> None:
    """timm.models.convnext._init_weights: trunc_normal(.02) on conv/linear weights, zero bias."""
    if isinstance(module, (nn.Conv2d, nn.Linear)):
        nn.init.trunc_normal_(module.weight, std=0.02)
        if module.bias is not None:
            nn.init.zeros_(module.bias)


def icnr_init(conv: nn.Module, upsample_factor: int, upsample_dims: int, init=nn.init.kaiming_normal_) -> None:
    """components/blocks.py:14-51."""
    out_channels, in_channels, *dims = conv.weight.shape
    scale_factor = upsample_factor**upsample_dims
    oc2 = int(out_channels / scale_factor)
    kernel = init(torch.zeros([oc2, in_channels] + dims))
    kernel = kernel.transpose(0, 1).reshape(oc2, in_channels, -1).repeat(1, 1, scale_factor)
    kernel = kernel.reshape([in_channels, out_channels] + dims).transpose(0, 1)
    conv.weight.data.copy_(kernel)


class UNeXt2Stem(nn.Module):
    """components/stems.py:8-50."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size, in_stack_depth: int):
        super().__init__()
        if in_stack_depth < kernel_size[0]:
            raise ValueError(f"in_stack_depth ({in_stack_depth}) must be >= kernel_size[0] ({kernel_size[0]})")
        ratio = in_stack_depth // kernel_size[0]
        if out_channels % ratio != 0:
            raise ValueError(
                f"out_channels ({out_channels}) must be divisible by in_stack_depth // kernel_size[0] ({ratio})"
            )
        self.conv = nn.Conv3d(in_channels, out_channels // ratio, kernel_size, stride=kernel_size)

    def forward(self, x: Tensor) -> Tensor:
        x = self.conv(x)
        b, c, d, h, w = x.shape
        return x.reshape(b, c * d, h, w)


class PixelShuffleUp(nn.Module):
    """MONAI ``SubpixelUpsample`` as ``UpSample(mode="pixelshuffle", pre_conv=None | "default"[, apply_pad_pool])`` builds it
    (MONAI 1.5.2 is not in this image: restated from its published source).  ``pre_conv=None``: parameter free.
    ``pre_conv="default"``: ``conv_block = Conv2d(in_channels, out_channels * scale², 3, stride 1, padding 1, bias)`` with
    ICNR-initialised weights in front of the shuffle — what ``UNeXt2(decoder_upsample_pre_conv=True)`` selects
    (components/blocks.py:138-146; there ``out_channels * scale² == in_channels``)."""

    def __init__(self, scale: int, pad_pool: bool, in_channels: int | None = None, pre_conv: bool = False):
        super().__init__()
        self.scale, self.pad_pool = scale, pad_pool
        if pre_conv:
            self.conv_block = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1, bias=True)
            icnr_init(self.conv_block, scale, upsample_dims=2)
        else:
            self.conv_block = nn.Identity()

    def forward(self, x: Tensor) -> Tensor:
        x = F.pixel_shuffle(self.conv_block(x), self.scale)
        if self.pad_pool:
            s = self.scale
            x = F.avg_pool2d(F.pad(x, (s - 1, 0, s - 1, 0)), kernel_size=s, stride=1)
        return x


class UNeXt2UpStage(nn.Module):
    """components/blocks.py:77-172 (pixelshuffle mode).  MONAI's ``UpSample`` is an ``nn.Sequential`` whose pixel-shuffle
    child is registered as ``pixelshuffle`` — the pre-convolution's keys are ``upsample.pixelshuffle.conv_block.*``."""

    def __init__(self, in_channels: int, skip_channels: int, out_channels: int, scale_factor: int, conv_blocks: int,
                 pre_conv: bool = False):
        super().__init__()
        mid = in_channels // scale_factor**2
        self.upsample = nn.Sequential()
        self.upsample.add_module("pixelshuffle", PixelShuffleUp(scale_factor, False, in_channels, pre_conv))
        self.conv = ConvNeXtStage(mid + skip_channels, out_channels, 1, conv_blocks, conv_mlp=True)
        self.conv.apply(timm_init_weights)
        if not pre_conv:  # blocks.py:147: conv_weight_init_factor = None if upsample_pre_conv else scale_factor
            icnr_init(self.conv.blocks[-1].mlp.fc2, scale_factor, upsample_dims=2)

    def forward(self, inp: Tensor, skip: Tensor) -> Tensor:
        return self.conv(torch.cat([self.upsample(inp), skip], dim=1))


class UNeXt2Decoder(nn.Module):
    """components/blocks.py:175-243."""

    def __init__(self, num_channels: list[int], conv_blocks: int, strides: list[int], pre_conv: bool = False):
        super().__init__()
        self.decoder_stages = nn.ModuleList(
            UNeXt2UpStage(num_channels[i], num_channels[i] // 2, num_channels[i + 1], strides[i], conv_blocks, pre_conv)
            for i in range(len(num_channels) - 1)
        )

    def forward(self, features: Sequence[Tensor]) -> Tensor:
        feat = features[0]
        skips = list(features[1:]) + [None]
        for skip, stage in zip(skips, self.decoder_stages):
            feat = stage(feat, skip)
        return feat


class _ADN(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.N = nn.InstanceNorm3d(channels, eps=1e-5, affine=False)
        self.A = nn.PReLU(num_parameters=1, init=0.25)

    def forward(self, x: Tensor) -> Tensor:
        return self.A(self.N(x))


class _MonaiConvolution(nn.Module):
    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, 3, padding=(0, 1, 1))
        self.adn = _ADN(cout)

    def forward(self, x: Tensor) -> Tensor:
        return self.adn(self.conv(x))


class PixelToVoxelHead(nn.Module):
    """components/heads.py:594-641."""

    def __init__(self, in_channels: int, out_channels: int, out_stack_depth: int, expansion_ratio: int, pool: bool):
        super().__init__()
        self.upsample = PixelShuffleUp(2, pad_pool=pool)
        mid = out_channels * expansion_ratio * 4
        self.conv = nn.Sequential(
            _MonaiConvolution(in_channels // 4 // (out_stack_depth + 2), mid),
            nn.Conv3d(mid, out_channels * 4, 1),
        )
        # monai.networks.utils.normal_init(std=0.02): normal on conv weights, zero bias
        nn.init.normal_(self.conv[0].conv.weight, 0.0, 0.02)
        nn.init.zeros_(self.conv[0].conv.bias)
        icnr_init(self.conv[-1], 2, upsample_dims=2)
        self.out = nn.PixelShuffle(2)
        self.out_stack_depth = out_stack_depth

    def forward(self, x: Tensor) -> Tensor:
        x = self.upsample(x)
        d = self.out_stack_depth + 2
        b, c, h, w = x.shape
        x = x.reshape((b, c // d, d, h, w))
        x = self.conv(x)
        x = x.transpose(1, 2)
        x = self.out(x)
        return x.transpose(1, 2)


class UNeXt2(nn.Module):
    """unet/unext2.py:13-82 with reference state-dict key names."""

    def __init__(
        self,
        in_channels: int = 1,
        out_channels: int = 1,
        in_stack_depth: int = 5,
        out_stack_depth: int | None = None,
        backbone: str = "convnextv2_tiny",
        pretrained: bool = False,
        stem_kernel_size=(5, 4, 4),
        decoder_mode: str = "pixelshuffle",
        decoder_conv_blocks: int = 2,
        decoder_norm_layer: str = "instance",
        decoder_upsample_pre_conv: bool = False,
        head_pool: bool = False,
        head_expansion_ratio: int = 4,
        drop_path_rate: float = 0.0,
    ):
        super().__init__()
        if in_stack_depth % stem_kernel_size[0] != 0:
            raise ValueError(
                f"Input stack depth {in_stack_depth} is not divisible by stem kernel depth {stem_kernel_size[0]}."
            )
        if decoder_mode != "pixelshuffle" or pretrained:
            raise NotImplementedError("oracle covers the pixelshuffle path only")
        if out_stack_depth is None:
            out_stack_depth = in_stack_depth
        enc = ConvNeXtFeatures(backbone, drop_path_rate=drop_path_rate)
        enc.apply(timm_init_weights)
        num_channels = enc.feature_info.channels()
        enc.stem_0 = nn.Identity()
        self.encoder_stages = enc
        self.stem = UNeXt2Stem(in_channels, num_channels[0], stem_kernel_size, in_stack_depth)
        dec = num_channels
        dec.reverse()
        dec[-1] = (out_stack_depth + 2) * out_channels * 2**2 * head_expansion_ratio
        self.decoder = UNeXt2Decoder(dec, decoder_conv_blocks, [2] * (len(num_channels) - 1) + [stem_kernel_size[-1]],
                                     pre_conv=bool(decoder_upsample_pre_conv))
        self.head = PixelToVoxelHead(dec[-1], out_channels, out_stack_depth, head_expansion_ratio, pool=head_pool)
        self.out_stack_depth = out_stack_depth

    @property
    def num_blocks(self) -> int:
        return 6

    def forward(self, x: Tensor) -> Tensor:
        x = self.stem(x)
        feats = self.encoder_stages(x)
        feats.reverse()
        return self.head(self.decoder(feats))


def randomize_(model: nn.Module, seed: int = 0, grn_std: float = 0.1, bias_std: float = 0.05) -> nn.Module:
    """Seeded non-degenerate parameters for parity tests: GRN γ/β and biases are zero at
    reference init, which would leave GRN / bias paths unexercised."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if ".grn." in name:
                p.copy_(torch.randn(p.shape, generator=g) * grn_std)
            elif name.endswith(".bias"):
                p.copy_(torch.randn(p.shape, generator=g) * bias_std)
            elif name.endswith("norm.weight") or ".downsample.0.weight" in name or name.endswith("stem_1.weight"):
                p.copy_(1.0 + torch.randn(p.shape, generator=g) * 0.1)
            elif name.endswith("adn.A.weight"):
                p.fill_(0.25)
            else:
                fan_in = p[0].numel() if p.ndim > 1 else p.numel()
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / math.sqrt(fan_in)))
    return model
