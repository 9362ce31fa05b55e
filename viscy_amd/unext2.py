"""UNeXt2 on MI355X: drop-in for ``viscy_models.unet.unext2.UNeXt2``
(/root/reference/packages/viscy-models/src/viscy_models/unet/unext2.py:13-82).

Same constructor keywords, same ``forward((B,C,Z,Y,X)) -> (B,Cout,Zout,Y,X)``, same
``num_blocks`` / ``out_stack_depth`` attributes, same ``ValueError`` on a bad stack depth and —
so that published checkpoints load with ``strict=True`` — the same ``state_dict()`` keys and
parameter shapes (stem.* / encoder_stages.* / decoder.* / head.*).  The ``nn.Module`` tree below
only *holds* parameters; all arithmetic is a fixed sequence of hand-written HIP kernels
(viscy_amd/csrc, C-ABI in include/vsx.h) driven by ``_Engine``:

  * trunk activations are channels-last ``[B*H*W, C]`` matrices (C is the contraction axis of
    88 % of the FLOPs), in bf16 (production) or fp32 (parity mode);
  * forward and backward are explicit kernel schedules (one autograd node for the whole model),
    saved tensors are chosen by hand: per ConvNeXt block only the input x, the normalised x̂,
    rstd, the pre-activation hidden h and the GRN sums survive the forward;
  * the block LayerNorm's affine is folded into fc1 (W1·diag(γ), b1 + W1·β) when the fp32 master
    weights are re-laid-out for the GEMMs each step; its gradient is unfolded afterwards.

There is no eager / CPU fallback: ``forward`` raises unless the input is on a HIP device and
libvsx.so is loadable.
"""

from __future__ import annotations

import math
from typing import Literal

import torch
from torch import Tensor, nn

from . import _lib as L

CONVNEXTV2_CFGS = {  # timm convnextv2_* (depths, dims, conv_mlp)
    "convnextv2_atto": ((2, 2, 6, 2), (40, 80, 160, 320), True),
    "convnextv2_femto": ((2, 2, 6, 2), (48, 96, 192, 384), True),
    "convnextv2_pico": ((2, 2, 6, 2), (64, 128, 256, 512), True),
    "convnextv2_nano": ((2, 2, 8, 2), (80, 160, 320, 640), True),
    "convnextv2_tiny": ((3, 3, 9, 3), (96, 192, 384, 768), False),
    "convnextv2_small": ((3, 3, 27, 3), (96, 192, 384, 768), False),
    "convnextv2_base": ((3, 3, 27, 3), (128, 256, 512, 1024), False),
}


# ------------------------------------------------------------------------------------------------
# parameter holders (names / shapes = reference; forward is never called)
# ------------------------------------------------------------------------------------------------
class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("viscy_amd parameter holder: run the model through UNeXt2.forward (HIP kernels)")


class _LN(_Holder):
    def __init__(self, c: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class _Conv(_Holder):
    def __init__(self, shape: tuple[int, ...]):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.zeros(shape[0]))


class _GRN(_Holder):
    def __init__(self, c: int):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(c))
        self.bias = nn.Parameter(torch.zeros(c))


class _Mlp(_Holder):
    def __init__(self, c: int, conv_mlp: bool):
        super().__init__()
        self.fc1 = _Conv((4 * c, c, 1, 1) if conv_mlp else (4 * c, c))
        self.grn = _GRN(4 * c)
        self.fc2 = _Conv((c, 4 * c, 1, 1) if conv_mlp else (c, 4 * c))


class _Block(_Holder):
    def __init__(self, c: int, conv_mlp: bool):
        super().__init__()
        self.conv_dw = _Conv((c, 1, 7, 7))
        self.norm = _LN(c)
        self.mlp = _Mlp(c, conv_mlp)


class _MlpV1(_Holder):
    def __init__(self, c: int):
        super().__init__()
        self.fc1 = _Conv((4 * c, c))
        self.fc2 = _Conv((c, 4 * c))


class _BlockV1(_Holder):
    """timm ConvNeXtBlock of the V1 family: layer scale ``gamma`` (ls_init_value 1e-6), plain MLP, no GRN"""

    def __init__(self, c: int):
        super().__init__()
        self.gamma = nn.Parameter(1e-6 * torch.ones(c))
        self.conv_dw = _Conv((c, 1, 7, 7))
        self.norm = _LN(c)
        self.mlp = _MlpV1(c)


class _Stage(_Holder):
    def __init__(self, cin: int, cout: int, stride: int, depth: int, conv_mlp: bool, v1: bool = False):
        super().__init__()
        if cin != cout or stride > 1:
            ks = 2 if stride > 1 else 1
            self.downsample = nn.Sequential(_LN(cin), _Conv((cout, cin, ks, ks)))
        else:
            self.downsample = nn.Identity()
        self.blocks = nn.Sequential(*[(_BlockV1(cout) if v1 else _Block(cout, conv_mlp)) for _ in range(depth)])


class _Encoder(_Holder):
    def __init__(self, depths, dims, conv_mlp, v1: bool = False):
        super().__init__()
        self.stem_0 = nn.Identity()
        self.stem_1 = _LN(dims[0])
        prev = dims[0]
        for i, (d, c) in enumerate(zip(depths, dims)):
            setattr(self, f"stages_{i}", _Stage(prev, c, 2 if i > 0 else 1, d, conv_mlp, v1))
            prev = c


class _Stem(_Holder):
    def __init__(self, cin, cout3d, kernel):
        super().__init__()
        self.conv = _Conv((cout3d, cin, *kernel))


class _UpStage(_Holder):
    """``pre_conv``: MONAI SubpixelUpsample's Conv2d(cin, cin, 3, padding 1) in front of the pixel shuffle
    (``decoder_upsample_pre_conv=True``, blocks.py:138-146); MONAI's ``UpSample`` registers the shuffle block as
    ``pixelshuffle``, hence the key ``upsample.pixelshuffle.conv_block.{weight,bias}``."""

    def __init__(self, cin, cskip, cout, depth, pre_conv=False):
        super().__init__()
        if pre_conv:
            self.upsample = _Holder()
            self.upsample.pixelshuffle = _Holder()
            self.upsample.pixelshuffle.conv_block = _Conv((cin, cin, 3, 3))
        else:
            self.upsample = nn.Identity()
        self.conv = _Stage(cin // 4 + cskip, cout, 1, depth, conv_mlp=True)

    @property
    def pre_conv(self):
        return None if isinstance(self.upsample, nn.Identity) else self.upsample.pixelshuffle.conv_block


class _Decoder(_Holder):
    def __init__(self, chans, depth, pre_conv=False):
        super().__init__()
        self.decoder_stages = nn.ModuleList(
            _UpStage(chans[i], chans[i] // 2, chans[i + 1], depth, pre_conv) for i in range(len(chans) - 1)
        )


class _ADN(_Holder):
    def __init__(self):
        super().__init__()
        self.A = nn.PReLU(num_parameters=1, init=0.25)


class _HeadConv0(_Holder):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = _Conv((cout, cin, 3, 3, 3))
        self.adn = _ADN()


class _ShuffleHead(_Holder):
    """PixelToVoxelShuffleHead: parameter free (pixel shuffle + pad-pool + reshape)."""

    def __init__(self):
        super().__init__()
        self.upsample = nn.Identity()


class _Head(_Holder):
    def __init__(self, c3, cmid, co4):
        super().__init__()
        self.upsample = nn.Identity()
        self.conv = nn.Sequential(_HeadConv0(c3, cmid), _Conv((co4, cmid, 1, 1, 1)))
        self.out = nn.Identity()


def _icnr_(weight: Tensor, scale: int) -> None:
    """ICNR initialisation for a 2-D pixel shuffle of factor ``scale`` (what the reference's ``icnr_init`` produces,
    components/blocks.py:14-51): draw ``out / scale²`` Kaiming-normal kernels and give each of them to the ``scale²``
    consecutive output channels that one shuffled output channel is assembled from, so the up-sampling starts as
    nearest-neighbour (no checkerboard).  (The reference reaches the same structure through a transpose / reshape / repeat
    chain on an i.i.d. tensor; the distribution and the group structure are what matter — tests/test_dp_cpu.py.)"""
    sf = scale * scale
    oc, ic, *dims = weight.shape
    base = nn.init.kaiming_normal_(torch.empty([oc // sf, ic] + dims))
    with torch.no_grad():
        weight.copy_(base.repeat_interleave(sf, dim=0))


# ------------------------------------------------------------------------------------------------
class _Core(nn.Module):
    """Shared machinery of the ConvNeXt-V2 U-Net family on this path (UNeXt2, the dense FCMAE U-Net): parameter tree in the
    engine's naming, flat-buffer engine, HIP-only forward."""

    def _build(self, *, in_channels, out_channels, in_stack_depth, out_stack_depth, depths, dims, conv_mlp, stem_kernel_size,
               decoder_conv_blocks, head: str, head_channels_from: int, head_pool: bool, head_expansion_ratio: int,
               decoder_upsample_pre_conv: bool = False) -> None:
        stem_kernel_size = tuple(stem_kernel_size)
        if stem_kernel_size[1] != 4 or stem_kernel_size[2] != 4:
            raise NotImplementedError("stem_kernel_size must be (k, 4, 4)")
        ratio = in_stack_depth // stem_kernel_size[0]
        if dims[0] % ratio != 0:
            raise ValueError(
                f"out_channels ({dims[0]}) must be divisible by in_stack_depth // kernel_size[0] ({ratio})"
            )
        self.cfg = dict(
            in_channels=in_channels, out_channels=out_channels, in_stack_depth=in_stack_depth,
            out_stack_depth=out_stack_depth, depths=tuple(depths), dims=tuple(dims), conv_mlp=conv_mlp,
            stem_kernel=stem_kernel_size, ratio=ratio, decoder_conv_blocks=decoder_conv_blocks,
            head_pool=bool(head_pool), head_expansion_ratio=head_expansion_ratio, head=head,
        )
        self.encoder_stages = _Encoder(depths, dims, conv_mlp)
        self.stem = _Stem(in_channels, dims[0] // ratio, stem_kernel_size)
        dec = list(reversed(dims))
        if head == "conv":  # PixelToVoxelHead: (D + 2) * C * 2^2 * expansion channels feed the 3-D convolution
            dec[-1] = (out_stack_depth + 2) * head_channels_from * 4 * head_expansion_ratio
        else:               # PixelToVoxelShuffleHead: C_out * D * s^2, s = stem XY kernel
            dec[-1] = out_channels * out_stack_depth * stem_kernel_size[-1] ** 2
        self.cfg["decoder_channels"] = dec
        self.cfg["pre_conv"] = bool(decoder_upsample_pre_conv)
        self.decoder = _Decoder(dec, decoder_conv_blocks, bool(decoder_upsample_pre_conv))
        if head == "conv":
            c3 = dec[-1] // 4 // (out_stack_depth + 2)
            cmid = out_channels * head_expansion_ratio * 4
            self.head = _Head(c3, cmid, out_channels * 4)
        else:
            self.head = _ShuffleHead()
        self.out_stack_depth = out_stack_depth
        self.compute_dtype: torch.dtype | None = None
        self.grad_mode = "autograd"  # or "flat": gradients are written straight into the flat buffer
        self._engine = None
        self.reset_parameters()

    # ---- reference-compatible initialisation (timm _init_weights, MONAI normal_init, ICNR, torch Conv3d default)
    def reset_parameters(self) -> None:
        def tn(w):
            nn.init.trunc_normal_(w, std=0.02)

        for name, mod in self.named_modules():
            if isinstance(mod, _Conv) and (name.startswith("encoder_stages") or name.startswith("decoder")):
                tn(mod.weight)
                nn.init.zeros_(mod.bias)
        w = self.stem.conv.weight  # torch Conv3d default: kaiming_uniform(a=sqrt(5))
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        bound = 1 / math.sqrt(w[0].numel())
        nn.init.uniform_(self.stem.conv.bias, -bound, bound)
        for st in self.decoder.decoder_stages:
            pc = st.pre_conv
            if pc is None:
                _icnr_(st.conv.blocks[-1].mlp.fc2.weight, 2)
            else:  # MONAI SubpixelUpsample: torch Conv2d default bias, ICNR weight; the stage keeps timm's init (blocks.py:147)
                _icnr_(pc.weight, 2)
                bound = 1 / math.sqrt(pc.weight[0].numel())
                nn.init.uniform_(pc.bias, -bound, bound)
        if self.cfg["head"] == "conv":
            nn.init.normal_(self.head.conv[0].conv.weight, 0.0, 0.02)
            nn.init.zeros_(self.head.conv[0].conv.bias)
            w = self.head.conv[1].weight
            _icnr_(w, 2)
            bound = 1 / math.sqrt(w[0].numel())
            nn.init.uniform_(self.head.conv[1].bias, -bound, bound)

    # ---- engine plumbing
    def _bf16_ok(self) -> bool:
        """every channels-last row must be a whole number of 16-byte bf16 vectors (8 channels): true for every published
        configuration except the `convnextv2_atto` U-Net, whose first decoder concat is 80 / 4 + 40 = 60 channels wide"""
        dims = list(self.cfg["dims"])
        widths = list(dims)
        dec = self.cfg.get("decoder_channels")
        if dec:
            skips = list(reversed(dims))[1:]
            widths += [dec[i] // 4 + skips[i] for i in range(len(skips))] + list(dec)
        return all(w % 8 == 0 for w in widths)

    def _resolve_dtype(self) -> torch.dtype:
        want = self.compute_dtype
        if want is None:
            want = torch.bfloat16 if torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16 else torch.float32
        if want == torch.bfloat16 and not self._bf16_ok():
            return torch.float32  # e.g. convnextv2_atto: exact fp32 kernels instead of failing on a 60-channel row
        return want

    def engine(self, ops=None):
        from .engine_unext2 import Engine

        dev = self.stem.conv.weight.device
        if self._engine is None or self._engine.device != dev or (ops is not None and self._engine.ops is not ops):
            self._engine = Engine(self, ops)
        return self._engine

    def _apply(self, fn, *a, **k):
        self._engine = None  # parameter storage moves: flat views must be rebuilt
        return super()._apply(fn, *a, **k)

    def forward(self, x: Tensor, masks=None, bn_groups: int = 1) -> Tensor:
        if not x.is_cuda:
            raise RuntimeError(
                f"viscy_amd.{type(self).__name__} runs on MI355X HIP kernels only (no CPU / eager fallback): move the model "
                "and the input to a 'cuda' (ROCm) device"
            )
        L.lib()  # raises loudly when libvsx.so is missing
        from .engine_unext2 import unext2_apply

        return unext2_apply(self, x, masks, bn_groups)


class UNeXt2(_Core):
    """MI355X-native UNeXt2 (see module docstring).  ``compute_dtype``: None → bf16 under
    ``torch.autocast(bfloat16)``, fp32 otherwise; or force ``torch.bfloat16`` / ``torch.float32``."""

    def __init__(
        self,
        in_channels: int = 1,
        out_channels: int = 1,
        in_stack_depth: int = 5,
        out_stack_depth: int = None,
        backbone: str = "convnextv2_tiny",
        pretrained: bool = False,
        stem_kernel_size: tuple[int, int, int] = (5, 4, 4),
        decoder_mode: Literal["deconv", "pixelshuffle"] = "pixelshuffle",
        decoder_conv_blocks: int = 2,
        decoder_norm_layer: str = "instance",
        decoder_upsample_pre_conv: bool = False,
        head_pool: bool = False,
        head_expansion_ratio: int = 4,
        drop_path_rate: float = 0.0,
    ) -> None:
        super().__init__()
        stem_kernel_size = tuple(stem_kernel_size)
        if in_stack_depth % stem_kernel_size[0] != 0:
            raise ValueError(
                f"Input stack depth {in_stack_depth} is not divisible by stem kernel depth {stem_kernel_size[0]}."
            )
        if backbone not in CONVNEXTV2_CFGS:
            raise ValueError(f"backbone {backbone!r} not available; choose from {sorted(CONVNEXTV2_CFGS)}")
        if decoder_mode != "pixelshuffle":
            raise NotImplementedError("only decoder_mode='pixelshuffle' is built (deconv is broken upstream too)")
        if pretrained:
            raise NotImplementedError("pretrained timm weights cannot be downloaded here; load a state_dict instead")
        if not 0.0 <= float(drop_path_rate) < 1.0:
            raise ValueError(f"drop_path_rate must be in [0, 1), got {drop_path_rate}")
        if out_stack_depth is None:
            out_stack_depth = in_stack_depth
        depths, dims, conv_mlp = CONVNEXTV2_CFGS[backbone]
        self._build(in_channels=in_channels, out_channels=out_channels, in_stack_depth=in_stack_depth,
                    out_stack_depth=out_stack_depth, depths=depths, dims=dims, conv_mlp=conv_mlp,
                    stem_kernel_size=stem_kernel_size, decoder_conv_blocks=decoder_conv_blocks, head="conv",
                    head_channels_from=out_channels, head_pool=head_pool, head_expansion_ratio=head_expansion_ratio,
                    decoder_upsample_pre_conv=decoder_upsample_pre_conv)
        if drop_path_rate:
            # timm ConvNeXt: stochastic-depth rates rise linearly over all encoder blocks (torch.linspace(0, rate, sum(depths)));
            # the decoder stages are built without drop path (blocks.py:54-74)
            self.cfg["drop_path"] = torch.linspace(0, float(drop_path_rate), sum(depths)).tolist()

    @property
    def num_blocks(self) -> int:
        """2-times downscaling factor of the smallest feature map (reference unext2.py:71-74)."""
        return 6
