"""The FCMAE U-Net on MI355X: drop-in for ``viscy_models.unet.fcmae.FullyConvolutionalMAE``
(/root/reference/packages/viscy-models/src/viscy_models/unet/fcmae.py:451-560) — the network behind
``cytoland.engine.FcmaeUNet`` (masked pre-training, fine-tuning, inference) and the architecture of the published VSCyto3D
checkpoint.

Same constructor keywords, ``forward(x, mask_ratio=0.0)`` (returns ``(y, mask)`` when ``pretraining`` is set — ``mask`` is
None without masking, like the reference), ``out_stack_depth`` / ``num_blocks`` / ``pretraining`` attributes and — so that published
checkpoints load with ``strict=True`` — the reference's ``state_dict()`` keys, order and shapes
(``encoder.stem.{conv3d,conv2d,norm}``, ``encoder.stages.i.{downsample,blocks.j.{dwconv,layernorm,mlp}}``,
``decoder.*``, ``head.*``).

Arithmetic: the dense FCMAE network is the UNeXt2 schedule with a different parameter naming, explicit encoder widths
(``nn.Linear`` MLPs) and — unless ``head_conv`` — the parameter-free ``PixelToVoxelShuffleHead`` (``vsx_voxel_shuffle_*``).
The reference-named tree below shares its ``nn.Parameter`` objects with the engine's own tree (``viscy_amd.unext2._Core``),
so the flat-buffer engine, the fused AdamW and the data-parallel all-reduce work unchanged.  Sparse masked pre-training
(``mask_ratio > 0``) runs the same kernel schedule with the kept tokens of every encoder block gathered into compact row
matrices (``vsx_rows_select``; see ``stage_row_maps``).
"""

from __future__ import annotations

import math
from typing import Sequence

import torch
from torch import Tensor, nn

from .unext2 import _Conv, _Core, _Holder


class _FcmaeCore(_Core):
    def __init__(self, in_channels, out_channels, encoder_blocks, dims, stem_kernel_size, in_stack_depth, decoder_conv_blocks,
                 head_conv, head_conv_expansion_ratio, head_conv_pool):
        super().__init__()
        # MaskedAdaptiveProjection.conv2d (fcmae.py:348-353): the stem of Z == 1 inputs.  A parameter of the core, so it lives
        # in the engine's flat buffers and trains like the rest; registered FIRST = next to the 3-D stem at the tail of the
        # (reverse-forward-order) flat buffer, i.e. in the gradient bucket that is reduced last.
        self.stem2d = _Conv((dims[0], in_channels, stem_kernel_size[1], stem_kernel_size[2]))
        nn.init.trunc_normal_(self.stem2d.weight, std=0.02)
        self._build(in_channels=in_channels, out_channels=out_channels, in_stack_depth=in_stack_depth,
                    out_stack_depth=in_stack_depth, depths=tuple(encoder_blocks), dims=tuple(dims), conv_mlp=False,
                    stem_kernel_size=stem_kernel_size, decoder_conv_blocks=decoder_conv_blocks,
                    head="conv" if head_conv else "shuffle", head_channels_from=in_channels, head_pool=head_conv_pool,
                    head_expansion_ratio=head_conv_expansion_ratio)


def _share(src: nn.Module) -> _Holder:
    """a holder whose ``weight`` / ``bias`` ARE the parameters of ``src`` (same objects, other name)"""
    h = _Holder()
    h.weight, h.bias = src.weight, src.bias
    return h


class _BlockView(_Holder):
    def __init__(self, blk):
        super().__init__()
        self.dwconv = _share(blk.conv_dw)
        self.layernorm = _share(blk.norm)
        mlp = _Holder()
        mlp.fc1, mlp.grn, mlp.fc2 = _share(blk.mlp.fc1), _share(blk.mlp.grn), _share(blk.mlp.fc2)
        self.mlp = mlp
        self.drop_path, self.shortcut = nn.Identity(), nn.Identity()


class _StageView(_Holder):
    def __init__(self, st):
        super().__init__()
        if isinstance(st.downsample, nn.Identity):
            self.downsample = nn.Identity()
        else:
            self.downsample = nn.Sequential(_share(st.downsample[0]), _share(st.downsample[1]))
        self.blocks = nn.ModuleList(_BlockView(b) for b in st.blocks)


def generate_mask(target, stride: int, mask_ratio: float, device) -> Tensor:
    """fcmae.py:40-66 — the same draw (``torch.rand(B, n).argsort(1) < int(n * ratio)`` on ``device``): (B,1,H/stride,W/stride)
    bool, True = masked, exactly ``int(n * ratio)`` cells per sample."""
    mh, mw = target[-2] // stride, target[-1] // stride
    n = mh * mw
    k = int(n * mask_ratio)
    return (torch.rand(target[0], n, device=device).argsort(1) < k).reshape(target[0], 1, mh, mw)


def upsample_mask(mask: Tensor, target) -> Tensor:
    """fcmae.py:69-92"""
    if tuple(target[-2:]) != tuple(mask.shape[-2:]):
        if target[-2] % mask.shape[-2] or target[-1] % mask.shape[-1]:
            raise ValueError(f"feature map shape {tuple(target)} must be divisible by mask shape {tuple(mask.shape)}.")
        mask = mask.repeat_interleave(target[-2] // mask.shape[-2], dim=-2).repeat_interleave(target[-1] // mask.shape[-1], dim=-1)
    return mask


def stage_row_maps(unmasked: Tensor, shapes, kept_cells: int):
    """Row maps that drive ``vsx_rows_select`` for each encoder stage: the boolean-index gather / zero-filled scatter of
    ``masked_patchify`` / ``masked_unpatchify`` (fcmae.py:95-141) on channels-last row matrices.

    ``unmasked``: (B,1,mh,mw) bool; ``shapes``: [(h_i, w_i)] per stage; ``kept_cells``: unmasked cells per sample (identical
    for every sample by construction of ``generate_mask``, so every sample keeps the same token count L_i and the compact
    matrix stays rectangular).  Index arithmetic only — static sizes, no host synchronisation (graph-capturable).
    Returns per stage ``(idx [B*L] dense row of each kept token (row-major, as boolean indexing orders them),
    inv [B*h*w] compact row or -1, keep [B*h*w] own row or -1, L)``."""
    B, _, mh, mw = unmasked.shape
    out = []
    for h, w in shapes:
        u = upsample_mask(unmasked, (B, 1, h, w)).reshape(-1)
        n = B * h * w
        L = kept_cells * (h // mh) * (w // mw)
        ar = torch.arange(n, dtype=torch.int32, device=u.device)
        pos = torch.cumsum(u, 0, dtype=torch.int32) - 1
        neg = torch.full_like(ar, -1)
        inv = torch.where(u, pos, neg)
        keep = torch.where(u, ar, neg)
        idx = torch.zeros(B * L + 1, dtype=torch.int32, device=u.device)
        idx.scatter_(0, torch.where(u, pos, torch.full_like(pos, B * L)).long(), ar)  # masked rows land in the dump slot
        out.append((idx[: B * L].contiguous(), inv, keep, L))
    return out


class FullyConvolutionalMAE(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, encoder_blocks: Sequence[int] = (3, 3, 9, 3),
                 dims: Sequence[int] = (96, 192, 384, 768), encoder_drop_path_rate: float = 0.0,
                 stem_kernel_size: Sequence[int] = (5, 4, 4), in_stack_depth: int = 5, decoder_conv_blocks: int = 1,
                 pretraining: bool = True, head_conv: bool = False, head_conv_expansion_ratio: int = 4,
                 head_conv_pool: bool = True) -> None:
        super().__init__()
        if not 0.0 <= float(encoder_drop_path_rate) < 1.0:
            raise ValueError(f"encoder_drop_path_rate must be in [0, 1), got {encoder_drop_path_rate}")
        if len(dims) != 4 or len(encoder_blocks) != 4:
            raise NotImplementedError("four encoder stages (as in every published configuration) are built")
        stem_kernel_size = tuple(stem_kernel_size)
        if in_stack_depth % stem_kernel_size[0] != 0:
            raise ValueError(f"Input stack depth {in_stack_depth} is not divisible by stem kernel depth {stem_kernel_size[0]}.")
        core = _FcmaeCore(in_channels, out_channels, encoder_blocks, dims, stem_kernel_size, in_stack_depth, decoder_conv_blocks,
                          head_conv, head_conv_expansion_ratio, head_conv_pool)
        if encoder_drop_path_rate:  # fcmae.py:404-414: the SAME rate for every encoder block (published recipes: 0.1)
            core.cfg["drop_path"] = [float(encoder_drop_path_rate)] * sum(encoder_blocks)
        object.__setattr__(self, "_core", core)  # NOT a registered submodule: its parameters appear below, under reference names
        enc, stem = _Holder(), _Holder()
        stem.conv3d = _share(core.stem.conv)
        stem.conv2d = _share(core.stem2d)  # Z == 1 inputs (fcmae.py:369-370)
        stem.norm = _share(core.encoder_stages.stem_1)
        enc.stem = stem
        enc.stages = nn.ModuleList(_StageView(getattr(core.encoder_stages, f"stages_{i}")) for i in range(4))
        self.encoder = enc
        self.decoder = core.decoder  # same key names as the reference's UNeXt2Decoder
        self.head = core.head
        self.out_stack_depth = in_stack_depth
        self.num_blocks = len(dims) * int(math.log2(stem_kernel_size[-1]))
        self.pretraining = pretraining
        # fcmae.py:26-37 `_init_weights` over the encoder: trunc-normal Conv3d? no — Conv2d / Linear only; the stem Conv3d keeps
        # torch's default init (already applied by the core)

    # ---- the engine's knobs live on the core
    @property
    def cfg(self):
        return self._core.cfg

    @property
    def compute_dtype(self):
        return self._core.compute_dtype

    @compute_dtype.setter
    def compute_dtype(self, v):
        self._core.compute_dtype = v

    @property
    def grad_mode(self):
        return self._core.grad_mode

    @grad_mode.setter
    def grad_mode(self, v):
        self._core.grad_mode = v

    def engine(self, ops=None):
        return self._core.engine(ops)

    def _apply(self, fn, *a, **k):
        self._core._engine = None  # parameter storage moves: flat views must be rebuilt
        return super()._apply(fn, *a, **k)

    def train(self, mode: bool = True):
        self._core.train(mode)  # stochastic depth is active in training mode only; the engine reads the core's flag
        return super().train(mode)

    @property
    def total_stride(self) -> int:
        """MaskedMultiscaleEncoder.total_stride, fcmae.py:422"""
        return self._core.cfg["stem_kernel"][1] * 2 ** 3

    def forward(self, x: Tensor, mask_ratio: float = 0.0, mask: Tensor | None = None):
        """fcmae.py:541-560.  ``mask`` (B,1,H/stride,W/stride bool, True = masked) is an extension for tests: it injects the
        draw ``generate_mask`` would make; every sample must mask the same number of cells."""
        masks = None
        if mask is None and mask_ratio > 0.0:
            mask = generate_mask(x.shape, self.total_stride, mask_ratio, x.device)
            kept = mask.shape[-2] * mask.shape[-1] - int(mask.shape[-2] * mask.shape[-1] * mask_ratio)
        elif mask is not None:
            per = (~mask).flatten(1).sum(1)
            kept = int(per[0])
            if not bool((per == kept).all()):
                raise ValueError("an injected mask must keep the same number of cells in every sample")
        if mask is not None:
            if kept == 0:
                raise ValueError("mask_ratio masks every cell: nothing left to encode")
            sk = self._core.cfg["stem_kernel"]
            h, w = x.shape[-2] // sk[1], x.shape[-1] // sk[2]
            masks = stage_row_maps(~mask, [(h >> i, w >> i) for i in range(4)], kept)
            mask = upsample_mask(mask, (x.shape[0], 1, x.shape[-2], x.shape[-1]))
        y = self._core(x, masks)
        return (y, mask) if self.pretraining else y
