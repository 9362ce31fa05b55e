"""Dense path of the FCMAE U-Net on MI355X: drop-in for ``viscy_models.unet.fcmae.FullyConvolutionalMAE`` with
``mask_ratio = 0`` (/root/reference/packages/viscy-models/src/viscy_models/unet/fcmae.py:451-560) — the network behind
``cytoland.engine.FcmaeUNet`` fine-tuning / inference and the architecture of the published VSCyto3D checkpoint.

Same constructor keywords, ``forward(x, mask_ratio=0.0)`` (returns ``(y, None)`` when ``pretraining`` is set, like the
reference with no mask), ``out_stack_depth`` / ``num_blocks`` / ``pretraining`` attributes and — so that published
checkpoints load with ``strict=True`` — the reference's ``state_dict()`` keys, order and shapes
(``encoder.stem.{conv3d,conv2d,norm}``, ``encoder.stages.i.{downsample,blocks.j.{dwconv,layernorm,mlp}}``,
``decoder.*``, ``head.*``).

Arithmetic: the dense FCMAE network is the UNeXt2 schedule with a different parameter naming, explicit encoder widths
(``nn.Linear`` MLPs) and — unless ``head_conv`` — the parameter-free ``PixelToVoxelShuffleHead`` (``vsx_voxel_shuffle_*``).
The reference-named tree below shares its ``nn.Parameter`` objects with the engine's own tree (``viscy_amd.unext2._Core``),
so the flat-buffer engine, the fused AdamW and the data-parallel all-reduce work unchanged.  Sparse masked pre-training
(``mask_ratio > 0``: masked patchify, mask generation) is not built.
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


class FullyConvolutionalMAE(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, encoder_blocks: Sequence[int] = (3, 3, 9, 3),
                 dims: Sequence[int] = (96, 192, 384, 768), encoder_drop_path_rate: float = 0.0,
                 stem_kernel_size: Sequence[int] = (5, 4, 4), in_stack_depth: int = 5, decoder_conv_blocks: int = 1,
                 pretraining: bool = True, head_conv: bool = False, head_conv_expansion_ratio: int = 4,
                 head_conv_pool: bool = True) -> None:
        super().__init__()
        if encoder_drop_path_rate:
            raise NotImplementedError("encoder_drop_path_rate > 0 is not built")
        if len(dims) != 4 or len(encoder_blocks) != 4:
            raise NotImplementedError("four encoder stages (as in every published configuration) are built")
        stem_kernel_size = tuple(stem_kernel_size)
        if in_stack_depth % stem_kernel_size[0] != 0:
            raise ValueError(f"Input stack depth {in_stack_depth} is not divisible by stem kernel depth {stem_kernel_size[0]}.")
        core = _FcmaeCore(in_channels, out_channels, encoder_blocks, dims, stem_kernel_size, in_stack_depth, decoder_conv_blocks,
                          head_conv, head_conv_expansion_ratio, head_conv_pool)
        object.__setattr__(self, "_core", core)  # NOT a registered submodule: its parameters appear below, under reference names
        enc, stem = _Holder(), _Holder()
        stem.conv3d = _share(core.stem.conv)
        stem.conv2d = _Conv((dims[0], in_channels, stem_kernel_size[1], stem_kernel_size[2]))  # Z == 1 inputs: kept for the key set
        nn.init.trunc_normal_(stem.conv2d.weight, std=0.02)
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

    def forward(self, x: Tensor, mask_ratio: float = 0.0):
        if mask_ratio > 0.0:
            raise NotImplementedError("viscy_amd builds the dense FCMAE path (mask_ratio = 0); masked pre-training is not built")
        if x.ndim == 5 and x.shape[2] == 1:
            raise NotImplementedError("the 2-D stem branch (Z == 1 inputs) is not built")
        y = self._core(x)
        return (y, None) if self.pretraining else y
