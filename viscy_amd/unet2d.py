"""The legacy 2-D U-Net (architecture ``"2D"`` of ``cytoland.engine._UNET_ARCHITECTURE``) as a plain-PyTorch CPU plumbing model.

BASELINE.json ``configs[0]`` is the reference's own CPU-runnable case: ``VSUNet("2D")`` on a tiny 1 → 1 channel, 256×256
input, one training step over a synthetic HCS zarr.  It exists so that the data module → ``VSUNet`` hooks → optimizer
plumbing can be exercised without a GPU; it is NOT on the accelerated path (no HIP kernel, no flat engine), takes
``torch.optim.AdamW`` and is never benchmarked.  Same constructor keywords, module names / state-dict keys and arithmetic as
``viscy_models.unet.Unet2d`` (/root/reference/packages/viscy-models/src/viscy_models/unet/unet2d.py:17-228) with its
``ConvBlock2D`` (/root/reference/packages/viscy-models/src/viscy_models/components/conv_block_2d.py:11-388) in the one
configuration ``Unet2d`` instantiates it with (``filter_steps="first"``, ``layer_order="can"``, batch norm inside the body
blocks, none in the terminal block).

Reference quirks kept on purpose (they change numerics or checkpoints):
* ``dropout`` builds ``Dropout2d(int(dropout))`` — ``int(0.2) == 0`` — so the default drop-out is the identity;
* a residual block with more output than input channels zero-pads the skip on the LOW channel side;
* every block owns a 1×1 ``resid_conv`` (state-dict keys) whether or not it is used;
* the input's depth axis is squeezed (``x.squeeze(2)``) and re-inserted on the output.
"""

from __future__ import annotations

from typing import Sequence

import torch
import torch.nn.functional as F
from torch import Tensor, nn


class _ConvStack2d(nn.Module):
    """``num_repeats`` × (conv 'same' → [drop-out] → ReLU → [BatchNorm]); all convolutions after the first keep the output
    width (``filter_steps="first"``); optional residual sum with the block input after the last repeat."""

    def __init__(self, in_filters: int, out_filters: int, *, dropout: float | bool, residual: bool, num_repeats: int,
                 kernel_size, batch_norm: bool = True, last_linear: bool = False):
        super().__init__()
        sizes = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        if len(sizes) != 2:
            raise ValueError("kernel_size length must be 2")
        if any(k % 2 != 1 for k in sizes):
            raise ValueError("Kernel dims must be odd")
        self.in_filters, self.out_filters = in_filters, out_filters
        self.residual, self.num_repeats, self.last_linear = residual, num_repeats, last_linear
        self.batch_norm = batch_norm
        # registration order (norms, convolutions, resid_conv) follows the reference so that state_dict() orders match
        if batch_norm:
            for r in range(num_repeats):
                self.add_module(f"batch_norm_{r}", nn.BatchNorm2d(out_filters))
        for r in range(num_repeats):
            self.add_module(f"Conv2d_{r}", nn.Conv2d(in_filters if r == 0 else out_filters, out_filters, kernel_size=kernel_size,
                                                     padding="same"))
        self.resid_conv = nn.Conv2d(in_filters, out_filters, kernel_size=1, padding=0)
        self.drop_p = float(int(dropout)) if dropout else None

    def forward(self, x: Tensor, validate_input: bool = False) -> Tensor:
        if validate_input and x.shape[1] != self.in_filters:
            raise ValueError(f"expected {self.in_filters} input channels, got {x.shape[1]}")
        skip = x
        for r in range(self.num_repeats):
            x = getattr(self, f"Conv2d_{r}")(x)
            if self.drop_p is not None:
                x = F.dropout2d(x, self.drop_p, self.training)
            if not (self.last_linear and r == self.num_repeats - 1):
                x = F.relu(x)
            if self.batch_norm:
                x = getattr(self, f"batch_norm_{r}")(x)
        if self.residual:
            extra = self.out_filters - self.in_filters
            if extra < 0:
                skip = self.resid_conv(skip)
            elif extra > 0:
                skip = torch.cat([skip.new_zeros(skip.shape[0], extra, *skip.shape[2:]), skip], dim=1)
            x = skip + x
        return x


class Unet2d(nn.Module):
    def __name__(self):  # the reference defines the same method (unet2d.py:18-19)
        return "Unet2d"

    def __init__(self, in_channels: int = 1, out_channels: int = 1, kernel_size=(3, 3), residual: bool = False,
                 dropout: float = 0.2, num_blocks: int = 4, num_block_layers: int = 2, num_filters: Sequence[int] = (),
                 task: str = "seg"):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.residual, self.dropout, self.num_blocks, self.num_block_layers, self.task = (residual, dropout, num_blocks,
                                                                                           num_block_layers, task)
        if len(num_filters):
            if len(num_filters) != num_blocks + 1:
                raise AssertionError("Length of num_filters must be equal to num_blocks + 1 (number of convolutional blocks "
                                     "per path).")
            widths = list(num_filters)
        else:
            widths = [16 << i for i in range(num_blocks + 1)]
        self.num_filters = widths
        body = dict(dropout=dropout, residual=residual, num_repeats=num_block_layers, kernel_size=kernel_size)
        for i in range(num_blocks):
            self.add_module(f"down_samp_{i}", nn.AvgPool2d(kernel_size=2))
        enc_in = [in_channels] + widths[:-1]
        for i in range(num_blocks):
            self.add_module(f"down_conv_block_{i}", _ConvStack2d(enc_in[i], widths[i], **body))
        self.bottom_transition_block = _ConvStack2d(widths[-2], widths[-1], **body)
        for i in range(num_blocks):  # decoder block i sees [upsampled (widths[-1-i]) | skip (widths[-2-i])] channels
            self.add_module(f"up_conv_block_{i}", _ConvStack2d(widths[-1 - i] + widths[-2 - i], widths[-2 - i], **body))
        self.terminal_block = _ConvStack2d(widths[0], out_channels, dropout=dropout, residual=False, num_repeats=1,
                                           kernel_size=kernel_size, batch_norm=False, last_linear=(task == "reg"))

    def forward(self, x: Tensor, validate_input: bool = False) -> Tensor:
        if validate_input:
            if x.shape[-1] != x.shape[-2]:
                raise AssertionError("Input must be square in xy")
            if x.shape[-3] != self.in_channels:
                raise AssertionError(f"Input channels must equal network input channels: {self.in_channels}")
        x = x.squeeze(2)
        skips = []
        for i in range(self.num_blocks):
            x = getattr(self, f"down_conv_block_{i}")(x, validate_input)
            skips.append(x)
            x = getattr(self, f"down_samp_{i}")(x)
        x = self.bottom_transition_block(x)
        for i in range(self.num_blocks):
            x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
            x = getattr(self, f"up_conv_block_{i}")(torch.cat([x, skips.pop()], dim=1), validate_input)
        return self.terminal_block(x).unsqueeze(2)
