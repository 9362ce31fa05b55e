"""DynaCLR contrastive path on MI355X: drop-ins for

  * ``viscy_models.contrastive.ContrastiveEncoder``  (/root/reference/packages/viscy-models/src/viscy_models/contrastive/encoder.py:52-154)
  * ``viscy_models.contrastive.loss.NTXentLoss`` / ``NTXentHCL``  (.../contrastive/loss.py:20-186)
  * the NT-Xent branch of ``dynaclr.engine.ContrastiveModule``  (/root/reference/applications/dynaclr/src/dynaclr/engine.py:33-347)

The trunk is the same ConvNeXt kernel schedule as the UNeXt2 encoder (``viscy_amd.engine_unext2``: stem patch GEMM, depthwise
7x7, LayerNorm, fc1 / GELU / GRN / fc2 GEMMs, 2x2 downsampling GEMMs); behind it ``vsx_avgpool_rows_*``, the LayerNorm kernel,
two small fp32 GEMMs and ``vsx_bn1d_*`` produce ``(embedding, projection)``; ``vsx_ntxent_*`` is the loss.  Same constructor
keywords and ``state_dict()`` keys as the reference (timm names: ``stem.conv``, ``encoder.stem.1``,
``encoder.stages.i.{downsample.{0,1},blocks.j.{[gamma,]conv_dw,norm,mlp.fc1,[mlp.grn,]mlp.fc2}}``, ``encoder.head.norm``,
``projection.{0,1,3,4}`` incl. the BatchNorm buffers), parameters shared with the flat-buffer engine so the fused AdamW and
the RCCL gradient all-reduce work unchanged.

Built: ``backbone="convnext_tiny"`` (V1 blocks: layer scale ``gamma`` folded into fc2 by ``vsx_layer_scale_fold`` /
``_unfold``, identity GRN) and ``"convnextv2_tiny"`` (GRN blocks); ``resnet50`` raises ``NotImplementedError``;
``pretrained`` must be False (no network here); ``drop_path_rate`` is timm's linear stochastic-depth schedule (training mode).  BatchNorm is per process
under data parallelism, as in the reference's default (no SyncBatchNorm in its recipes' trainer sections).
"""

from __future__ import annotations

import math
from typing import Literal, Sequence

import torch
from torch import Tensor, nn

from . import _lib as L
from .unext2 import _Conv, _Core, _Encoder, _Holder, _LN, _Stem


class _BN(_Holder):
    def __init__(self, c: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _Tail(_Holder):
    def __init__(self, feat: int, embedding_dim: int, projection_dim: int):
        super().__init__()
        self.norm = _LN(feat)
        self.fc0, self.bn1 = _Conv((embedding_dim, feat)), _BN(embedding_dim)
        self.fc3, self.bn4 = _Conv((projection_dim, embedding_dim)), _BN(projection_dim)


class _EmbedCore(_Core):
    def __init__(self, in_channels, in_stack_depth, stem_kernel_size, depths, dims, embedding_dim, projection_dim, v1):
        super().__init__()
        kz = stem_kernel_size[0]
        ratio = (in_stack_depth - kz) // kz + 1
        mismatch = dims[0] - ratio * (dims[0] // ratio)
        if mismatch != 0:  # stems.py:113-119
            raise ValueError(f"Stem needs to output {mismatch} more channels to match the encoder. Adjust the in_stack_depth.")
        if in_stack_depth % kz:
            raise NotImplementedError("in_stack_depth must be a multiple of the stem kernel depth (stride == kernel)")
        self.cfg = dict(in_channels=in_channels, out_channels=0, in_stack_depth=in_stack_depth, out_stack_depth=0,
                        depths=tuple(depths), dims=tuple(dims), conv_mlp=False, stem_kernel=tuple(stem_kernel_size), ratio=ratio,
                        head="embed")
        self.encoder_stages = _Encoder(depths, dims, False, v1=v1)
        self.stem = _Stem(in_channels, dims[0] // ratio, tuple(stem_kernel_size))
        self.tail = _Tail(dims[-1], embedding_dim, projection_dim)
        self.compute_dtype = None
        self.grad_mode = "autograd"
        self._engine = None
        self.reset_parameters()

    def reset_parameters(self) -> None:
        for name, mod in self.named_modules():
            if isinstance(mod, _Conv) and name.startswith("encoder_stages"):  # timm _init_weights
                nn.init.trunc_normal_(mod.weight, std=0.02)
                nn.init.zeros_(mod.bias)
        for conv in (self.stem.conv, self.tail.fc0, self.tail.fc3):  # torch defaults (Conv3d / Linear)
            nn.init.kaiming_uniform_(conv.weight, a=math.sqrt(5))
            bound = 1 / math.sqrt(conv.weight[0].numel())
            nn.init.uniform_(conv.bias, -bound, bound)


class ContrastiveEncoder(nn.Module):
    def __init__(self, backbone: Literal["convnext_tiny", "convnextv2_tiny", "resnet50"], in_channels: int, in_stack_depth: int,
                 stem_kernel_size: Sequence[int] = (5, 4, 4), stem_stride: Sequence[int] = (5, 4, 4), embedding_dim: int = 768,
                 projection_dim: int = 128, drop_path_rate: float = 0.0, pretrained: bool = False,
                 depths: Sequence[int] = (3, 3, 9, 3), dims: Sequence[int] = (96, 192, 384, 768)) -> None:
        """``depths`` / ``dims`` are an extension for tests (the reference takes them from the timm model name)."""
        super().__init__()
        if backbone not in ("convnext_tiny", "convnextv2_tiny"):
            raise NotImplementedError(f"backbone {backbone!r}: viscy_amd builds the convnext_tiny / convnextv2_tiny trunks")
        if pretrained:
            raise NotImplementedError("pretrained timm weights cannot be downloaded here; load a state_dict instead")
        if not 0.0 <= float(drop_path_rate) < 1.0:
            raise ValueError(f"drop_path_rate must be in [0, 1), got {drop_path_rate}")
        if tuple(stem_kernel_size) != tuple(stem_stride):
            raise NotImplementedError("stem_stride must equal stem_kernel_size (patchifying stem)")
        if embedding_dim % 4 or projection_dim % 4:
            raise NotImplementedError("embedding_dim and projection_dim must be multiples of 4")
        self.backbone = backbone
        core = _EmbedCore(in_channels, in_stack_depth, tuple(stem_kernel_size), tuple(depths), tuple(dims), embedding_dim, projection_dim,
                          v1=backbone == "convnext_tiny")
        if drop_path_rate:  # timm ConvNeXt: rates rise linearly over all blocks; active in training mode only
            core.cfg["drop_path"] = torch.linspace(0, float(drop_path_rate), sum(depths)).tolist()
        object.__setattr__(self, "_core", core)  # NOT a registered submodule: its parameters appear below, under reference names
        self.stem = core.stem
        enc = _Holder()
        enc.stem = nn.Sequential(nn.Identity(), core.encoder_stages.stem_1)
        enc.stages = nn.Sequential(*[getattr(core.encoder_stages, f"stages_{i}") for i in range(4)])
        enc.norm_pre = nn.Identity()
        head = _Holder()
        head.norm = core.tail.norm
        enc.head = head
        self.encoder = enc
        self.projection = nn.Sequential(core.tail.fc0, core.tail.bn1, nn.ReLU(inplace=True), core.tail.fc3, core.tail.bn4)

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
        self._core._engine = None
        return super()._apply(fn, *a, **k)

    def train(self, mode: bool = True):
        self._core.train(mode)  # BatchNorm mode is read from the core by the engine
        return super().train(mode)

    def forward(self, x: Tensor) -> tuple[Tensor, Tensor]:
        """(embedding [B, num_features], projection [B, projection_dim]) — encoder.py:138-154"""
        return self._core(x)

    def forward_groups(self, x: Tensor, groups: int) -> tuple[Tensor, Tensor]:
        """``groups`` batches concatenated along dim 0, equal to ``groups`` separate ``forward`` calls (BatchNorm batch
        statistics and running-statistics updates per group, in order) with ONE pass of the trunk: half the launches and
        twice the rows per GEMM for the (anchor, positive) step"""
        return self._core(x, None, groups)


# ------------------------------------------------------------------------------------------------ losses
class _NTXentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, embeddings: Tensor, labels: Tensor, temperature: float, beta: float):
        from . import ops as O

        acc, saved = O.ntxent_fwd(embeddings.contiguous().float(), labels.to(torch.int32).contiguous(), temperature, beta)
        ctx.saved, ctx.acc, ctx.in_dtype = saved, acc, embeddings.dtype
        return acc[0].clone()

    @staticmethod
    def backward(ctx, gout: Tensor):
        from . import ops as O

        return O.ntxent_bwd(ctx.saved, ctx.acc, gout.contiguous().float()).to(ctx.in_dtype), None, None, None


def cosine_anneal(start: float, end: float, epoch: int, warmup_epochs: int) -> float:
    """viscy_models/schedule.py:8-33"""
    if epoch >= warmup_epochs:
        return end
    return end + (start - end) * 0.5 * (1.0 + math.cos(math.pi * epoch / warmup_epochs))


class NTXentLoss(nn.Module):
    """NT-Xent over cosine similarities with an optional temperature schedule (loss.py:20-73): ``loss(embeddings, labels)``,
    every ordered pair of distinct samples with equal labels is a positive pair, all samples with another label are its
    negatives, mean over positive pairs."""

    beta = 0.0

    def __init__(self, temperature: float = 0.07, temperature_schedule: Literal["cosine", "constant"] = "constant",
                 temperature_start: float = 0.1, temperature_warmup_epochs: int = 50, **kwargs):
        super().__init__()
        if kwargs:
            raise NotImplementedError(f"pytorch-metric-learning options {sorted(kwargs)} are not built")
        self.temperature = temperature
        self.temperature_schedule, self.temperature_start = temperature_schedule, temperature_start
        self.temperature_end, self.temperature_warmup_epochs = temperature, temperature_warmup_epochs

    def step(self, epoch: int) -> None:
        if self.temperature_schedule == "cosine":
            self.temperature = cosine_anneal(self.temperature_start, self.temperature_end, epoch, self.temperature_warmup_epochs)

    def forward(self, embeddings: Tensor, labels: Tensor) -> Tensor:
        if not embeddings.is_cuda:
            raise RuntimeError(f"viscy_amd.{type(self).__name__} runs on MI355X HIP kernels only (no CPU / eager fallback)")
        if embeddings.ndim != 2 or labels.shape != embeddings.shape[:1]:
            raise ValueError(f"embeddings must be (N, D) and labels (N,), got {tuple(embeddings.shape)} / {tuple(labels.shape)}")
        L.lib()
        return _NTXentFn.apply(embeddings, labels, float(self.temperature), float(self.beta))


class NTXentHCL(NTXentLoss):
    """hard-negative concentration (loss.py:76-186): each negative's term in the denominator is weighted by
    ``exp(beta * sim)``, weights normalised to sum to the number of negatives; ``beta = 0`` is plain NT-Xent."""

    def __init__(self, temperature: float = 0.07, beta: float = 0.5, **kwargs):
        super().__init__(temperature=temperature, **kwargs)
        self.beta = beta


# ------------------------------------------------------------------------------------------------ engine
class ContrastiveModule(nn.Module):
    """``dynaclr.engine.ContrastiveModule`` for the NT-Xent family (engine.py:33-347): ``training_step`` /
    ``validation_step`` on a ``TripletSample`` (``anchor``, ``positive``), ``predict_step`` -> features / projections,
    ``on_train_epoch_start`` temperature schedule, ``configure_optimizers`` -> fused flat AdamW.  Triplet / cosine-embedding
    losses, auxiliary heads and image / PCA logging are not built."""

    def __init__(self, encoder: ContrastiveEncoder, loss_function: nn.Module | None = None, lr: float = 1e-3,
                 schedule: Literal["WarmupCosine", "Constant"] = "Constant", log_batches_per_epoch: int = 8,
                 log_samples_per_batch: int = 1, example_input_array_shape: Sequence[int] = (1, 2, 15, 256, 256),
                 ckpt_path: str | None = None, freeze_backbone: bool = False, gather_embeddings: bool = False, **unused) -> None:
        super().__init__()
        if freeze_backbone:
            raise NotImplementedError("freeze_backbone is not built (the fused flat-buffer optimiser updates every parameter)")
        self.model = encoder
        self.loss_function = loss_function if loss_function is not None else NTXentLoss()
        if not isinstance(self.loss_function, NTXentLoss):
            raise NotImplementedError(f"{type(self.loss_function).__name__}: viscy_amd builds the NT-Xent family of the DynaCLR losses")
        self.lr, self.schedule = lr, schedule
        self.log_batches_per_epoch, self.log_samples_per_batch = log_batches_per_epoch, log_samples_per_batch
        self.example_input_array = torch.rand(*example_input_array_shape)
        # extension beyond the reference (BASELINE config 5): under torch.distributed the projections of all ranks are
        # all-gathered so that every anchor sees world * 2B - 2 negatives instead of 2B - 2; False = the reference's behaviour
        self.gather_embeddings = gather_embeddings
        self.paired_forward = True  # False: two separate forwards, literally as the reference does
        self.current_epoch = 0
        self._logging = True
        self.logged: dict[str, list] = {}
        if ckpt_path is not None:
            self.load_state_dict(torch.load(ckpt_path, weights_only=True, map_location="cpu")["state_dict"], strict=False)

    def _log(self, key: str, value) -> None:
        if self._logging:
            self.logged.setdefault(key, []).append(value.detach() if torch.is_tensor(value) else value)

    def make_train_step(self, optimizer, ddp=None, use_graph: bool = True):
        """the whole contrastive step (zero-grad, two forwards, NT-Xent, backward, fused AdamW) as ONE hipGraph replay per batch
        (``viscy_amd.step.TrainStep``); call ``step(anchor, positive) -> loss`` with fixed shapes"""
        from .step import TrainStep

        def loss_fn(anchor, positive):
            was, self._logging = self._logging, False  # the captured tensors must not pile up in the log
            try:
                return self._step({"anchor": anchor, "positive": positive}, "train")
            finally:
                self._logging = was

        return TrainStep(self.model, None, optimizer, ddp=ddp, use_graph=use_graph, loss_fn=loss_fn)

    def on_train_epoch_start(self) -> None:
        if hasattr(self.loss_function, "step"):
            self.loss_function.step(self.current_epoch)
        self._log("hparams/temperature", self.loss_function.temperature)

    def forward(self, x: Tensor) -> tuple[Tensor, Tensor]:
        return self.model(x)

    def _step(self, batch: dict, stage: str) -> Tensor:
        a, p_ = batch["anchor"], batch["positive"]
        if self.paired_forward and a.shape == p_.shape:
            # one trunk pass over [anchor; positive]; BatchNorm statistics stay per call (dynaclr/engine.py:265-266)
            _, proj = self.model.forward_groups(torch.cat((a, p_)), 2)
            anchor_projection, positive_projection = proj[: a.shape[0]], proj[a.shape[0]:]
        else:
            _, anchor_projection = self(a)
            _, positive_projection = self(p_)
        if self.gather_embeddings:
            from .parallel import all_gather_with_local_grad, scale_for_mean_reduction

            anchor_projection = all_gather_with_local_grad(anchor_projection)
            positive_projection = all_gather_with_local_grad(positive_projection)
        indices = torch.arange(0, anchor_projection.size(0), device=anchor_projection.device)
        loss = self.loss_function(torch.cat((anchor_projection, positive_projection)), torch.cat((indices, indices)))
        self._log(f"loss/{stage}", loss)
        if self.gather_embeddings:
            loss = scale_for_mean_reduction(loss)
        return loss

    def training_step(self, batch: dict, batch_idx: int) -> Tensor:
        return self._step(batch, "train")

    def validation_step(self, batch: dict, batch_idx: int, dataloader_idx: int = 0) -> Tensor:
        return self._step(batch, "val")

    def predict_step(self, batch: dict, batch_idx: int, dataloader_idx: int = 0) -> dict:
        features, projections = self.model(batch["anchor"])
        return {"features": features, "projections": projections, "index": batch.get("index")}

    def configure_optimizers(self, t_total: int | None = None):
        from .optim import FlatAdamW

        self.model.grad_mode = "flat"
        return FlatAdamW(self.model.engine(), lr=self.lr, schedule=self.schedule, t_total=t_total or 0)
