"""``VSUNet`` on MI355X — same surface as ``cytoland.engine.VSUNet``
(/root/reference/applications/cytoland/src/cytoland/engine.py:167-560): constructor keywords,
``forward / training_step / validation_step / predict_step / on_predict_start /
on_validation_epoch_end / configure_optimizers``, logged keys ``loss/train``, ``loss/val/{i}``,
``loss/validate``, ``example_input_array``, divisible padding + centre crop at predict time, 90-degree
rotation TTA and the Z-sliding-window prediction with ``_blend_in`` feathering
(engine.py:432-501,760-805; viscy_utils/callbacks/prediction_writer.py:74-111).

When ``lightning`` is importable the class IS a ``LightningModule`` and drops into
``python -m cytoland fit -c …`` through ``class_path``; here (no Lightning in the image) it derives
from ``nn.Module`` and is driven by ``viscy_amd.trainer.Trainer``, which reproduces the automatic-
optimisation semantics the reference relies on.
"""

from __future__ import annotations

import inspect
from typing import Callable, Literal, Sequence

import torch
from torch import Tensor, nn

from . import _lib as L
from ._lib import check, lib, ptr, stream
from .unext2 import UNeXt2

try:  # pragma: no cover - lightning is not installed in the build image
    from lightning.pytorch import LightningModule as _Base

    _HAVE_LIGHTNING = True
except Exception:  # noqa: BLE001
    _Base = nn.Module
    _HAVE_LIGHTNING = False

from .fcmae import FullyConvolutionalMAE  # noqa: E402
from .unet2d import Unet2d  # noqa: E402

# cytoland.engine._UNET_ARCHITECTURE (engine.py:36-43): "UNeXt2" / "fcmae" are the accelerated path; "2D" is the reference's
# CPU plumbing case (BASELINE configs[0]) as a plain-PyTorch module (viscy_amd.unet2d) with torch.optim.AdamW — no kernels
_UNET_ARCHITECTURE = {"UNeXt2": UNeXt2, "fcmae": FullyConvolutionalMAE, "2D": Unet2d}


class _TorchOptimizer:
    """optimizer + per-step scheduler pair for the plain-PyTorch "2D" model, with the two calls viscy_amd.trainer makes"""

    def __init__(self, params, lr, schedule, warmup_steps, t_total, warmup_multiplier):
        from .optim import warmup_cosine_lambda

        self.opt = torch.optim.AdamW(params, lr=lr)
        if schedule == "WarmupCosine":
            self.sched = torch.optim.lr_scheduler.LambdaLR(
                self.opt, lambda s: warmup_cosine_lambda(s, warmup_steps, t_total, warmup_multiplier))
        else:
            self.sched = torch.optim.lr_scheduler.ConstantLR(self.opt, factor=1, total_iters=1)

    def zero_grad(self):
        self.opt.zero_grad(set_to_none=True)

    def step(self):
        self.opt.step()
        self.sched.step()


def _divisible_pad_amounts(shape_yx: Sequence[int], k: int) -> list[tuple[int, int]]:
    """MONAI DivisiblePad: symmetric zero padding of each dim to the next multiple of k."""
    out = []
    for s in shape_yx:
        tot = (-s) % k
        out.append((tot // 2, tot - tot // 2))
    return out


def _center_crop_to_shape(tensor: Tensor, spatial_shape: Sequence[int]) -> Tensor:
    """Undo the symmetric DivisiblePad: keep the centred window of the trailing ``len(spatial_shape)`` axes (window start =
    floor of half the surplus — the same convention as the reference's helper, engine.py:61-71)."""
    out = tensor
    first = tensor.ndim - len(spatial_shape)
    for k, want in enumerate(spatial_shape):
        axis = first + k
        have = out.shape[axis]
        if have < want:
            raise ValueError(f"Cannot crop dimension {axis} from {have} to {want}")
        if have != want:
            out = out.narrow(axis, (have - want) // 2, want)
    return out


def blend_in(old_stack: Tensor, new_stack: Tensor, z_slice: slice) -> Tensor:
    """prediction_writer._blend_in on the device: linear Z feathering of overlapping windows."""
    if z_slice.start == 0:
        return new_stack
    depth = z_slice.stop - z_slice.start
    samples = min(z_slice.start + 1, depth)
    factors = [float(min(i + 1, samples)) for i in reversed(range(depth))]
    old_c, new_c = old_stack.contiguous().float(), new_stack.contiguous().float()
    if not old_c.is_cuda or (old_c.shape[-1] * old_c.shape[-2]) % 4:
        raise RuntimeError("viscy_amd.blend_in runs on the HIP device only (plane size must be a multiple of 4)")
    fz = torch.tensor(factors, dtype=torch.float32).to(old_c.device)
    out = torch.empty_like(old_c)
    plane = old_c.shape[-1] * old_c.shape[-2]
    check(lib().vsx_blend_in(ptr(old_c), ptr(new_c), ptr(out), ptr(fz), depth, plane, old_c.numel(), stream()), "blend_in")
    return out


class VSUNet(_Base):
    def __init__(
        self,
        architecture: Literal["UNeXt2", "fcmae", "2D"] = "UNeXt2",
        model_config: dict | None = None,
        loss_function: nn.Module | None = None,
        lr: float = 1e-3,
        schedule: Literal["WarmupCosine", "Constant"] = "Constant",
        warmup_steps: int = 3,
        warmup_multiplier: float = 1e-3,
        freeze_encoder: bool = False,
        ckpt_path: str | None = None,
        log_batches_per_epoch: int = 8,
        log_samples_per_batch: int = 1,
        example_input_yx_shape: Sequence[int] = (256, 256),
        test_cellpose_model_path: str | None = None,
        test_cellpose_diameter: float | None = None,
        test_evaluate_cellpose: bool | None = False,
        test_time_augmentations: bool | None = False,
        tta_type: Literal["mean", "median", "product"] = "mean",
    ) -> None:
        super().__init__()
        if _HAVE_LIGHTNING:  # pragma: no cover
            self.save_hyperparameters(ignore=["loss_function", "ckpt_path"])
        model_config = model_config or {}
        net_class = _UNET_ARCHITECTURE.get(architecture)
        if not net_class:
            raise ValueError(f"Architecture {architecture} not in {_UNET_ARCHITECTURE.keys()} (this build accelerates the "
                             "UNeXt2 path only)")
        self.model = net_class(**model_config)
        self._native = hasattr(self.model, "engine")  # False: the plain-PyTorch "2D" plumbing model
        if freeze_encoder:  # engine.py:204-206; only the FCMAE network has `.encoder` (as in the reference)
            self.model.encoder.requires_grad_(False)
        if loss_function is None and not self._native:
            loss_function = nn.MSELoss()  # engine.py:197
        if loss_function is None:
            from .losses import MixedLoss

            loss_function = MixedLoss(l1_alpha=0.0, l2_alpha=1.0, ms_dssim_alpha=0.0)  # == nn.MSELoss() (engine.py:197)
        self.loss_function = loss_function
        self.lr, self.schedule = lr, schedule
        self.warmup_steps, self.warmup_multiplier = warmup_steps, warmup_multiplier
        self.log_batches_per_epoch, self.log_samples_per_batch = log_batches_per_epoch, log_samples_per_batch
        self.training_step_outputs, self.validation_losses, self.validation_step_outputs = [], [], []
        self.example_input_array = torch.rand(1, model_config.get("in_channels") or 1, model_config.get("in_stack_depth") or 5,
                                              *example_input_yx_shape)
        sig = inspect.signature(self.loss_function.forward)
        self._loss_accepts_fg_mask = "fg_mask" in sig.parameters or any(
            p.kind == inspect.Parameter.VAR_KEYWORD for p in sig.parameters.values())
        self.test_time_augmentations, self.tta_type = test_time_augmentations, tta_type
        self._predict_pad_k = None
        self.predict_graph = False  # True: hipGraph-captured forward per window shape in predict (viscy_amd.step.InferStep)
        self._infer_step = None
        self.logged: dict[str, list[float]] = {}
        if ckpt_path is not None:
            self.load_state_dict(torch.load(ckpt_path, weights_only=True, map_location="cpu")["state_dict"])

    # ---- logging shim (LightningModule.log when available)
    def _log(self, key: str, value: Tensor, **kw) -> None:
        if _HAVE_LIGHTNING:  # pragma: no cover
            self.log(key, value, **kw)
        else:
            value = value.detach()
            if kw.get("sync_dist") and torch.distributed.is_available() and torch.distributed.is_initialized() \
                    and torch.distributed.get_world_size() > 1:
                value = value.clone()
                torch.distributed.all_reduce(value, op=torch.distributed.ReduceOp.SUM)  # Lightning sync_dist: mean over ranks
                value = value / torch.distributed.get_world_size()
            self.logged.setdefault(key, []).append(value)

    def forward(self, x: Tensor) -> Tensor:
        return self.model(x)

    def _compute_loss(self, pred: Tensor, target: Tensor, batch: dict) -> Tensor:
        if "fg_mask" in batch:
            if not self._loss_accepts_fg_mask:
                raise TypeError(f"{type(self.loss_function).__name__} does not accept 'fg_mask'. "
                                f"Use SpotlightLoss or remove fg_mask_key from the data config.")
            return self.loss_function(pred, target, fg_mask=batch["fg_mask"])
        return self.loss_function(pred, target)

    def training_step(self, batch, batch_idx: int):
        losses, batch_size = [], 0
        if not isinstance(batch, Sequence):
            batch = [batch]
        for b in batch:
            pred = self.forward(b["source"])
            losses.append(self._compute_loss(pred, b["target"], b))
            batch_size += b["source"].shape[0]
        loss_step = torch.stack(losses).mean()
        self._log("loss/train", loss_step, on_step=True, on_epoch=True, prog_bar=True, logger=True, sync_dist=True,
                  batch_size=batch_size)
        return loss_step

    def validation_step(self, batch, batch_idx: int, dataloader_idx: int = 0):
        pred = self.forward(batch["source"])
        loss = self._compute_loss(pred, batch["target"], batch)
        if dataloader_idx + 1 > len(self.validation_losses):
            self.validation_losses.append([])
        self.validation_losses[dataloader_idx].append(loss.detach())
        self._log(f"loss/val/{dataloader_idx}", loss, sync_dist=True, batch_size=batch["source"].shape[0])

    def test_step(self, batch, batch_idx: int):
        """Test stage of the reference (engine.py:334-372): the centre Z slice of the first channel of prediction and target
        is scored with MAE, MSE, cosine similarity (along X, averaged), Pearson r and R² over all pixels, logged per batch
        as ``test_metrics/<name>`` together with ``position`` / ``time`` / ``slice`` of the first sample.  Not built: the
        torchmetrics SSIM entry and the Cellpose segmentation metrics (third-party models, outside the hot path).  The
        metric arithmetic is plain tensor code on the forward result (no kernel of the path)."""
        target = batch["target"]
        zc = target.shape[-3] // 2
        t = target[:, 0, zc:zc + 1].float()
        p = self.forward(batch["source"])[:, 0, zc:zc + 1].float()
        pf, tf = p.flatten(), t.flatten()
        pc, tc = pf - pf.mean(), tf - tf.mean()
        mets = {
            "test_metrics/MAE": (pf - tf).abs().mean(),
            "test_metrics/MSE": ((pf - tf) ** 2).mean(),
            "test_metrics/cosine": torch.nn.functional.cosine_similarity(p, t, dim=-1).mean(),
            "test_metrics/pearson": (pc * tc).sum() / (pc.norm() * tc.norm()).clamp_min(1e-20),
            "test_metrics/r2": 1 - ((pf - tf) ** 2).sum() / (tc ** 2).sum().clamp_min(1e-20),
        }
        for k, v in mets.items():
            self._log(k, v, on_step=True, on_epoch=True)
        if "index" in batch:
            names, ts, zs = batch["index"]
            where = {"position": float(names[0].split("/")[-2]), "time": float(ts[0]), "slice": float(zs[0])}
            for k, v in where.items():
                self._log(k, torch.tensor(v), on_step=True, on_epoch=False)
        return mets

    def on_validation_epoch_end(self):
        loss_means = [torch.stack(l).mean() for l in self.validation_losses]
        if loss_means:
            self._log("loss/validate", torch.stack(loss_means).mean(), sync_dist=True)
        self.validation_step_outputs.clear()
        self.validation_losses.clear()

    def on_train_epoch_end(self):
        self.training_step_outputs = []

    # ---- predict path
    def on_predict_start(self):
        self._predict_pad_k = 2 ** self.model.num_blocks  # _make_divisible_pad (engine.py:48-53); UNeXt2 does not downsample Z

    def _pad_forward_crop(self, source: Tensor) -> Tensor:
        if self._predict_pad_k is None:
            self.on_predict_start()
        original_shape = source.shape[2:]
        (py0, py1), (px0, px1) = _divisible_pad_amounts(source.shape[-2:], self._predict_pad_k)
        if py0 or py1 or px0 or px1:
            source = torch.nn.functional.pad(source, (px0, px1, py0, py1))
        if self.predict_graph and source.is_cuda and not torch.is_grad_enabled():
            # tiled inference: every window has the same padded shape -> replay one captured forward per window
            from .step import InferStep

            if self._infer_step is None:
                self._infer_step = InferStep(self.model)
            prediction = self._infer_step(source.contiguous()).clone()
        else:
            prediction = self.forward(source.contiguous())
        return _center_crop_to_shape(prediction, original_shape)

    def predict_step(self, batch, batch_idx: int, dataloader_idx: int = 0):
        source = batch["source"]
        if self.test_time_augmentations:
            return self.perform_test_time_augmentations(source)
        return self._pad_forward_crop(source)

    def perform_test_time_augmentations(self, source: Tensor) -> Tensor:
        """4x rot90 TTA with mean / median / product aggregation (engine.py:464-501)."""
        yx = source.shape[-2:]
        preds = []
        for i in range(4):
            aug = torch.rot90(source, k=i, dims=(-2, -1))
            p = self._pad_forward_crop(aug)
            p = torch.rot90(p, k=4 - i, dims=(-2, -1))
            preds.append(_center_crop_to_shape(p, yx))
        st = torch.stack(preds)
        if self.tta_type == "mean":
            return st.mean(dim=0)
        if self.tta_type == "median":
            return st.median(dim=0).values
        if self.tta_type == "product":
            return torch.exp(torch.log(st + 1e-9).sum(dim=0))
        raise ValueError(f"unknown tta_type {self.tta_type}")

    def predict_sliding_windows(self, x: Tensor, out_channel: int = 2, step: int = 1) -> Tensor:
        """Z sliding-window inference with linear feathering (engine.py:760-805)."""
        if x.ndim != 5:
            raise ValueError(f"Expected input with 5 dimensions (B, C, Z, Y, X), got {x.shape}")
        window = getattr(self.model, "out_stack_depth", None)
        if window is None:
            raise ValueError(f"Model {type(self.model).__name__} does not support sliding window prediction "
                             "(missing out_stack_depth attribute).")
        nz = x.shape[2]
        if window > nz:
            raise ValueError(f"in_stack_depth {window} > input depth {nz}")
        result = x.new_zeros((x.shape[0], out_channel, nz, *x.shape[3:]))
        for z0 in range(0, nz - window + 1, step):
            zs = slice(z0, z0 + window)
            pred = self.predict_step({"source": x[:, :, zs].contiguous()}, 0)
            result[:, :, zs] = blend_in(result[:, :, zs], pred, zs)
        return result

    # ---- optimiser (viscy_utils/optimizers.py:10-62)
    def configure_optimizers(self, t_total: int | None = None):
        """Returns the fused flat-buffer AdamW (+ WarmupCosine) used by viscy_amd.trainer; under Lightning
        the standard ``([optimizer], [scheduler])`` pair over ``model.parameters()`` with the same hyper-parameters."""
        if _HAVE_LIGHTNING:  # pragma: no cover
            from .optim import warmup_cosine_lambda

            opt = torch.optim.AdamW(self.model.parameters(), lr=self.lr)
            if self.schedule == "WarmupCosine":
                tt = t_total or self.trainer.estimated_stepping_batches
                sch = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: warmup_cosine_lambda(s, self.warmup_steps, tt, self.warmup_multiplier))
                return [opt], [{"scheduler": sch, "interval": "step"}]
            return [opt], [torch.optim.lr_scheduler.ConstantLR(opt, factor=1, total_iters=1)]
        if not self._native:
            return _TorchOptimizer(self.model.parameters(), self.lr, self.schedule, self.warmup_steps, t_total or 0,
                                   self.warmup_multiplier)
        from .optim import FlatAdamW

        self.model.grad_mode = "flat"
        return FlatAdamW(self.model.engine(), lr=self.lr, schedule=self.schedule, warmup_steps=self.warmup_steps,
                         t_total=t_total or 0, warmup_multiplier=self.warmup_multiplier)


class FcmaeUNet(VSUNet):
    """``cytoland.engine.FcmaeUNet`` (/root/reference/applications/cytoland/src/cytoland/engine.py:808-1058): FCMAE
    self-supervised pre-training (``fit_mask_ratio > 0`` with ``MaskedMSELoss`` and ``model_config["pretraining"] = True``)
    and supervised fine-tuning (``pretraining = False``; ``encoder_only=True`` loads just the ``model.encoder.*`` weights of
    a pre-trained checkpoint) on the MI355X FCMAE network (``viscy_amd.fcmae``).

    Same constructor keywords, hooks and logged keys (``loss/train``, ``loss/val``).  ``freeze_encoder=True`` stops the
    backward after the decoder and restricts the fused AdamW launch to the decoder / head.  Difference: ``on_fit_start`` checks the loss type only — the reference additionally insists on its
    ``CombinedDataModule`` / ``GPUTransformDataModule`` containers (engine.py:870-878), which are outside this build; any data
    module that yields ``Sample`` dicts (or a list of them, merged like ``CombinedLoader`` batches) works."""

    def __init__(self, fit_mask_ratio: float = 0.0, encoder_only: bool = False, **kwargs):
        # encoder_only: the checkpoint is withheld from the base constructor (which would load it whole and strictly) and only
        # its encoder entries are loaded afterwards
        encoder_ckpt = kwargs.pop("ckpt_path", None) if encoder_only else None
        if encoder_only and encoder_ckpt is None:
            raise ValueError("encoder_only=True requires ckpt_path")
        super().__init__(architecture="fcmae", **kwargs)
        self.fit_mask_ratio = fit_mask_ratio
        if encoder_ckpt is not None:
            self._load_encoder_weights(encoder_ckpt)

    def _load_encoder_weights(self, ckpt_path: str) -> None:
        """``encoder_only=True``: of a pre-trained Lightning checkpoint only the ``model.encoder.*`` entries are loaded, strictly
        (reference behaviour: engine.py:855-868)"""
        full = torch.load(ckpt_path, weights_only=True, map_location="cpu")["state_dict"]
        tag = "model.encoder."
        self.model.encoder.load_state_dict({name[len(tag):]: w for name, w in full.items() if name.startswith(tag)}, strict=True)

    def on_fit_start(self):
        from .losses import MaskedMSELoss

        if self.model.pretraining and not isinstance(self.loss_function, MaskedMSELoss):
            raise ValueError(f"MaskedMSELoss is required for FCMAE pre-training, got {type(self.loss_function)}")

    def forward(self, x: Tensor, mask_ratio: float = 0.0):
        return self.model(x, mask_ratio)

    def forward_fit_fcmae(self, batch: dict, return_target: bool = False):
        """engine.py:897-919: reconstruct the hidden patches of ``source`` itself."""
        x = batch["source"]
        pred, mask = self.forward(x, mask_ratio=self.fit_mask_ratio)
        loss = self.loss_function(pred, x, mask)
        target = x * mask.unsqueeze(2) if return_target else None
        return pred, target, loss

    def make_pretrain_step(self, optimizer, ddp=None, use_graph: bool = True):
        """masked pre-training step (device-side mask draw, masked forward, MaskedMSELoss, backward, fused AdamW) as one
        hipGraph replay (``viscy_amd.step.TrainStep``); call ``step(source, source) -> loss`` with fixed shapes"""
        from .step import TrainStep

        def loss_fn(x, _unused):
            pred, mask = self.forward(x, mask_ratio=self.fit_mask_ratio)
            return self.loss_function(pred, x, mask)

        return TrainStep(self.model, None, optimizer, ddp=ddp, use_graph=use_graph, loss_fn=loss_fn)

    def forward_fit_supervised(self, batch: dict):
        """engine.py:921-939"""
        x, target = batch["source"], batch["target"]
        pred = self.forward(x)
        return pred, target, self._compute_loss(pred, target, batch)

    def forward_fit_task(self, batch: dict, batch_idx: int):
        """engine.py:941-964"""
        if self.model.pretraining:
            return self.forward_fit_fcmae(batch, return_target=batch_idx < self.log_batches_per_epoch)
        return self.forward_fit_supervised(batch)

    @staticmethod
    def _merge_batches(batch):
        """A combined loader hands over one ``Sample`` per dataset; they become ONE sample along the batch axis (reference
        behaviour: engine.py:966-1004).  Tensors are concatenated; the ``index`` tuple is merged member by member (tensors
        concatenated, FOV-name lists chained); anything else (e.g. ``norm_meta``) is taken from the first dataset."""
        if not isinstance(batch, list):
            return batch

        def join(items):
            head = items[0]
            if isinstance(head, Tensor):
                return torch.cat(items, dim=0)
            if isinstance(head, list):
                return [e for it in items for e in it]
            return head

        out = {}
        for key, head in batch[0].items():
            items = [b[key] for b in batch if key in b]
            out[key] = tuple(join(list(member)) for member in zip(*items)) if isinstance(head, tuple) else join(items) \
                if isinstance(head, Tensor) else head
        return out

    def training_step(self, batch, batch_idx: int):
        batch = self._merge_batches(batch)
        pred, target, loss = self.forward_fit_task(batch, batch_idx)
        self._log("loss/train", loss, on_step=True, on_epoch=True, prog_bar=True, logger=True, sync_dist=True,
                  batch_size=pred.shape[0])
        return loss

    def validation_step(self, batch, batch_idx: int, dataloader_idx: int = 0):
        pred, target, loss = self.forward_fit_task(batch, batch_idx)
        if dataloader_idx + 1 > len(self.validation_losses):
            self.validation_losses.append([])
        self.validation_losses[dataloader_idx].append(loss.detach())
        self._log("loss/val", loss, sync_dist=True, batch_size=pred.shape[0])
