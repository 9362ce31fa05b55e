"""The YAML seam of the reference pipeline: configs written for ``python -m cytoland fit --config x.yml``
(LightningCLI ``class_path`` / ``init_args`` trees with ``base:`` composition) run on the MI355X classes:

    python -m viscy_amd fit     --config finetune.yml [--config override.yml]
    python -m viscy_amd predict --config predict.yml

* ``load_composed_config`` — ``viscy_utils.compose.load_composed_config``
  (/root/reference/packages/viscy-utils/src/viscy_utils/compose.py:46-140): ``base:`` lists resolved recursively relative to
  the file, deep dict merge (lists REPLACE), top-level ``_private`` keys stripped at every level, circular references rejected.
* ``CLASS_MAP`` — the reference ``class_path`` of every class this build provides -> the MI355X class.  Trainer sections
  keep ``max_epochs`` / ``precision`` / ``fast_dev_run`` / ``limit_train_batches`` and the mapped callbacks
  (``HCSPredictionWriter``); Lightning loggers / checkpoint callbacks / strategies have no counterpart here and are
  reported, not instantiated.
Host-side plumbing only.
"""

from __future__ import annotations

import copy
import importlib
import sys
from pathlib import Path

import yaml

CLASS_MAP = {
    "cytoland.engine.VSUNet": "viscy_amd.vsunet.VSUNet",
    "cytoland.engine.FcmaeUNet": "viscy_amd.vsunet.FcmaeUNet",
    "cytoland.engine.MaskedMSELoss": "viscy_amd.losses.MaskedMSELoss",
    "viscy_utils.losses.MixedLoss": "viscy_amd.losses.MixedLoss",
    "viscy_utils.losses.mixed_loss.MixedLoss": "viscy_amd.losses.MixedLoss",
    "viscy_data.hcs.HCSDataModule": "viscy_amd.data.hcs.HCSDataModule",
    "viscy_data.HCSDataModule": "viscy_amd.data.hcs.HCSDataModule",
    "viscy_data.combined.CombinedDataModule": "viscy_amd.data.combined.CombinedDataModule",
    "viscy_data.CombinedDataModule": "viscy_amd.data.combined.CombinedDataModule",
    "viscy_data.combined.ConcatDataModule": "viscy_amd.data.combined.ConcatDataModule",
    "viscy_data.combined.BatchedConcatDataModule": "viscy_amd.data.combined.BatchedConcatDataModule",
    "viscy_data.BatchedConcatDataModule": "viscy_amd.data.combined.BatchedConcatDataModule",
    "viscy_utils.callbacks.prediction_writer.HCSPredictionWriter": "viscy_amd.prediction_writer.HCSPredictionWriter",
    "viscy_utils.callbacks.HCSPredictionWriter": "viscy_amd.prediction_writer.HCSPredictionWriter",
    "dynaclr.engine.ContrastiveModule": "viscy_amd.contrastive.ContrastiveModule",
    "viscy_models.contrastive.ContrastiveEncoder": "viscy_amd.contrastive.ContrastiveEncoder",
    "viscy_models.contrastive.encoder.ContrastiveEncoder": "viscy_amd.contrastive.ContrastiveEncoder",
    "viscy_models.contrastive.loss.NTXentLoss": "viscy_amd.contrastive.NTXentLoss",
    "viscy_models.contrastive.loss.NTXentHCL": "viscy_amd.contrastive.NTXentHCL",
    "viscy_models.unet.UNeXt2": "viscy_amd.unext2.UNeXt2",
    "viscy_models.unet.unext2.UNeXt2": "viscy_amd.unext2.UNeXt2",
    "viscy_models.unet.FullyConvolutionalMAE": "viscy_amd.fcmae.FullyConvolutionalMAE",
    "viscy_models.unet.Unet2d": "viscy_amd.unet2d.Unet2d",  # CPU plumbing model (BASELINE configs[0]); plain PyTorch
    "viscy_models.unet.unet2d.Unet2d": "viscy_amd.unet2d.Unet2d",
    "viscy_models.unet.fcmae.FullyConvolutionalMAE": "viscy_amd.fcmae.FullyConvolutionalMAE",
}
for _t in ("NormalizeSampled", "MinMaxSampled", "RandWeightedCropd", "CenterSpatialCropd", "BatchedCenterSpatialCropd",
           "BatchedRandAffined", "BatchedRandAdjustContrastd", "BatchedRandScaleIntensityd", "BatchedRandGaussianNoised",
           "BatchedRandGaussianSmoothd", "BatchedRandFlipd", "BatchedRandWeightedCropd", "BatchedRandInvertIntensityd",
           "BatchedStackChannelsd"):
    CLASS_MAP[f"viscy_transforms.{_t}"] = f"viscy_amd.transforms.{_t}"


def deep_merge(base: dict, override: dict) -> dict:
    """compose.py:31-43: dicts merge key by key, everything else (lists included) is replaced"""
    result = dict(base)
    for k, v in override.items():
        if k in result and isinstance(result[k], dict) and isinstance(v, dict):
            result[k] = deep_merge(result[k], v)
        else:
            result[k] = v
    return result


def load_composed_config(path, _seen=None, *, resolver=None) -> dict:
    """compose.py:46-140"""
    path = Path(path).resolve()
    _seen = _seen or frozenset()
    if path in _seen:
        raise ValueError(f"Circular base: reference detected: {path}")
    _seen = _seen | {path}
    with open(path) as f:
        cfg = copy.deepcopy(yaml.safe_load(f) or {})
    bases = cfg.pop("base", [])
    if bases is None:
        bases = []
    elif isinstance(bases, str):
        bases = [bases]
    merged: dict = {}
    for rel in bases:
        merged = deep_merge(merged, load_composed_config(path.parent / rel, _seen))
    result = deep_merge(merged, cfg)
    if resolver is not None:
        result = resolver(result)
    return {k: v for k, v in result.items() if not k.startswith("_")}


def _resolve(class_path: str):
    target = CLASS_MAP.get(class_path)
    if target is None:
        if class_path.startswith("viscy_amd."):
            target = class_path
        else:
            raise KeyError(class_path)
    mod, _, name = target.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def instantiate(node, skipped: list | None = None):
    """``{"class_path": ..., "init_args": {...}}`` trees -> objects (recursively); plain values pass through.  Class paths with
    no MI355X counterpart raise ``KeyError`` unless ``skipped`` is given (then they are recorded there and dropped)."""
    if isinstance(node, list):
        out = [instantiate(v, skipped) for v in node]
        return [v for v in out if v is not _SKIP]
    if isinstance(node, dict):
        if "class_path" in node:
            try:
                cls = _resolve(node["class_path"])
            except KeyError:
                if skipped is None:
                    raise KeyError(f"{node['class_path']}: no viscy_amd counterpart (see viscy_amd.config.CLASS_MAP)") from None
                skipped.append(node["class_path"])
                return _SKIP
            kwargs = {k: instantiate(v, skipped) for k, v in (node.get("init_args") or {}).items()}
            return cls(**{k: v for k, v in kwargs.items() if v is not _SKIP})
        return {k: v2 for k, v in node.items() if (v2 := instantiate(v, skipped)) is not _SKIP}
    return node


_SKIP = object()
_TRAINER_KEYS = ("max_epochs", "precision", "fast_dev_run", "limit_train_batches", "default_root_dir")


def build(cfg: dict):
    """(module, datamodule, trainer, skipped class paths) from a composed config"""
    from .trainer import Trainer

    skipped: list[str] = []
    module = instantiate(cfg["model"])
    datamodule = instantiate(cfg["data"])
    tcfg = cfg.get("trainer") or {}
    callbacks = instantiate(tcfg.get("callbacks") or [], skipped)
    for key in ("logger", "strategy"):
        v = tcfg.get(key)
        if isinstance(v, dict) and "class_path" in v:
            skipped.append(v["class_path"])
        elif isinstance(v, str):
            skipped.append(f"trainer.{key}={v}")
    kw = {k: tcfg[k] for k in _TRAINER_KEYS if k in tcfg}
    if isinstance(kw.get("fast_dev_run"), int):
        kw["fast_dev_run"] = bool(kw["fast_dev_run"])
    if "seed_everything" in cfg and cfg["seed_everything"] is not None:
        kw["seed"] = int(cfg["seed_everything"])
    if "return_predictions" in cfg:
        kw["return_predictions"] = bool(cfg["return_predictions"])
    trainer = Trainer(callbacks=callbacks, **kw)
    ckpt = cfg.get("ckpt_path")
    if ckpt:  # LightningCLI's top-level ckpt_path (predict / resume): Lightning-layout checkpoint, weights only here
        import torch

        module.load_state_dict(torch.load(ckpt, weights_only=True, map_location="cpu")["state_dict"])
    return module, datamodule, trainer, skipped


def main(argv=None) -> int:
    import argparse

    ap = argparse.ArgumentParser(prog="python -m viscy_amd", description="run a cytoland / dynaclr YAML recipe on the MI355X classes")
    ap.add_argument("subcommand", choices=["fit", "predict"])
    ap.add_argument("--config", "-c", action="append", required=True, help="YAML file(s); later ones override earlier ones")
    args = ap.parse_args(argv)
    cfg: dict = {}
    for c in args.config:
        cfg = deep_merge(cfg, load_composed_config(c))
    module, datamodule, trainer, skipped = build(cfg)
    for s in skipped:
        print(f"[viscy_amd] not instantiated (no counterpart in this build): {s}", file=sys.stderr)
    if args.subcommand == "fit":
        trainer.fit(module, datamodule)
    else:
        trainer.predict(module, datamodule)
    return 0
