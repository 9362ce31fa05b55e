"""Minimal trainer for ``viscy_amd.VSUNet`` + ``viscy_amd.data.HCSDataModule`` reproducing the Lightning
automatic-optimisation semantics the reference relies on: per batch
``zero_grad → autocast(bf16){training_step} → backward → optimizer.step → scheduler step``;
``on_after_batch_transfer`` runs on the device batch; ``fast_dev_run`` = 1 train + 1 val batch;
validation logs the mean of per-dataloader mean losses as ``loss/validate``; DDP = one process per GPU,
gradient all-reduce through ``viscy_amd.parallel.FlatDataParallel`` (RCCL), index sharding with
``DistributedSampler`` semantics, ``sync_dist`` → mean all-reduce of logged scalars.

The training step itself is the one ``bench.py`` measures: for the flat-engine models with the stock ``VSUNet.training_step``
and static batch shapes, ``fit`` drives ``viscy_amd.step.TrainStep`` — the engine's forward / hand-written backward called
directly (no autograd graph), the whole step captured as hipGraph segments that end where a gradient bucket completes, each
bucket's RCCL all-reduce issued between two replays, AdamW as its own graph.  Anything else (a list of batches from a
``CombinedLoader``, a ``fg_mask`` batch, a subclass with its own ``training_step``, a new batch shape after two captured
ones, the CPU "2D" plumbing model) takes the eager path above; both paths advance the same optimiser / schedule state.
"""

from __future__ import annotations

import torch
import torch.distributed as dist

from .parallel import FlatDataParallel


class Trainer:
    def __init__(self, max_epochs: int = 1, fast_dev_run: bool = False, precision: str = "bf16-mixed",
                 accelerator: str = "gpu", seed: int | None = 42, limit_train_batches: int | None = None, callbacks=None,
                 default_root_dir=None, return_predictions: bool = True, graph_step: bool = True):
        self.max_epochs, self.fast_dev_run, self.precision = max_epochs, fast_dev_run, precision
        self.default_root_dir, self.return_predictions = default_root_dir, return_predictions
        self.callbacks = list(callbacks or [])
        self.datamodule = None
        self.limit_train_batches = limit_train_batches
        self.finished = False
        self.global_step = 0
        self.graph_step = graph_step   # False: always the eager zero_grad / training_step / backward / step loop
        self.graph_steps = 0           # optimisation steps taken through the captured TrainStep (tests, logs)
        self._train_steps: dict = {}   # batch signature -> TrainStep
        if seed is not None:
            torch.manual_seed(seed)
        self.device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")

    def _to_device(self, batch):
        if isinstance(batch, dict):
            return {k: self._to_device(v) for k, v in batch.items()}
        if isinstance(batch, (list, tuple)):
            return type(batch)(self._to_device(v) for v in batch)
        if torch.is_tensor(batch):
            return batch.to(self.device, non_blocking=True)
        return batch

    _MAX_CAPTURED_SHAPES = 2  # each captured step owns a memory pool the size of the step's activations

    GRAPH_BELOW_PIXELS = 256 * 256 * 256  # per-rank batch pixels (B x Y x X) below which the captured step is replayed as a hipGraph

    def _graphed_step(self, module, opt, ddp, batch):
        """the captured ``TrainStep`` serving this batch, or None (eager path): see the module docstring"""
        from .vsunet import VSUNet

        if not (self.graph_step and self.device.type == "cuda" and getattr(module, "_native", False)):
            return None
        if type(module).training_step is not VSUNet.training_step or type(module)._compute_loss is not VSUNet._compute_loss:
            return None
        if "training_step" in vars(module) or "_compute_loss" in vars(module):  # instance-level overrides count too
            return None
        if not isinstance(batch, dict) or "fg_mask" in batch:
            return None
        src, tgt = batch.get("source"), batch.get("target")
        if not (torch.is_tensor(src) and torch.is_tensor(tgt) and src.is_cuda and tgt.is_cuda):
            return None
        key = (tuple(src.shape), tuple(tgt.shape), src.dtype, tgt.dtype)
        step = self._train_steps.get(key)
        if step is None:
            if len(self._train_steps) >= self._MAX_CAPTURED_SHAPES:
                # every captured shape holds a private graph memory pool of about the step's activation footprint: release the
                # least recently used one instead of growing (a second shape at a large batch could run out of memory where the
                # eager loop would not) — but only for a shape that COMES BACK: a first sighting runs eagerly, so that
                # three or more alternating shapes do not pay two warm-up steps, a state restore and a capture per batch
                # ... and with a per-shape sighting count that is NOT reset by a capture (with A, B, C, A, B, C ... a set
                # that forgot a shape at its capture re-captured on every other cycle): a shape takes a captured slot only when it
                # has been seen more often than the least recently used captured shape since THAT one was captured
                hits = self.__dict__.setdefault("_shape_hits", {})
                hits[key] = hits.get(key, 0) + 1
                old_key = next(iter(self._train_steps))
                if hits[key] < 2 or hits[key] <= hits.get(old_key, 0):
                    return None
                hits[key] = hits[old_key] = 0
                del self._train_steps[old_key]
                import gc
                import logging

                gc.collect()  # a TrainStep with reference cycles would keep its graph pool alive past empty_cache()
                torch.cuda.empty_cache()
                logging.getLogger("viscy_amd").info("Trainer.fit: released the captured step of batch %s for batch %s", old_key[0], key[0])
            from .step import TrainStep

            # hipGraph replay or eager launches of the SAME direct engine step: measured on the bench workload, the replay pays
            # below ~128 patches of 256 x 256 per rank (launch gaps were 23 % of the step at B = 32) and costs 0.5 - 0.7 ms per step
            # at B = 512, where every launch is long (profiles/r03_fit_throughput.json)
            voxels = src.shape[0] * src.shape[-1] * src.shape[-2]
            use_graph = voxels < self.GRAPH_BELOW_PIXELS
            step = self._train_steps[key] = TrainStep(module.model, module.loss_function, opt, ddp, use_graph=use_graph)
            import logging

            logging.getLogger("viscy_amd").info("Trainer.fit: direct engine step for batch %s (%s)", tuple(src.shape),
                                                "hipGraph replay" if use_graph else "eager launches")
        else:
            self._train_steps[key] = self._train_steps.pop(key)  # most recently used last
            hits = self.__dict__.setdefault("_shape_hits", {})
            hits[key] = hits.get(key, 0) + 1
        return step

    def fit(self, module, datamodule) -> None:
        module.to(self.device)
        module.trainer = self
        self.datamodule = datamodule
        datamodule.trainer = self
        if not dist.is_initialized() or dist.get_rank() == 0:
            datamodule.prepare_data()  # e.g. mmap staging (prepare_data_per_node: once per node)
        if dist.is_initialized():
            dist.barrier()
        datamodule.setup("fit")
        if hasattr(module, "on_fit_start"):
            module.on_fit_start()
        train_dl = datamodule.train_dataloader()
        steps_per_epoch = 1 if self.fast_dev_run else min(len(train_dl), self.limit_train_batches or len(train_dl))
        opt = module.configure_optimizers(t_total=steps_per_epoch * (1 if self.fast_dev_run else self.max_epochs))
        native = hasattr(module.model, "engine")  # False: the plain-PyTorch "2D" plumbing model (single process, CPU or GPU)
        if dist.is_initialized() and not native:
            raise NotImplementedError("data-parallel training is built for the flat-engine models (UNeXt2 / fcmae) only")
        ddp = FlatDataParallel(module.model.engine(), opt) if dist.is_initialized() else None
        self._train_steps = {}  # captured steps are bound to THIS fit's optimiser / process group
        use_bf16 = self.precision.startswith("bf16") and self.device.type == "cuda"
        from .data.combined import CombinedLoader

        for epoch in range(1 if self.fast_dev_run else self.max_epochs):
            module.train()
            datamodule.training = True
            if hasattr(module, "on_train_epoch_start"):
                module.current_epoch = epoch
                module.on_train_epoch_start()
            if isinstance(train_dl, CombinedLoader):
                train_dl.set_epoch(epoch)
            elif hasattr(train_dl, "sampler") and hasattr(train_dl.sampler, "set_epoch"):
                train_dl.sampler.set_epoch(epoch)
            for i, batch in enumerate(train_dl):
                if i >= steps_per_epoch:
                    break
                di = 0
                if isinstance(train_dl, CombinedLoader):  # yields (batch, batch_idx, dataloader_idx) like Lightning's
                    batch, _, di = batch
                batch = datamodule.on_after_batch_transfer(self._to_device(batch), di)
                step = self._graphed_step(module, opt, ddp, batch) if native else None
                if step is not None:
                    # the benched step: direct engine driver, hipGraph segments, bucketed all-reduce between replays, AdamW
                    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=use_bf16):
                        loss = step(batch["source"], batch["target"])
                    module._log("loss/train", loss.clone(), on_step=True, on_epoch=True, prog_bar=True, logger=True,
                                sync_dist=True, batch_size=batch["source"].shape[0])  # what VSUNet.training_step logs
                    self.graph_steps += 1
                else:
                    opt.zero_grad()
                    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=use_bf16):
                        loss = module.training_step(batch, i)
                    loss.backward()
                    if ddp is not None:
                        ddp.finish()
                    opt.step()
                self.global_step += 1
            module.on_train_epoch_end()
            module.eval()
            datamodule.training = False
            with torch.no_grad():
                val_dl = datamodule.val_dataloader()
                for j, batch in enumerate(val_dl):
                    di = 0
                    if isinstance(val_dl, CombinedLoader):
                        batch, j, di = batch
                    batch = datamodule.on_after_batch_transfer(self._to_device(batch), di)
                    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=use_bf16):
                        module.validation_step(batch, j, di)
                    if self.fast_dev_run:
                        break
            datamodule.training = True
            module.on_validation_epoch_end()
        if self.device.type == "cuda":
            torch.cuda.synchronize()
        self.finished = True
        if self.default_root_dir is not None and (not dist.is_initialized() or dist.get_rank() == 0):
            self.save_checkpoint(module)

    def save_checkpoint(self, module, path=None):
        """a Lightning-layout checkpoint (``{"state_dict": ..., "epoch", "global_step"}``) that the reference's
        ``VSUNet(ckpt_path=...)`` / ``predict --ckpt_path`` — and this build's — load"""
        from pathlib import Path

        path = Path(path) if path is not None else Path(self.default_root_dir) / "checkpoints" / "last.ckpt"
        path.parent.mkdir(parents=True, exist_ok=True)
        sd = {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}
        torch.save({"state_dict": sd, "epoch": self.max_epochs, "global_step": self.global_step}, path)
        return path

    def predict(self, module, datamodule) -> list[torch.Tensor]:
        """Lightning's predict loop as the reference uses it: ``on_predict_start`` hooks, ``predict_step`` per batch,
        prediction-writer callbacks (``write_on_batch_end``) after every batch, ``on_predict_end``."""
        module.to(self.device).eval()
        self.datamodule = datamodule
        datamodule.setup("predict")
        module.on_predict_start()
        for cb in self.callbacks:
            if hasattr(cb, "on_predict_start"):
                cb.on_predict_start(self, module)
        outs = []
        try:
            with torch.no_grad():
                for j, batch in enumerate(datamodule.predict_dataloader()):
                    batch = self._to_device(batch)
                    if hasattr(datamodule, "on_after_batch_transfer"):
                        was = getattr(datamodule, "training", False)
                        datamodule.training = False
                        batch = datamodule.on_after_batch_transfer(batch, 0)
                        datamodule.training = was
                    pred = module.predict_step(batch, j)
                    if self.return_predictions:
                        outs.append(pred)
                    for cb in self.callbacks:
                        if hasattr(cb, "write_on_batch_end") and getattr(cb, "interval", "batch") in ("batch", "batch_and_epoch"):
                            cb.write_on_batch_end(self, module, pred, None, batch, j, 0)
        finally:
            # also on an exception / interrupt: a writer with a running device-side blend still holds predicted slices that the
            # reference would have on disk by now — on_predict_end flushes them and closes the store
            for cb in self.callbacks:
                if hasattr(cb, "on_predict_end"):
                    cb.on_predict_end(self, module)
        return outs
