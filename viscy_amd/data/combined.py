"""``CombinedDataModule`` — several data modules behind one (/root/reference/packages/viscy-data/src/viscy_data/combined.py:22-120),
the container of the multi-dataset fine-tuning recipes (``train_mode: MAX_SIZE_CYCLE``, ``val_mode: SEQUENTIAL``:
applications/cytoland/examples/configs/vscyto3d/finetune_a549_infected.yml), and the part of Lightning's
``CombinedLoader`` it relies on (lightning 2.6 ``utilities/combined_loader.py``; not installed here, semantics restated):

  min_size        stop with the shortest loader; every step yields the list of one batch per loader
  max_size_cycle  run as long as the longest loader, restarting the shorter ones; list of batches per step
  max_size        run as long as the longest loader; exhausted loaders contribute ``None``
  sequential      one loader after the other; every step yields ONE batch and the index of its loader

Iterating a ``CombinedLoader`` yields ``(batch, batch_idx, dataloader_idx)`` (``dataloader_idx`` is 0 outside sequential mode),
as Lightning's does.  Host-side plumbing only.
"""

from __future__ import annotations

from enum import Enum
from typing import Sequence

import torch

try:  # pragma: no cover
    from lightning.pytorch import LightningDataModule as _DMBase
except Exception:  # lightning is not installed in this image
    _DMBase = object


class CombineMode(Enum):
    MIN_SIZE = "min_size"
    MAX_SIZE_CYCLE = "max_size_cycle"
    MAX_SIZE = "max_size"
    SEQUENTIAL = "sequential"


class CombinedLoader:
    def __init__(self, iterables: Sequence, mode: str = "min_size"):
        self.iterables, self.mode = list(iterables), CombineMode(mode).value

    def __len__(self) -> int:
        lens = [len(it) for it in self.iterables]
        if self.mode == "min_size":
            return min(lens)
        if self.mode == "sequential":
            return sum(lens)
        return max(lens)

    def set_epoch(self, epoch: int) -> None:
        for it in self.iterables:
            s = getattr(it, "sampler", None)
            if hasattr(s, "set_epoch"):
                s.set_epoch(epoch)

    def __iter__(self):
        if self.mode == "sequential":
            for di, it in enumerate(self.iterables):
                for bi, batch in enumerate(it):
                    yield batch, bi, di
            return
        iters = [iter(it) for it in self.iterables]
        done = [False] * len(iters)
        for bi in range(len(self)):
            out = []
            for k in range(len(iters)):
                try:
                    out.append(next(iters[k]))
                except StopIteration:
                    done[k] = True
                    if self.mode == "max_size_cycle":
                        iters[k] = iter(self.iterables[k])
                        out.append(next(iters[k]))
                    elif self.mode == "max_size":
                        out.append(None)
                    else:
                        return
            if self.mode == "max_size" and all(done):
                return
            yield out, bi, 0


class CombinedDataModule(_DMBase):
    """combined.py:31-120 — same constructor (``data_modules``, ``train_mode``, ``val_mode``, ``test_mode``,
    ``predict_mode``; modes as ``CombineMode`` or their string values) and hooks."""

    def __init__(self, data_modules: Sequence, train_mode=CombineMode.MAX_SIZE_CYCLE, val_mode=CombineMode.SEQUENTIAL,
                 test_mode=CombineMode.SEQUENTIAL, predict_mode=CombineMode.SEQUENTIAL):
        if _DMBase is not object:  # pragma: no cover
            super().__init__()
        self.data_modules = list(data_modules)

        def mode(m):
            return CombineMode[m].value if isinstance(m, str) and m in CombineMode.__members__ else CombineMode(m).value

        self.train_mode, self.val_mode = mode(train_mode), mode(val_mode)
        self.test_mode, self.predict_mode = mode(test_mode), mode(predict_mode)
        self.prepare_data_per_node = True
        self.trainer = None
        self._training = True

    # the trainer toggles `training`; the children own the GPU transforms that depend on it
    @property
    def training(self) -> bool:
        return self._training

    @training.setter
    def training(self, v: bool) -> None:
        self._training = v
        for dm in self.data_modules:
            dm.training = v

    def prepare_data(self):
        for dm in self.data_modules:
            dm.trainer = self.trainer
            dm.prepare_data()

    def setup(self, stage: str):
        for dm in self.data_modules:
            dm.trainer = self.trainer
            dm.setup(stage)

    @torch.no_grad()
    def on_after_batch_transfer(self, batch, dataloader_idx: int):
        """combined.py:79-108: a list of sub-batches (one per child) or, in sequential mode, the batch of child ``dataloader_idx``"""
        if isinstance(batch, torch.Tensor):
            return batch
        if isinstance(batch, (list, tuple)):
            return [dm.on_after_batch_transfer(sub, dataloader_idx) if sub is not None else None
                    for dm, sub in zip(self.data_modules, batch)]
        return self.data_modules[dataloader_idx].on_after_batch_transfer(batch, dataloader_idx)

    def train_dataloader(self):
        return CombinedLoader([dm.train_dataloader() for dm in self.data_modules], mode=self.train_mode)

    def val_dataloader(self):
        return CombinedLoader([dm.val_dataloader() for dm in self.data_modules], mode=self.val_mode)

    def test_dataloader(self):
        return CombinedLoader([dm.test_dataloader() for dm in self.data_modules], mode=self.test_mode)

    def predict_dataloader(self):
        return CombinedLoader([dm.predict_dataloader() for dm in self.data_modules], mode=self.predict_mode)
