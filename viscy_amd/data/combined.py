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


# ------------------------------------------------------------------------------------------------ concatenated modules
import bisect  # noqa: E402
import math  # noqa: E402
from collections import defaultdict  # noqa: E402

from torch.utils.data import ConcatDataset, DataLoader  # noqa: E402
from torch.utils.data.distributed import DistributedSampler  # noqa: E402


class ShardedDistributedSampler(DistributedSampler):
    """viscy_data/distributed.py:16-58: rank r draws a random permutation of ITS contiguous shard of the dataset (the last
    shard is shifted back so that every shard has ``num_samples`` elements — the last two may overlap), seeded by
    ``seed + epoch``; unshuffled it is the stock strided split.  Keeps a rank's reads inside one region of the store."""

    def _sharded_randperm(self, max_size: int, generator) -> list[int]:
        perms = [torch.randperm(self.num_samples, generator=generator) + min(i * self.num_samples, max_size - self.num_samples)
                 for i in range(self.num_replicas)]
        return torch.stack(perms, dim=1).reshape(-1).tolist()

    def __iter__(self):
        max_size = len(self.dataset)
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            indices = self._sharded_randperm(max_size, g)
        else:
            indices = list(range(max_size))
        if not self.drop_last:
            pad = self.total_size - len(indices)
            indices += indices[:pad] if pad <= len(indices) else (indices * math.ceil(pad / len(indices)))[:pad]
        else:
            indices = indices[: self.total_size]
        assert len(indices) == self.total_size
        indices = indices[self.rank : self.total_size : self.num_replicas]
        assert len(indices) == self.num_samples
        return iter(indices)


def _no_collation(x):
    return x


class BatchedConcatDataset(ConcatDataset):
    """combined.py:123-183: batched access; the indices of one batch are grouped by constituent dataset and each group comes
    back as ONE collated micro-batch tagged with ``_dataset_idx``"""

    def __getitem__(self, idx):
        raise NotImplementedError

    def _get_sample_indices(self, idx: int) -> tuple[int, int]:
        if idx < 0:
            if -idx > len(self):
                raise ValueError("absolute value of index should not exceed dataset length")
            idx = len(self) + idx
        d = bisect.bisect_right(self.cumulative_sizes, idx)
        return d, idx if d == 0 else idx - self.cumulative_sizes[d - 1]

    def __getitems__(self, indices: list[int]) -> list[dict]:
        from .hcs import _collate_samples

        grouped = defaultdict(list)
        for idx in indices:
            d, s = self._get_sample_indices(idx)
            grouped[d].append(s)
        out = []
        for d, sample_indices in grouped.items():
            ds = self.datasets[d]
            mb = ds.__getitems__(sample_indices) if hasattr(ds, "__getitems__") else _collate_samples([ds[i] for i in sample_indices])
            mb["_dataset_idx"] = d
            out.append(mb)
        return out


class ConcatDataModule(_DMBase):
    """combined.py:186-283: one loader over the concatenation of the children's datasets (uniform sampling over all elements);
    batch size, workers etc. are the first child's and must agree."""

    _ConcatDataset = ConcatDataset

    def __init__(self, data_modules: Sequence):
        if _DMBase is not object:  # pragma: no cover
            super().__init__()
        self.data_modules = list(data_modules)
        first = self.data_modules[0]
        self.num_workers, self.batch_size = first.num_workers, first.batch_size
        self.persistent_workers, self.prefetch_factor = first.persistent_workers, first.prefetch_factor
        self.pin_memory = first.pin_memory
        for dm in self.data_modules:
            if dm.num_workers != self.num_workers:
                raise ValueError("Inconsistent number of workers")
            if dm.batch_size != self.batch_size:
                raise ValueError("Inconsistent batch size")
        self.prepare_data_per_node = True
        self.trainer = None
        self._training = True

    training = CombinedDataModule.training

    def prepare_data(self):
        for dm in self.data_modules:
            dm.trainer = self.trainer
            dm.prepare_data()

    def setup(self, stage: str):
        """Fit-only container (as the reference's): every child is set up, their train / val datasets are concatenated, and
        the children that multi-sample (``train_patches_per_stack`` > 0) must agree on that count."""
        if stage != "fit":
            raise NotImplementedError("Only fit stage is supported")
        for child in self.data_modules:
            child.trainer = self.trainer
            child.setup(stage)
        counts = {int(n) for n in (getattr(child, "train_patches_per_stack", 0) for child in self.data_modules) if n}
        if len(counts) > 1:
            raise ValueError("Inconsistent patches per stack")
        self.train_patches_per_stack = counts.pop() if counts else 0
        self.train_dataset, self.val_dataset = (self._ConcatDataset([getattr(child, name) for child in self.data_modules])
                                                for name in ("train_dataset", "val_dataset"))

    def _dataloader_kwargs(self) -> dict:
        return {"num_workers": self.num_workers, "persistent_workers": self.persistent_workers and self.num_workers > 0,
                "prefetch_factor": self.prefetch_factor if self.num_workers else None,
                "pin_memory": self.pin_memory and torch.cuda.is_available()}

    def train_dataloader(self):
        from .hcs import _collate_samples

        return DataLoader(self.train_dataset, shuffle=True, batch_size=self.batch_size // max(self.train_patches_per_stack, 1),
                          collate_fn=_collate_samples, drop_last=True, **self._dataloader_kwargs())

    def val_dataloader(self):
        from .hcs import _collate_samples

        return DataLoader(self.val_dataset, shuffle=False, batch_size=self.batch_size, drop_last=False, collate_fn=_collate_samples,
                          **self._dataloader_kwargs())

    def on_after_batch_transfer(self, batch, dataloader_idx: int):
        return batch


class BatchedConcatDataModule(ConcatDataModule):
    """combined.py:286-378 (the joint-dataset benchmark recipes): the loader hands over un-collated lists of per-dataset
    micro-batches; ``on_after_batch_transfer`` runs each child's GPU transforms on its micro-batch and concatenates the
    tensors into one batch.  Under torch.distributed every rank iterates its own shard (``ShardedDistributedSampler``)."""

    _ConcatDataset = BatchedConcatDataset

    def setup(self, stage: str):
        for dm in self.data_modules:
            dm._is_batched_concat_child = True
        super().setup(stage)

    def _maybe_sampler(self, dataset, shuffle: bool):
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            return ShardedDistributedSampler(dataset, shuffle=shuffle)
        return None

    def train_dataloader(self):
        sampler = self._maybe_sampler(self.train_dataset, shuffle=True)
        return DataLoader(self.train_dataset, batch_size=self.batch_size, shuffle=False if sampler else True, sampler=sampler,
                          drop_last=True, collate_fn=_no_collation, **self._dataloader_kwargs())

    def val_dataloader(self):
        sampler = self._maybe_sampler(self.val_dataset, shuffle=False)
        return DataLoader(self.val_dataset, batch_size=self.batch_size, shuffle=False, sampler=sampler, drop_last=False,
                          collate_fn=_no_collation, **self._dataloader_kwargs())

    def on_after_batch_transfer(self, batch, dataloader_idx: int):
        if not isinstance(batch, list):
            return batch
        done = []
        for mb in batch:
            if isinstance(mb, dict) and "_dataset_idx" in mb:
                dm = self.data_modules[mb.pop("_dataset_idx")]
                mb = dm.on_after_batch_transfer(mb, dataloader_idx)
            done.append(mb)
        out = {}
        for key, first in done[0].items():
            if isinstance(first, list):
                out[key] = [v for mb in done if key in mb for v in mb[key]]
            elif isinstance(first, torch.Tensor):
                out[key] = torch.cat([mb[key] for mb in done if key in mb], dim=0)
            # per-dataset metadata (norm_meta dicts, index tuples) has no joint meaning: dropped, as in the reference
        return out
