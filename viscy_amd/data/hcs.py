"""``HCSDataModule`` / ``SlidingWindowDataset`` — the data-module surface of the Cytoland pipeline
(/root/reference/packages/viscy-data/src/viscy_data/hcs.py:124-829, sliding_window.py:21-286), kept so
that ``VSUNet`` + the HIP path drop in behind the same batch contract:

  Sample = {"source": (B,Cs,Z,Y,X) f32, "target": (B,Ct,Z,Y,X) f32,
            "index": (list["/row/col/fov/0"], t, z), "norm_meta": {channel: {level: {stat: Tensor}}}}

Covered: fit / predict / test set-up, seeded FOV train/val split, Z-sliding windows over T, CPU
normalisations + augmentations composed per sample in the workers, collation of multi-sample crops,
``on_after_batch_transfer`` GPU augmentations + spatial-shape validation + ``target_2d`` slicing,
``DistributedSampler`` sharding under DP, mmap preload (``prepare_data`` stages the fit FOVs into one memory-mapped buffer
under ``scratch_dir``), precomputed foreground masks (``fg_mask_key``: loaded next to the images, co-transformed by
every spatial augmentation, staged into ``fg_mask.mmap``) and non-zero rejection sampling (``min_nonzero_fraction``).
Not built: ``ground_truth_masks`` (test-stage segmentation labels for the Cellpose metrics, out of scope).  I/O is plain
host code.
"""

from __future__ import annotations

from pathlib import Path
from typing import Callable, Iterable, Sequence

import numpy as np
import torch
from torch import Tensor
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.distributed import DistributedSampler

from .ome_zarr import open_ome_zarr

try:  # pragma: no cover
    from lightning.pytorch import LightningDataModule as _DMBase
except Exception:  # noqa: BLE001
    _DMBase = object


def _ensure_channel_list(x) -> list[str]:
    return [x] if isinstance(x, str) else list(x)


def _read_norm_meta(fov) -> dict | None:
    """viscy_data/_utils.py:139-165: zattrs["normalization"] → float32 0-d tensors."""
    meta = fov.zattrs.get("normalization")
    if meta is None:
        return None
    out = {}
    for ch, levels in meta.items():
        out[ch] = {}
        for level, stats in levels.items():
            if level == "timepoint_statistics":
                out[ch][level] = {t: {k: torch.tensor(v, dtype=torch.float32) for k, v in s.items()} for t, s in stats.items()}
            else:
                out[ch][level] = {k: torch.tensor(v, dtype=torch.float32) for k, v in stats.items()}
    return out


class Compose:
    def __init__(self, transforms: Sequence[Callable]):
        self.transforms = list(transforms)

    def __call__(self, sample):
        for t in self.transforms:
            if isinstance(sample, list):
                sample = [t(s) for s in sample]
            else:
                sample = t(sample)
        return sample


class _MmapArray:
    """the slice of the staged buffer that holds one FOV, with the ImageArray surface SlidingWindowDataset uses"""

    def __init__(self, slab: np.ndarray, path: str):
        self._slab, self.path = slab, path
        self.frames, self.channels, self.slices, self.height, self.width = slab.shape
        self.shape, self.dtype = slab.shape, slab.dtype
        self.oindex = self

    def __getitem__(self, idx):
        t, c, z = idx
        return np.ascontiguousarray(self._slab[t, :, z][:, list(c)])


class _MmapPosition:
    def __init__(self, pos, slab: np.ndarray, channel_names: list[str], array_key: str, mask_key: str | None = None,
                 mask_slab: np.ndarray | None = None):
        self._pos, self._names, self._array_key = pos, list(channel_names), array_key
        self._arr = _MmapArray(slab, pos[array_key].path)
        self._mask_key = mask_key
        self._mask = _MmapArray(mask_slab, pos[mask_key].path) if mask_slab is not None else None
        self.name, self.zattrs, self.channel_names = pos.name, pos.zattrs, list(channel_names)

    def __contains__(self, key):
        return key == self._array_key or key in self._pos

    def __getitem__(self, key):
        if key == self._array_key:
            return self._arr
        if self._mask is not None and key == self._mask_key:
            return self._mask  # target channels only (resolve_mask_ch_indices: "target-only" layout)
        return self._pos[key]

    def get_channel_index(self, name: str) -> int:
        return self._names.index(name)


def _is_spatial(t) -> bool:
    """foreground_masks.py:13-27: ``is_spatial`` where the transform declares it (every viscy_amd.transforms class does)"""
    if hasattr(t, "is_spatial"):
        return bool(t.is_spatial)
    return any("spatial" in getattr(b, "__module__", "") or "croppad" in getattr(b, "__module__", "") for b in type(t).__mro__)


class ForegroundMaskSupport:
    """Precomputed foreground masks next to the images (viscy_data/foreground_masks.py:30-228): owns the mask arrays of a
    ``SlidingWindowDataset``, reads the window the dataset reads, and names the temporary per-channel keys under which
    the masks ride through the spatial transforms of the CPU pipeline."""

    def __init__(self, fg_mask_key: str, target_channels: list[str]):
        self.fg_mask_key, self.target_channels = fg_mask_key, list(target_channels)
        self._mask_keys = self.mask_temp_keys(self.target_channels)
        self._mask_arrays, self._mask_ch_indices = [], []

    @staticmethod
    def mask_temp_keys(target_channels) -> tuple[str, ...]:
        return tuple(f"__fg_mask_{ch}" for ch in target_channels)

    @property
    def mask_keys(self) -> tuple[str, ...]:
        return self._mask_keys

    @staticmethod
    def resolve_mask_ch_indices(n_mask_ch: int, n_image_ch: int, n_target: int, target_ch_idx, mask_key: str = "fg_mask") -> list[int]:
        """full-channel mask (same channels as the image) -> the target channels' indices; target-only mask -> 0 .. n-1"""
        if n_mask_ch == n_image_ch:
            return list(target_ch_idx)
        if n_mask_ch == n_target:
            return list(range(n_target))
        raise ValueError(f"Mask array '{mask_key}' has {n_mask_ch} channels, expected {n_image_ch} (all image channels) or "
                         f"{n_target} (target channels only).")

    def validate_and_store(self, fov, img_arr, target_ch_idx) -> None:
        if self.fg_mask_key not in fov:
            raise FileNotFoundError(f"Mask array '{self.fg_mask_key}' not found in position. "
                                    "Run preprocessing with --compute_fg_masks first.")
        mask_arr = fov[self.fg_mask_key]
        self._mask_ch_indices.append(self.resolve_mask_ch_indices(mask_arr.channels, img_arr.channels, len(self.target_channels),
                                                                  target_ch_idx, self.fg_mask_key))
        self._mask_arrays.append(mask_arr)

    def read_window(self, arr_idx: int, t: int, z: int, z_window: int) -> list[Tensor]:
        data = self._mask_arrays[arr_idx].oindex[slice(t, t + 1), self._mask_ch_indices[arr_idx], slice(z, z + z_window)]
        return list(torch.from_numpy(np.asarray(data).astype(np.float32)).unbind(dim=1))  # per channel (1, Z, Y, X)

    def inject_into_sample(self, sample_images: dict, mask_images: list[Tensor]) -> None:
        for key, m in zip(self._mask_keys, mask_images):
            sample_images[key] = m

    @staticmethod
    def patch_spatial_transforms(transforms, target_keys, mask_keys) -> None:
        """append the mask keys to every SPATIAL transform that touches a target key (in place, idempotent); intensity
        transforms never see the masks"""
        for t in transforms:
            keys = getattr(t, "keys", None)
            if keys is None or not _is_spatial(t):
                continue
            if any(k in keys for k in target_keys) and not any(k in keys for k in mask_keys):
                t.keys = type(keys)(list(keys) + list(mask_keys)) if isinstance(keys, (list, tuple)) else list(keys) + list(mask_keys)
                t.allow_missing_keys = True


class SlidingWindowDataset(Dataset):
    """All (FOV, t, z-window) positions of an HCS plate; see sliding_window.py:21-286.  ``min_nonzero_fraction`` > 0 turns on
    rejection sampling (a window whose check channel — ``nonzero_channel`` or the first target — has a smaller fraction of
    voxels >= ``nonzero_threshold``, or of foreground-mask voxels when masks are loaded, is replaced by a uniformly drawn
    other window, up to ``max_nonzero_retries`` times); ``fg_mask_key`` adds the ``"fg_mask"`` entry (B, Ct, Z, Y, X)."""

    def __init__(self, positions, channels: dict[str, list[str]], z_window_size: int, array_key: str = "0",
                 transform: Callable | None = None, load_normalization_metadata: bool = True, min_nonzero_fraction: float = 0.0,
                 nonzero_threshold: float = 0.0, nonzero_channel: str | None = None, max_nonzero_retries: int = 100,
                 fg_mask_key: str | None = None):
        if not 0.0 <= min_nonzero_fraction <= 1.0:
            raise ValueError(f"min_nonzero_fraction must be in [0, 1], got {min_nonzero_fraction}")
        if max_nonzero_retries < 0:
            raise ValueError(f"max_nonzero_retries must be >= 0, got {max_nonzero_retries}")
        self.positions, self.channels = list(positions), {k: list(v) for k, v in channels.items()}
        self.z_window_size, self.array_key, self.transform = z_window_size, array_key, transform
        self.load_normalization_metadata = load_normalization_metadata
        self.min_nonzero_fraction, self.nonzero_threshold = min_nonzero_fraction, nonzero_threshold
        self.nonzero_channel, self.max_nonzero_retries = nonzero_channel, max_nonzero_retries
        names = self.channels["source"] + [c for c in self.channels.get("target", []) if c not in self.channels["source"]]
        self._all_ch_names = names
        if nonzero_channel is not None and nonzero_channel not in names:
            raise ValueError(f"nonzero_channel '{nonzero_channel}' not found in channels: {names}")
        targets = self.channels.get("target", [])
        self.fg_mask_support = ForegroundMaskSupport(fg_mask_key, targets) if fg_mask_key is not None and targets else None
        self._windows = []  # cumulative window counts
        self._arrays, self._ch_idx, self._norm = [], [], []
        total = 0
        for fov in self.positions:
            img = fov[array_key]
            zs = img.slices - z_window_size + 1
            if zs < 1:
                raise ValueError(f"z_window_size {z_window_size} exceeds the {img.slices} slices of {fov.name}")
            total += img.frames * zs
            self._windows.append(total)
            self._arrays.append(img)
            self._ch_idx.append([fov.get_channel_index(c) for c in names])
            self._norm.append(_read_norm_meta(fov))
            if self.fg_mask_support is not None:
                self.fg_mask_support.validate_and_store(fov, img, [fov.get_channel_index(c) for c in targets])
        self._max_window = total

    def __len__(self) -> int:
        return self._max_window

    def _find_window(self, index: int):
        i = int(np.searchsorted(self._windows, index, side="right"))
        tz = index - (self._windows[i - 1] if i else 0)
        return i, tz

    def _index_of(self, index: int):
        """the (path, t, z) a window index addresses (before any rejection re-draw)"""
        i, tz = self._find_window(index)
        zs = self._arrays[i].slices - self.z_window_size + 1
        return (f"/{self._arrays[i].path}", tz // zs, tz % zs)

    def _read_window(self, index: int):
        i, tz = self._find_window(index)
        img = self._arrays[i]
        zs = img.slices - self.z_window_size + 1
        t, z = tz // zs, tz % zs
        data = img.oindex[slice(t, t + 1), self._ch_idx[i], slice(z, z + self.z_window_size)].astype(np.float32)
        images = dict(zip(self._all_ch_names, torch.from_numpy(data).unbind(dim=1)))  # each (1, Z, Y, X)
        masks = None
        if self.fg_mask_support is not None:
            masks = self.fg_mask_support.read_window(i, t, z, self.z_window_size)
        return i, t, z, img, images, masks

    def __getitem__(self, index: int):
        # rejection sampling (sliding_window.py:222-252): the mask, when loaded, is read once and decides the fraction
        check_key = (self.nonzero_channel or self.channels.get("target", [None])[0]) if self.min_nonzero_fraction > 0 else None
        idx = index
        for attempt in range(self.max_nonzero_retries + 1):
            i, t, z, img, images, masks = self._read_window(idx)
            if check_key is not None:
                if masks is not None and check_key in self.channels.get("target", []):
                    m = masks[self.channels["target"].index(check_key)]
                    frac = m.sum().item() / m.numel()
                elif check_key in images:
                    patch = images[check_key]
                    frac = (patch >= self.nonzero_threshold).sum().item() / patch.numel()
                else:
                    break
                if frac < self.min_nonzero_fraction:
                    if attempt < self.max_nonzero_retries:
                        idx = torch.randint(len(self), ()).item()
                        continue
                    import logging

                    logging.getLogger("viscy_amd.data").warning(
                        f"Exhausted {self.max_nonzero_retries} retries for nonzero fraction >= {self.min_nonzero_fraction} on "
                        f"channel '{check_key}' (index {index}). Returning last sample.")
            break
        has_masks = masks is not None
        if has_masks:  # temporary per-channel keys: the spatial transforms of the pipeline co-align them with the target
            self.fg_mask_support.inject_into_sample(images, masks)
        sample_index = (f"/{img.path}", t, z)
        norm_meta = self._norm[i]
        if norm_meta is not None:
            nm = {}
            for ch, levels in norm_meta.items():  # resolve timepoint statistics (sliding_window.py:148-164)
                nm[ch] = {lv: (st.get(str(t), next(iter(st.values()))) if lv == "timepoint_statistics" else st)
                          for lv, st in levels.items()}
            images["norm_meta"] = norm_meta = nm
        if "target" in self.channels:
            images["weight"] = images[self.channels["target"][0]]
        if self.transform:
            images = self.transform(images)

        def build(im):
            im.pop("weight", None)
            s = {"index": sample_index, "source": torch.stack([im[c][0] for c in self.channels["source"]])}
            if "target" in self.channels:
                s["target"] = torch.stack([im[c][0] for c in self.channels["target"]])
            if has_masks:
                s["fg_mask"] = torch.stack([im[k][0] for k in self.fg_mask_support.mask_keys])
            if self.load_normalization_metadata and norm_meta is not None:
                s["norm_meta"] = norm_meta
            return s

        return [build(im) for im in images] if isinstance(images, list) else build(images)


def _collate_samples(batch):
    """viscy_data/_utils.py:112-136: flatten per-stack lists of patches, stack tensors, keep indices as lists."""
    flat = []
    for b in batch:
        flat.extend(b if isinstance(b, list) else [b])
    out = {}
    for k in flat[0]:
        vals = [s[k] for s in flat]
        if torch.is_tensor(vals[0]):
            out[k] = torch.stack(vals)
        elif k == "index":
            out[k] = ([v[0] for v in vals], torch.tensor([v[1] for v in vals]), torch.tensor([v[2] for v in vals]))
        elif k == "norm_meta":
            out[k] = {ch: {lv: {st: torch.stack([v[ch][lv][st] for v in vals]) for st in vals[0][ch][lv]}
                           for lv in vals[0][ch]} for ch in vals[0]}
        else:
            out[k] = vals
    return out


class HCSDataModule(_DMBase):
    def __init__(self, data_path: str, source_channel, target_channel, z_window_size: int, split_ratio: float = 0.8,
                 batch_size: int = 16, num_workers: int = 8, target_2d: bool = False, yx_patch_size=(256, 256),
                 normalizations: list | None = None, augmentations: list | None = None, mmap_preload: bool = False,
                 scratch_dir=None, ground_truth_masks=None, persistent_workers=False, prefetch_factor=None,
                 array_key: str = "0", pin_memory=True, min_nonzero_fraction: float = 0.0, nonzero_threshold: float = 0.0,
                 nonzero_channel=None, max_nonzero_retries: int = 100, fg_mask_key=None, gpu_augmentations: list | None = None,
                 val_augmentations: list | None = None, val_gpu_augmentations: list | None = None,
                 include_fov_names: Iterable[str] | None = None, exclude_fov_names: Iterable[str] | None = None, seed: int = 42,
                 normalize_on_device: bool = True):
        if _DMBase is not object:  # pragma: no cover
            super().__init__()
        if ground_truth_masks is not None:
            raise NotImplementedError("ground_truth_masks (test-stage segmentation labels for the Cellpose metrics) is out of scope")
        self.min_nonzero_fraction, self.nonzero_threshold = min_nonzero_fraction, nonzero_threshold
        self.nonzero_channel, self.max_nonzero_retries, self.fg_mask_key = nonzero_channel, max_nonzero_retries, fg_mask_key
        if fg_mask_key is not None:  # hcs.py:188-191: the batched masks ride under "fg_mask" next to "target"
            for chain in (gpu_augmentations, val_gpu_augmentations):
                if chain:
                    ForegroundMaskSupport.patch_spatial_transforms(chain, ("target",), ("fg_mask",))
        self.mmap_preload = bool(mmap_preload)
        self.scratch_dir = Path(scratch_dir) if scratch_dir is not None else None
        self.data_path = Path(data_path)
        self.source_channel, self.target_channel = _ensure_channel_list(source_channel), _ensure_channel_list(target_channel)
        self.batch_size, self.num_workers, self.target_2d = batch_size, num_workers, target_2d
        self.z_window_size, self.split_ratio, self.yx_patch_size = z_window_size, split_ratio, tuple(yx_patch_size)
        self.normalizations, self.augmentations = normalizations or [], augmentations or []
        self.val_augmentations = val_augmentations or []
        self.array_key, self.pin_memory = array_key, pin_memory
        self.persistent_workers, self.prefetch_factor = persistent_workers, prefetch_factor
        self.include_fov_names = set(include_fov_names) if include_fov_names is not None else None
        self.exclude_fov_names = set(exclude_fov_names) if exclude_fov_names is not None else None
        from ..transforms import fuse_affine_crop

        self._gpu_augmentations = Compose(fuse_affine_crop(gpu_augmentations)) if gpu_augmentations else None
        self._val_gpu_augmentations = Compose(fuse_affine_crop(val_gpu_augmentations)) if val_gpu_augmentations else None
        self.prepare_data_per_node = True
        self.seed = seed
        self.training = True  # set by the trainer loop (Lightning: trainer.training / trainer.validating)
        self.train_patches_per_stack = 1
        self._patch_error, self._is_batched_concat_child = None, False
        # MI355X-first: when nothing intensity-dependent runs in the workers, ship raw patches + statistics and normalise
        # in HBM right after the transfer (one vsx_normalize launch per stacked key) instead of per sample on host cores.
        from ..transforms import NormalizeSampled
        self.normalize_on_device = bool(normalize_on_device and self.normalizations and not self.augmentations
                                        and not self.val_augmentations
                                        and all(type(n) is NormalizeSampled for n in self.normalizations))
        for aug in self.augmentations:
            n = getattr(getattr(aug, "cropper", None), "num_samples", None) or getattr(aug, "num_samples", None)
            if n:
                if batch_size % n:  # raised in setup("fit") (hcs.py:787-800), not for children of a BatchedConcatDataModule
                    self._patch_error = (f"Batch size must be divisible by `num_samples` per stack. Got batch size {batch_size} "
                                         f"and number of samples {n} for transform type {type(aug)}.")
                self.train_patches_per_stack = n

    @staticmethod
    def _inject_mask_keys(transforms, target_keys, mask_keys) -> None:
        """hcs.py:195-216"""
        ForegroundMaskSupport.patch_spatial_transforms(transforms, target_keys, mask_keys)

    # ---- mmap preload (viscy_data/hcs.py:218-349): stage the fit FOVs once, uncompressed, into one memory-mapped buffer
    @property
    def _mmap_cache_dir(self) -> Path:
        """hcs.py:218-234 — same fingerprint (dataset path, channels, array key, FOV filters), same directory layout"""
        import hashlib
        import os
        import tempfile

        scratch = self.scratch_dir or Path(tempfile.gettempdir())
        ch_key = "|".join(self.source_channel) + "||" + "|".join(self.target_channel) + f"||{self.array_key}"
        include_key = "|".join(sorted(self.include_fov_names)) if self.include_fov_names else ""
        exclude_key = "|".join(sorted(self.exclude_fov_names)) if self.exclude_fov_names else ""
        path_key = str(self.data_path.resolve()) + "||" + ch_key + f"||incl={include_key}||excl={exclude_key}"
        fingerprint = hashlib.md5(path_key.encode()).hexdigest()[:12]
        return scratch / os.getenv("SLURM_JOB_ID", "viscy_cache") / f"{self.data_path.name}_{fingerprint}"

    @staticmethod
    def _fov_t_offsets(positions, array_key: str) -> list[int]:
        """hcs.py:351-378: cumulative T offsets, one slab of the buffer per FOV (T may differ between FOVs)"""
        offsets = [0]
        for pos in positions:
            offsets.append(offsets[-1] + pos[array_key].frames)
        return offsets

    def _mmap_layout(self, positions):
        all_ch = list(self.source_channel) + [c for c in self.target_channel if c not in self.source_channel]
        arr0 = positions[0][self.array_key]
        offsets = self._fov_t_offsets(positions, self.array_key)
        shape = (offsets[-1], len(all_ch), arr0.slices, arr0.height, arr0.width)
        return all_ch, offsets, shape, np.dtype(arr0.dtype)

    def prepare_data(self):
        """hcs.py:241-349.  The buffer is a raw ``numpy.memmap`` file ``data.mmap`` (the reference uses tensordict's
        ``MemoryMappedTensor``, absent here: same bytes, same ``.done`` marker protocol, partial caches are rebuilt)."""
        if not self.mmap_preload:
            return
        import shutil
        from concurrent.futures import ThreadPoolExecutor

        cache_dir = self._mmap_cache_dir
        if self._mmap_cache_ready(cache_dir):
            return
        if cache_dir.exists():
            shutil.rmtree(cache_dir)  # partial files of a killed preload, or a marker whose buffers were cleaned up
        cache_dir.mkdir(parents=True, exist_ok=True)
        try:
            plate = open_ome_zarr(self.data_path, mode="r")
            positions = self._filtered_positions(plate)
            all_ch, offsets, shape, dtype = self._mmap_layout(positions)
            buf = np.lib.format.open_memmap(cache_dir / "data.mmap", mode="w+", dtype=dtype, shape=shape)

            def write_fov(i_pos):
                i, pos = i_pos
                img = pos[self.array_key]
                if (img.slices, img.height, img.width) != shape[2:]:
                    raise ValueError(f"{pos.name}: array shape {(img.slices, img.height, img.width)} differs from {shape[2:]}")
                ch_idx = [pos.get_channel_index(c) for c in all_ch]
                buf[offsets[i]:offsets[i + 1]] = img.oindex[slice(None), ch_idx, slice(None)]

            with ThreadPoolExecutor(max_workers=min(len(positions), 16)) as pool:
                list(pool.map(write_fov, enumerate(positions)))
            buf.flush()
            del buf
            if self.fg_mask_key:
                # hcs.py:313-340: the masks of the TARGET channels, row for row aligned with the data buffer (same offsets)
                arr0, mask0 = positions[0][self.array_key], positions[0][self.fg_mask_key]
                n_target = len(self.target_channel)
                tgt_idx = [positions[0].get_channel_index(c) for c in self.target_channel]
                mask_ch = ForegroundMaskSupport.resolve_mask_ch_indices(mask0.channels, arr0.channels, n_target, tgt_idx, self.fg_mask_key)
                mbuf = np.lib.format.open_memmap(cache_dir / "fg_mask.mmap", mode="w+", dtype=np.dtype(mask0.dtype),
                                                 shape=(shape[0], n_target, *shape[2:]))

                def write_mask(i_pos):
                    i, pos = i_pos
                    mbuf[offsets[i]:offsets[i + 1]] = pos[self.fg_mask_key].oindex[slice(None), mask_ch, slice(None)]

                with ThreadPoolExecutor(max_workers=min(len(positions), 16)) as pool:
                    list(pool.map(write_mask, enumerate(positions)))
                mbuf.flush()
                del mbuf
            (cache_dir / ".done").touch()
        except BaseException:
            if cache_dir.exists():
                shutil.rmtree(cache_dir)
            raise

    def _mmap_cache_ready(self, cache_dir: Path) -> bool:
        """hcs.py:515-545: the marker AND every buffer this configuration needs"""
        required = [cache_dir / ".done", cache_dir / "data.mmap"]
        if self.fg_mask_key is not None:
            required.append(cache_dir / "fg_mask.mmap")
        return all(f.exists() for f in required)

    def _mmap_positions(self, positions):
        """the FOVs of ``positions`` served from the staged buffer (read-only map shared by forked workers)"""
        cache_dir = self._mmap_cache_dir
        if not self._mmap_cache_ready(cache_dir):
            raise RuntimeError(f"mmap_preload=True but no staged buffer at {cache_dir}: call prepare_data() before setup('fit')")
        all_ch, offsets, shape, dtype = self._mmap_layout(positions)
        buf = np.load(cache_dir / "data.mmap", mmap_mode="r")
        if tuple(buf.shape) != tuple(shape) or buf.dtype != dtype:
            raise RuntimeError(f"stale mmap cache at {cache_dir}: buffer {buf.shape} {buf.dtype}, dataset needs {shape} {dtype}; "
                               "delete the directory to rebuild it")
        mbuf = np.load(cache_dir / "fg_mask.mmap", mmap_mode="r") if self.fg_mask_key is not None else None
        if mbuf is not None and (mbuf.shape[0] != shape[0] or tuple(mbuf.shape[2:]) != tuple(shape[2:])):
            raise RuntimeError(f"stale mmap cache at {cache_dir}: fg_mask buffer {mbuf.shape} does not match the data buffer {shape}")
        return [_MmapPosition(pos, buf[offsets[i]:offsets[i + 1]], all_ch, self.array_key, self.fg_mask_key,
                              mbuf[offsets[i]:offsets[i + 1]] if mbuf is not None else None) for i, pos in enumerate(positions)]

    def _filtered_positions(self, plate):
        pos = [p for name, p in plate.positions()
               if (self.include_fov_names is None or name in self.include_fov_names)
               and (self.exclude_fov_names is None or name not in self.exclude_fov_names)]
        if not pos:
            raise ValueError(f"No positions left in {self.data_path} after applying include_fov_names / exclude_fov_names filters")
        return pos

    def setup(self, stage: str):
        settings = dict(channels={"source": self.source_channel}, z_window_size=self.z_window_size, array_key=self.array_key)
        if self.fg_mask_key is not None and stage != "predict":  # hcs.py:669-671: masks feed the training loss only
            settings["fg_mask_key"] = self.fg_mask_key
        plate = open_ome_zarr(self.data_path, mode="r")
        positions = self._filtered_positions(plate)
        if stage in ("fit", "validate"):
            if self._patch_error and not self._is_batched_concat_child:
                raise ValueError(self._patch_error)
            if self.mmap_preload:  # predict / test read the zarr store directly (hcs.py:243-247)
                positions = self._mmap_positions(positions)
            settings["channels"]["target"] = self.target_channel
            g = torch.Generator().manual_seed(self.seed)
            idx = torch.randperm(len(positions), generator=g).tolist()  # hcs.py:490-494,566-569
            positions = [positions[i] for i in idx]
            n_train = int(len(positions) * self.split_ratio)
            norms = [] if self.normalize_on_device else self.normalizations
            if self.fg_mask_key is not None:  # hcs.py:776-783: the CPU spatial augmentations co-transform the per-channel masks
                mask_keys = ForegroundMaskSupport.mask_temp_keys(self.target_channel)
                ForegroundMaskSupport.patch_spatial_transforms(self.augmentations, tuple(self.target_channel), mask_keys)
                ForegroundMaskSupport.patch_spatial_transforms(self.val_augmentations, tuple(self.target_channel), mask_keys)
            train_filter = {}
            if self.min_nonzero_fraction > 0:  # hcs.py:466-476: rejection sampling is a training-set setting only
                train_filter = dict(min_nonzero_fraction=self.min_nonzero_fraction, nonzero_threshold=self.nonzero_threshold,
                                    max_nonzero_retries=self.max_nonzero_retries)
                if self.nonzero_channel is not None:
                    train_filter["nonzero_channel"] = self.nonzero_channel
            self.train_dataset = SlidingWindowDataset(positions[:n_train], transform=Compose(norms + self.augmentations), **settings,
                                                      **train_filter)
            self.val_dataset = SlidingWindowDataset(positions[n_train:], transform=Compose(norms + self.val_augmentations), **settings)
        elif stage == "test":
            settings["channels"]["target"] = self.target_channel
            self.test_dataset = SlidingWindowDataset(positions, transform=Compose([] if self.normalize_on_device else self.normalizations), **settings)
        elif stage == "predict":
            self.predict_dataset = SlidingWindowDataset(positions, transform=Compose([] if self.normalize_on_device else self.normalizations), **settings)
        else:
            raise NotImplementedError(f"{stage} stage not supported")

    def _loader(self, ds, batch_size, shuffle, drop_last=False):
        sampler = None
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            # what Lightning injects for plain HCSDataModule (hcs.py:723-735): stock DistributedSampler
            sampler = DistributedSampler(ds, shuffle=shuffle, seed=self.seed, drop_last=drop_last)
            shuffle = False
        return DataLoader(ds, batch_size=batch_size, num_workers=self.num_workers, shuffle=shuffle, sampler=sampler,
                          drop_last=drop_last, collate_fn=_collate_samples, pin_memory=self.pin_memory and torch.cuda.is_available(),
                          persistent_workers=self.persistent_workers and self.num_workers > 0,
                          prefetch_factor=self.prefetch_factor if self.num_workers else None)

    def train_dataloader(self):
        return self._loader(self.train_dataset, self.batch_size // self.train_patches_per_stack, shuffle=True, drop_last=True)

    def val_dataloader(self):
        return self._loader(self.val_dataset, self.batch_size, shuffle=False)

    def test_dataloader(self):
        return self._loader(self.test_dataset, 1, shuffle=False)

    def predict_dataloader(self):
        return self._loader(self.predict_dataset, self.batch_size, shuffle=False)

    def _device_normalize(self, batch):
        """NormalizeSampled (_normalize.py:27-81) for the stacked ``source`` / ``target`` keys: per-(sample, channel)
        subtrahend / divisor gathered from the collated ``norm_meta`` and applied by ONE kernel launch per key."""
        from ..transforms import normalize_stacked
        meta = batch.get("norm_meta")
        if meta is None:
            raise ValueError("normalize_on_device needs `norm_meta` in the batch (load_normalization_metadata=True)")
        for key, channels in (("source", self.source_channel), ("target", self.target_channel)):
            x = batch.get(key)
            if x is None:
                continue
            B, C = x.shape[:2]
            sub, div = torch.zeros(B, C), torch.ones(B, C)
            touched = False
            for n in self.normalizations:
                for ch in n.keys:
                    if ch in channels:
                        c = channels.index(ch)
                        st = meta[ch][n.level]
                        # composition of successive normalisations of one channel is not affine-foldable in general
                        if (sub[:, c] != 0).any() or (div[:, c] != 1).any():
                            raise NotImplementedError(f"two normalisations of channel {ch}: set normalize_on_device=False")
                        sub[:, c], div[:, c] = st[n.subtrahend].float().cpu(), st[n.divisor].float().cpu()
                        touched = True
            if touched:
                batch[key] = normalize_stacked(x, sub, div)
        if any(n.remove_meta for n in self.normalizations):
            batch.pop("norm_meta", None)
        return batch

    @torch.no_grad()
    def on_after_batch_transfer(self, batch, dataloader_idx: int):
        """hcs.py:679-721: GPU augmentations, target_2d slicing, training-shape validation."""
        if isinstance(batch, Tensor):
            return batch
        if self.normalize_on_device:
            batch = self._device_normalize(batch)
        if self.training and self._gpu_augmentations is not None:
            batch = self._gpu_augmentations(batch)
        elif not self.training and self._val_gpu_augmentations is not None:
            batch = self._val_gpu_augmentations(batch)
        if self.target_2d and "target" in batch:
            z_index = self.z_window_size // 2
            batch["target"] = batch["target"][:, :, slice(z_index, z_index + 1)]
            if "fg_mask" in batch:
                batch["fg_mask"] = batch["fg_mask"][:, :, slice(z_index, z_index + 1)]
        if self.training and self._gpu_augmentations is None and "source" in batch:
            expected = (self.z_window_size, self.yx_patch_size[0], self.yx_patch_size[1])
            actual = tuple(batch["source"].shape[2:])
            if actual != expected:
                raise ValueError(f"Source spatial shape {actual} does not match expected {expected} "
                                 f"(z_window_size={self.z_window_size}, yx_patch_size={list(self.yx_patch_size)}). "
                                 f"Configure gpu_augmentations with a spatial crop to match yx_patch_size.")
        return batch
