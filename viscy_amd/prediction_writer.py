"""``HCSPredictionWriter``: stores virtual-staining predictions as an HCS OME-Zarr plate, blending the overlapping
Z windows of 2.5-D / 3-D sliding-window prediction.

Same constructor, hook names and on-disk result as the reference callback
(/root/reference/packages/viscy-utils/src/viscy_utils/callbacks/prediction_writer.py:114-362): prediction channels are
named ``<target>_prediction``; a new store is created with the data module's channel layout (optionally with the
source / target centre slices, ``write_input``), an existing one is appended to (``FileExistsError`` if a prediction
channel is already there and ``overwrite`` is off); arrays are 5-D TCZYX with ``(1, 1, 1, Y, X)`` chunks and the input
store's scale transform; a window written at Z offset ``z`` is feathered into what earlier windows left there with
``old * (f - 1) / f + new / f``, ``f = min(i + 1, min(z + 1, depth))`` counted from the far end of the window
(prediction_writer.py:74-111).

Round 5 — where the blend happens.  The reference re-reads and re-writes the overlapped slices of the store for every window
(``old = image.oindex[...]``, blend, write back: 5 chunk reads + 5 chunk writes per channel and window at depth 5).  Here the
windows of one (image, timepoint) are blended **where the prediction already is**: a running stack of the not-yet-final slices
stays on the prediction's device (``viscy_amd.vsunet.blend_in`` = ``vsx_blend_in`` on the GPU; the same arithmetic in torch for
CPU predictions), a slice is written to the store ONCE — when the next window starts behind it or prediction ends — and nothing
is read back.  Windows that do not continue the running stack (a store being overwritten, shuffled or resumed prediction)
take the reference's read-blend-write path on the host, so the on-disk result is the reference's in every order of arrival.
Derives from Lightning's ``BasePredictionWriter`` when Lightning is installed; otherwise
``viscy_amd.trainer.Trainer(callbacks=[...])`` drives the same hooks.
"""

from __future__ import annotations

import os
from typing import Literal, Optional, Sequence

import numpy as np
import torch

from .data.ome_zarr import Plate, Position, open_ome_zarr

try:  # pragma: no cover - Lightning is not part of this image
    from lightning.pytorch.callbacks import BasePredictionWriter as _Base
except Exception:  # noqa: BLE001

    class _Base:
        def __init__(self, write_interval: str = "batch"):
            if write_interval not in ("batch", "epoch", "batch_and_epoch"):
                raise ValueError(f"write_interval {write_interval!r}")
            self.interval = write_interval


def _pad_shape(shape: tuple[int, ...], target: int = 5) -> tuple[int, ...]:
    return (1,) * (target - len(shape)) + tuple(shape)


def blend_in_host(old_stack: np.ndarray, new_stack: np.ndarray, z_slice: slice) -> np.ndarray:
    """prediction_writer._blend_in for (C, Z, Y, X) numpy stacks."""
    if z_slice.start == 0:
        return new_stack
    depth = z_slice.stop - z_slice.start
    samples = min(z_slice.start + 1, depth)
    f = np.array([min(i + 1, samples) for i in reversed(range(depth))], dtype=np.float64)[None, :, None, None]
    return (old_stack * (f - 1) / f + new_stack / f).astype(new_stack.dtype)


class HCSPredictionWriter(_Base):
    def __init__(self, output_store: str, overwrite: bool = False, write_input: bool = False,
                 write_interval: Literal["batch", "epoch", "batch_and_epoch"] = "batch") -> None:
        super().__init__(write_interval)
        self.output_store = output_store
        self.overwrite = overwrite
        self.write_input = write_input
        self._dataset_scale = None

    def _get_scale_metadata(self, metadata_store) -> None:
        if metadata_store is None:
            return
        store = open_ome_zarr(metadata_store, mode="r")
        if isinstance(store, Position):
            self._dataset_scale = [{"type": "scale", "scale": list(store.scale)}]
        elif isinstance(store, Plate):
            for _, pos in store.positions():
                self._dataset_scale = [{"type": "scale", "scale": list(pos.scale)}]
                break

    def on_predict_start(self, trainer, pl_module) -> None:
        dm = trainer.datamodule
        self._get_scale_metadata(dm.data_path)
        self.z_padding = dm.z_window_size // 2 if dm.target_2d else 0
        source_channel, target_channel = list(dm.source_channel), list(dm.target_channel)
        prediction_channel = [ch + "_prediction" for ch in target_channel]
        if os.path.exists(self.output_store):
            if self.write_input:
                raise FileExistsError("Cannot write input to an existing store. Aborting.")
            self.plate = open_ome_zarr(self.output_store, mode="r+")
            needs_append = []
            for _, pos in self.plate.positions():  # validate every position before mutating any
                existing = set(pos.channel_names)
                for ch in prediction_channel:
                    if ch in existing and not self.overwrite:
                        self.plate.close()
                        raise FileExistsError(f"Channel '{ch}' already exists in '{self.output_store}'. Set overwrite=True to replace.")
                missing = [ch for ch in prediction_channel if ch not in existing]
                if missing:
                    needs_append.append((pos, missing))
            for pos, channels in needs_append:
                for ch in channels:
                    pos.append_channel(ch, resize_arrays=True)
            self.plate._channel_names = None  # re-read after the appends
        else:
            channel_names = prediction_channel
            if self.write_input:
                channel_names = source_channel + target_channel + channel_names
            self.plate = open_ome_zarr(self.output_store, layout="hcs", mode="a", channel_names=channel_names)
        if self.write_input:
            self.source_index = self._get_channel_indices(source_channel)
            self.target_index = self._get_channel_indices(target_channel)
        self.prediction_index = self._get_channel_indices(prediction_channel)

    def _get_channel_indices(self, channel_names) -> list[int]:
        return [self.plate.get_channel_index(ch) for ch in channel_names]

    def write_on_batch_end(self, trainer, pl_module, prediction: torch.Tensor, batch_indices: Optional[Sequence[int]], batch,
                           batch_idx: int, dataloader_idx: int) -> None:
        pred = prediction.detach().float()   # stays where it is: blended windows leave the device once, when they are final
        for sample_index, _ in enumerate(batch["index"][0]):
            self.write_sample(batch, pred[sample_index], sample_index)

    def on_predict_end(self, trainer, pl_module) -> None:
        for key in list(getattr(self, "_running", {})):
            self._flush(key)
        self.plate.close()

    # ---- running blend of the open (image, timepoint) stacks
    MAX_OPEN = 4  # stacks held at once: the loader walks one field of view after the other

    def _flush(self, key, upto: int | None = None) -> None:
        """write the slices [z0, upto) of a running stack (all of it by default) — each slice reaches the store exactly once"""
        run = self._running[key]
        n = run["stack"].shape[1] if upto is None else min(max(upto - run["z0"], 0), run["stack"].shape[1])
        if n > 0:
            img_name, t_index = key
            image = self.plate[img_name]
            image.oindex[t_index, self.prediction_index, slice(run["z0"], run["z0"] + n)] = run["stack"][:, :n].cpu().numpy()
        if upto is None or n == run["stack"].shape[1]:
            del self._running[key]
        else:
            run["stack"], run["z0"] = run["stack"][:, n:], run["z0"] + n

    def _blend_running(self, key, pred: torch.Tensor, z_slice: slice) -> bool:
        """merge one Z window into the running stack of its (image, timepoint); False if the window does not continue it (the
        caller then blends against the store, the reference's way)"""
        running = self.__dict__.setdefault("_running", {})
        depth = z_slice.stop - z_slice.start
        run = running.get(key)
        if run is None:
            if z_slice.start != 0:
                return False   # the middle of a volume with nothing held: whatever is in the store has to be read
            while len(running) >= self.MAX_OPEN:
                self._flush(next(iter(running)))
            running[key] = {"z0": 0, "last": 0, "stack": pred.clone()}
            return True
        if z_slice.start != run["last"] + 1 or z_slice.start < run["z0"] or z_slice.start > run["z0"] + run["stack"].shape[1]:
            self._flush(key)
            return False
        self._flush(key, upto=z_slice.start)          # slices behind this window are final
        run = running.get(key)
        held = 0 if run is None else run["stack"].shape[1]
        old = pred.new_zeros(pred.shape)              # beyond the held slices the window meets a weight of (f - 1) / f = 0
        if held:
            old[:, :held] = run["stack"]
        if pred.is_cuda and (pred.shape[-1] * pred.shape[-2]) % 4 == 0:  # the device kernel moves 16-byte vectors of a plane
            from .vsunet import blend_in

            merged = blend_in(old, pred, z_slice)
        else:
            samples = min(z_slice.start + 1, depth)
            f = torch.tensor([float(min(i + 1, samples)) for i in reversed(range(depth))], dtype=torch.float64).view(1, -1, 1, 1)
            merged = (old.double() * (f - 1) / f + pred.double() / f).to(pred.dtype)
        running[key] = {"z0": z_slice.start, "last": z_slice.start, "stack": merged}
        return True

    def write_sample(self, batch, sample_prediction, sample_index: int) -> None:
        if not torch.is_tensor(sample_prediction):
            sample_prediction = torch.as_tensor(np.asarray(sample_prediction))
        sample_prediction = sample_prediction.detach().float()
        img_name, t_index, z_index = [batch["index"][i][sample_index] for i in range(3)]
        t_index, z_index = int(t_index), int(z_index)
        z_index += self.z_padding  # slices lost at the borders in 2.5-D
        depth = sample_prediction.shape[-3]
        z_slice = slice(z_index, z_index + depth)
        image = self._create_image(img_name, tuple(sample_prediction.shape), np.float32)
        if image.shape[0] <= t_index or image.shape[2] < z_slice.stop:
            image.resize((max(t_index + 1, image.shape[0]), image.channels, max(z_slice.stop, image.shape[2]), *image.shape[-2:]))
        if self.write_input:
            source_stack = batch["source"][sample_index].detach().float().cpu().numpy()
            centre = source_stack.shape[-3] // 2
            image[t_index, self.source_index, z_index] = source_stack[:, centre]
            if "target" in batch:
                target_stack = batch["target"][sample_index].detach().float().cpu().numpy()
                image[t_index, self.target_index, z_index] = target_stack[:, target_stack.shape[-3] // 2]
        if self.z_padding == 0 and depth > 1:
            if self._blend_running((img_name, t_index), sample_prediction, z_slice):
                return
            host = sample_prediction.cpu().numpy()
            old_stack = image.oindex[slice(t_index, t_index + 1), self.prediction_index, z_slice][0]
            image.oindex[t_index, self.prediction_index, z_slice] = blend_in_host(old_stack, host, z_slice)
            return
        image.oindex[t_index, self.prediction_index, z_slice] = sample_prediction.cpu().numpy()

    def _create_image(self, img_name: str, shape, dtype):
        try:
            return self.plate[img_name]
        except KeyError:
            pass
        _, row_name, col_name, pos_name, arr_name = img_name.split("/")
        position = self.plate.create_position(row_name, col_name, pos_name)
        shape = [1] + list(shape)
        shape[1] = len(position.channel_names)
        return position.create_zeros(arr_name, shape=shape, dtype=dtype, chunks=_pad_shape(tuple(shape[-2:]), 5),
                                     transform=self._dataset_scale)
