"""Minimal OME-Zarr (NGFF 0.4, zarr v2 directory store) HCS reader / writer.

The reference reaches its data through ``iohub.open_ome_zarr``; iohub / zarr / numcodecs
are not installed here, so this module implements just the surface the hot path touches:
``plate.positions()`` → ``(name, Position)``, ``Position.channel_names / get_channel_index / zattrs /
["0"]``, and 5-D TCZYX arrays with orthogonal indexing ``img.oindex[t_slice, [channels], z_slice]``.
Chunks may be uncompressed (``compressor: null``) or zlib; blosc needs numcodecs and raises a clear error.
I/O is host-side plumbing — nothing here is accelerated or on the timed path.
"""

from __future__ import annotations

import json
import os
import zlib
from pathlib import Path

import numpy as np


class _OIndex:
    def __init__(self, arr):
        self.arr = arr

    def __getitem__(self, key):
        t, c, z = key[:3]
        a = self.arr
        ts = range(*t.indices(a.shape[0])) if isinstance(t, slice) else [int(t)]
        cs = [int(i) for i in c] if not isinstance(c, slice) else list(range(*c.indices(a.shape[1])))
        zs = range(*z.indices(a.shape[2])) if isinstance(z, slice) else [int(z)]
        out = np.empty((len(ts), len(cs), len(zs), a.shape[3], a.shape[4]), dtype=a.dtype)
        for i, tt in enumerate(ts):
            for j, cc in enumerate(cs):
                out[i, j] = a.read_zrange(tt, cc, zs[0], zs[-1] + 1) if len(zs) else out[i, j]
        return out


class ImageArray:
    """zarr v2 array, 5-D TCZYX."""

    def __init__(self, path: Path, rel: str):
        self.fs_path, self.path = Path(path), rel
        meta = json.loads((self.fs_path / ".zarray").read_text())
        self.shape = tuple(meta["shape"])
        self.chunks = tuple(meta["chunks"])
        self.dtype = np.dtype(meta["dtype"])
        self.fill = meta.get("fill_value", 0) or 0
        self.sep = meta.get("dimension_separator", ".")
        comp = meta.get("compressor")
        self.codec = None if comp is None else comp.get("id")
        if self.codec not in (None, "zlib"):
            raise NotImplementedError(f"compressor {self.codec!r} needs numcodecs (not installed); use null or zlib")
        if meta.get("order", "C") != "C" or meta.get("filters"):
            raise NotImplementedError("only C-order, unfiltered zarr v2 arrays are supported")
        self.frames, self.channels, self.slices, self.height, self.width = self.shape
        self.oindex = _OIndex(self)

    def _chunk(self, idx):
        f = self.fs_path / self.sep.join(str(i) for i in idx)
        if not f.exists():
            return np.full(self.chunks, self.fill, dtype=self.dtype)
        raw = f.read_bytes()
        if self.codec == "zlib":
            raw = zlib.decompress(raw)
        return np.frombuffer(raw, dtype=self.dtype).reshape(self.chunks)

    def read_zrange(self, t: int, c: int, z0: int, z1: int) -> np.ndarray:
        ct, cc, cz, cy, cx = self.chunks
        out = np.empty((z1 - z0, self.height, self.width), dtype=self.dtype)
        for zc in range(z0 // cz, (z1 - 1) // cz + 1):
            for yc in range((self.height + cy - 1) // cy):
                for xc in range((self.width + cx - 1) // cx):
                    ch = self._chunk((t // ct, c // cc, zc, yc, xc))[t % ct, c % cc]
                    zs, ze = max(z0, zc * cz), min(z1, (zc + 1) * cz)
                    ys, ye = yc * cy, min(self.height, (yc + 1) * cy)
                    xs, xe = xc * cx, min(self.width, (xc + 1) * cx)
                    out[zs - z0 : ze - z0, ys:ye, xs:xe] = ch[zs - zc * cz : ze - zc * cz, : ye - ys, : xe - xs]
        return out

    def __getitem__(self, key):
        return self.oindex[key]


class Position:
    def __init__(self, root: Path, name: str):
        self.fs_path, self.name = Path(root) / name, name
        self.zattrs = json.loads((self.fs_path / ".zattrs").read_text())
        self.channel_names = [c["label"] for c in self.zattrs["omero"]["channels"]]

    def get_channel_index(self, name: str) -> int:
        return self.channel_names.index(name)

    def __getitem__(self, key: str) -> ImageArray:
        return ImageArray(self.fs_path / key, f"{self.name}/{key}")

    @property
    def scale(self):
        return self.zattrs["multiscales"][0]["datasets"][0]["coordinateTransformations"][0]["scale"]


class Plate:
    def __init__(self, path):
        self.fs_path = Path(path)
        self.zattrs = json.loads((self.fs_path / ".zattrs").read_text())

    def positions(self):
        for well in self.zattrs["plate"]["wells"]:
            wattrs = json.loads((self.fs_path / well["path"] / ".zattrs").read_text())
            for img in wattrs["well"]["images"]:
                name = f"{well['path']}/{img['path']}"
                yield name, Position(self.fs_path, name)

    def __getitem__(self, name: str) -> Position:
        return Position(self.fs_path, name)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def open_ome_zarr(path, mode: str = "r", layout: str = "hcs", **kw):
    if mode != "r":
        raise NotImplementedError("use write_hcs_plate() to create stores")
    p = Path(path)
    if not (p / ".zattrs").exists():
        raise FileNotFoundError(f"{p} is not an OME-Zarr store")
    attrs = json.loads((p / ".zattrs").read_text())
    return Plate(p) if "plate" in attrs else Position(p.parent.parent.parent, "/".join(p.parts[-3:]))


def write_hcs_plate(path, positions: dict[str, np.ndarray], channel_names: list[str], norm_meta: dict | None = None,
                    chunks: tuple[int, ...] | None = None, compress: bool = False) -> None:
    """Write ``{"A/1/0": array(T,C,Z,Y,X)}`` as an HCS plate (test fixtures / synthetic data)."""
    root = Path(path)
    rows = sorted({k.split("/")[0] for k in positions})
    cols = sorted({k.split("/")[1] for k in positions})
    wells = sorted({"/".join(k.split("/")[:2]) for k in positions})
    root.mkdir(parents=True, exist_ok=True)
    (root / ".zgroup").write_text(json.dumps({"zarr_format": 2}))
    (root / ".zattrs").write_text(json.dumps({"plate": {
        "rows": [{"name": r} for r in rows], "columns": [{"name": c} for c in cols],
        "wells": [{"path": w, "rowIndex": rows.index(w.split("/")[0]), "columnIndex": cols.index(w.split("/")[1])} for w in wells],
        "version": "0.4"}}))
    for w in wells:
        (root / w).mkdir(parents=True, exist_ok=True)
        (root / w.split("/")[0] / ".zgroup").write_text(json.dumps({"zarr_format": 2}))
        (root / w / ".zgroup").write_text(json.dumps({"zarr_format": 2}))
        fovs = sorted(k.split("/")[2] for k in positions if k.startswith(w + "/"))
        (root / w / ".zattrs").write_text(json.dumps({"well": {"images": [{"path": f} for f in fovs], "version": "0.4"}}))
    for name, arr in positions.items():
        arr = np.ascontiguousarray(arr)
        pos = root / name
        (pos / "0").mkdir(parents=True, exist_ok=True)
        (pos / ".zgroup").write_text(json.dumps({"zarr_format": 2}))
        attrs = {"multiscales": [{"axes": [{"name": a} for a in "tczyx"], "version": "0.4", "datasets": [
            {"path": "0", "coordinateTransformations": [{"type": "scale", "scale": [1.0] * 5}]}]}],
            "omero": {"channels": [{"label": c} for c in channel_names]}}
        if norm_meta is not None:
            attrs["normalization"] = norm_meta.get(name, norm_meta)  # per-position dict or one dict for all
        (pos / ".zattrs").write_text(json.dumps(attrs))
        ck = chunks or (1, 1, arr.shape[2], arr.shape[3], arr.shape[4])
        (pos / "0" / ".zarray").write_text(json.dumps({
            "zarr_format": 2, "shape": list(arr.shape), "chunks": list(ck), "dtype": arr.dtype.str, "order": "C",
            "fill_value": 0, "filters": None, "dimension_separator": "/",
            "compressor": {"id": "zlib", "level": 1} if compress else None}))
        grid = [range((s + c - 1) // c) for s, c in zip(arr.shape, ck)]
        import itertools

        for idx in itertools.product(*grid):
            sl = tuple(slice(i * c, (i + 1) * c) for i, c in zip(idx, ck))
            block = np.zeros(ck, dtype=arr.dtype)
            sub = arr[sl]
            block[tuple(slice(0, s) for s in sub.shape)] = sub
            f = pos / "0" / "/".join(str(i) for i in idx)
            f.parent.mkdir(parents=True, exist_ok=True)
            raw = block.tobytes()
            f.write_bytes(zlib.compress(raw, 1) if compress else raw)
