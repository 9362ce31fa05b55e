"""Chunk codecs of the OME-Zarr stores the reference's data layer produces.

The reference reads and writes through iohub / zarr-python / numcodecs (`packages/viscy-data/src/viscy_data/hcs.py:36-829`);
iohub's default compressor is Blosc(zstd, bit-shuffle) for zarr v2 arrays and the `blosc` codec inside `sharding_indexed` for
zarr v3 (its own fixtures build both: `packages/viscy-data/tests/conftest.py:17-66`).  None of those packages is part of this
image, so the decoders live here:

* ``blosc_decompress`` — the c-blosc 1.x container (16-byte header, block offsets, per-block split streams, byte / bit
  un-shuffle), restated from the published format and **pinned on the real library**: the golden vectors of
  ``tests/golden/blosc_vectors.npz`` were compressed by libblosc 1.21.0 itself (``tools/gen_blosc_vectors.py``, run in the
  build container where /opt/conda/lib/libblosc.so exists) and the decoder reproduces every one of them bit for bit
  (``tests/test_codecs_cpu.py``).  Inner codecs: zstd / lz4 / snappy through ``pyarrow.Codec`` (part of the image), zlib through
  the standard library; blosclz (blosc's own LZ77 variant; neither iohub's nor numcodecs' default) is not built.
* plain ``zlib`` / ``gzip`` / ``zstd`` / ``lz4`` (numcodecs framing) chunk compressors of zarr v2, and the zarr v3 codec
  pipeline ``bytes`` -> [``blosc`` | ``zstd`` | ``gzip``] -> [``crc32c``] including ``sharding_indexed``.

Host-side plumbing: nothing here is on the timed path.  When numcodecs is importable it is used instead (same bytes).
"""

from __future__ import annotations

import gzip
import struct
import zlib

import numpy as np

_BLOSC_FORMATS = {0: "blosclz", 1: "lz4", 2: "snappy", 3: "zlib", 4: "zstd"}
_MAX_SPLITS, _MIN_BUFFERSIZE = 16, 128


def _arrow_codec(name: str):
    try:
        import pyarrow as pa
    except ImportError as e:  # pragma: no cover - pyarrow is part of the image
        raise NotImplementedError(f"{name}-compressed chunks need pyarrow (or numcodecs) for the inner codec") from e
    return pa.Codec({"lz4": "lz4_raw"}.get(name, name))


def _inner_decompress(fmt: str, data: bytes, nbytes: int) -> bytes:
    if fmt == "zlib":
        return zlib.decompress(data)
    if fmt in ("zstd", "lz4", "snappy"):
        return _arrow_codec(fmt).decompress(data, decompressed_size=nbytes).to_pybytes()
    raise NotImplementedError(f"blosc inner codec {fmt!r} is not built (zstd, lz4, zlib and snappy are)")


def _unshuffle_bytes(block: bytes, typesize: int) -> bytes:
    n = len(block) // typesize
    a = np.frombuffer(block, dtype=np.uint8)
    out = np.empty(len(block), dtype=np.uint8)
    out[: n * typesize] = a[: n * typesize].reshape(typesize, n).T.reshape(-1)
    out[n * typesize :] = a[n * typesize :]
    return out.tobytes()


def _unshuffle_bits(block: bytes, typesize: int) -> bytes:
    """c-blosc `bitunshuffle` (shuffle.c): the bitshuffle library's layout [byte of element][bit][elements / 8], bit b of a
    packed byte = element 8 g + b; blocks whose element count is not a multiple of 8 were stored un-shuffled"""
    n = len(block) // typesize
    if n % 8:
        return block
    a = np.frombuffer(block, dtype=np.uint8)
    bits = np.unpackbits(a[: n * typesize].reshape(typesize, 8, n // 8), axis=-1, bitorder="little")  # [j, k, g, b]
    elems = np.packbits(bits.reshape(typesize, 8, n).transpose(2, 0, 1), axis=-1, bitorder="little")  # [e, j, 1]
    out = np.empty(len(block), dtype=np.uint8)
    out[: n * typesize] = elems.reshape(-1)
    out[n * typesize :] = a[n * typesize :]
    return out.tobytes()


def blosc_decompress(buf: bytes) -> bytes:
    """one c-blosc 1.x frame -> the bytes it was made from"""
    if len(buf) < 16:
        raise ValueError("blosc frame shorter than its header")
    version, _versionlz, flags, typesize = buf[0], buf[1], buf[2], buf[3]
    nbytes, blocksize, cbytes = struct.unpack_from("<III", buf, 4)
    if version != 2:
        raise NotImplementedError(f"blosc container version {version} (c-blosc 1.x writes 2)")
    if cbytes > len(buf):
        raise ValueError(f"blosc frame truncated: header says {cbytes} bytes, got {len(buf)}")
    typesize = max(typesize, 1)
    if nbytes == 0:
        return b""
    if flags & 0x2:  # memcpyed: the payload follows the header
        return bytes(buf[16 : 16 + nbytes])
    fmt = _BLOSC_FORMATS.get(flags >> 5)
    if fmt is None:
        raise NotImplementedError(f"blosc compressor format {flags >> 5}")
    byte_shuffle, bit_shuffle, dont_split = bool(flags & 0x1) and typesize > 1, bool(flags & 0x4), bool(flags & 0x10)
    nblocks = (nbytes + blocksize - 1) // blocksize
    bstarts = struct.unpack_from(f"<{nblocks}i", buf, 16)
    out = bytearray(nbytes)
    for i in range(nblocks):
        bsize = blocksize
        leftover = i == nblocks - 1 and nbytes % blocksize != 0
        if leftover:
            bsize = nbytes % blocksize
        nsplits = typesize if (not dont_split and typesize <= _MAX_SPLITS and blocksize // typesize >= _MIN_BUFFERSIZE and not leftover) else 1
        neblock = bsize // nsplits
        pos = bstarts[i]
        parts = []
        for _ in range(nsplits):
            (cb,) = struct.unpack_from("<i", buf, pos)
            pos += 4
            chunk = bytes(buf[pos : pos + cb])
            pos += cb
            parts.append(chunk if cb == neblock else _inner_decompress(fmt, chunk, neblock))
        block = b"".join(parts)
        if len(block) != bsize:
            raise ValueError(f"blosc block {i}: decoded {len(block)} bytes, expected {bsize}")
        if byte_shuffle:
            block = _unshuffle_bytes(block, typesize)
        elif bit_shuffle and bsize >= typesize:
            block = _unshuffle_bits(block, typesize)
        out[i * blocksize : i * blocksize + bsize] = block
    return bytes(out)


def _numcodecs():
    try:
        import numcodecs  # noqa: F401

        return numcodecs
    except ImportError:
        return None


def decode_v2(raw: bytes, compressor: dict | None, nbytes: int) -> bytes:
    """a zarr v2 chunk file -> raw C-order bytes.  ``compressor`` = the `.zarray` entry (numcodecs configuration) or None"""
    if compressor is None:
        return raw
    cid = compressor.get("id")
    nc = _numcodecs()
    if nc is not None:  # the reference's own decoder when it is installed
        return bytes(nc.get_codec(compressor).decode(raw))
    if cid == "blosc":
        return blosc_decompress(raw)
    if cid == "zlib":
        return zlib.decompress(raw)
    if cid == "gzip":
        return gzip.decompress(raw)
    if cid == "zstd":
        return _arrow_codec("zstd").decompress(raw, decompressed_size=nbytes).to_pybytes()
    if cid == "lz4":  # numcodecs.LZ4: 4-byte little-endian size, then one raw LZ4 block
        (n,) = struct.unpack_from("<I", raw, 0)
        return _arrow_codec("lz4").decompress(raw[4:], decompressed_size=n).to_pybytes()
    raise NotImplementedError(f"zarr v2 compressor {cid!r} is not built (null, zlib, gzip, zstd, lz4 and blosc are)")


def encode_v2(raw: bytes, compressor: dict | None) -> bytes:
    """the writer's side: stores are created uncompressed or zlib-compressed (readable by every zarr implementation)"""
    if compressor is None:
        return raw
    if compressor.get("id") == "zlib":
        return zlib.compress(raw, int(compressor.get("level", 1)))
    nc = _numcodecs()
    if nc is not None:
        return bytes(nc.get_codec(compressor).encode(raw))
    raise NotImplementedError(f"writing {compressor.get('id')!r}-compressed chunks needs numcodecs; new stores use null or zlib")


# ------------------------------------------------------------------ zarr v3 codec pipeline
def _v3_bytes_to_bytes(raw: bytes, codec: dict, nbytes: int) -> bytes:
    name, cfg = codec["name"], codec.get("configuration", {}) or {}
    if name == "blosc":
        return blosc_decompress(raw)
    if name == "zstd":
        return _arrow_codec("zstd").decompress(raw, decompressed_size=nbytes).to_pybytes()
    if name == "gzip":
        return gzip.decompress(raw)
    if name == "crc32c":
        return raw[:-4]  # checksum of the stored bytes (no crc32c in the standard library: not verified)
    raise NotImplementedError(f"zarr v3 codec {name!r} ({cfg}) is not built")


def decode_v3(raw: bytes, codecs: list[dict], dtype: np.dtype, chunk_shape: tuple[int, ...]) -> np.ndarray:
    """apply a zarr v3 codec list backwards to one stored (inner) chunk -> ndarray of ``chunk_shape``"""
    nbytes = int(np.prod(chunk_shape)) * dtype.itemsize
    a2b = [c for c in codecs if c["name"] in ("bytes", "transpose")]
    for c in reversed([c for c in codecs if c["name"] not in ("bytes", "transpose")]):
        raw = _v3_bytes_to_bytes(raw, c, nbytes)
    order = None
    for c in a2b:
        if c["name"] == "transpose":
            order = tuple(c["configuration"]["order"])
        elif (c.get("configuration") or {}).get("endian", "little") != "little" and dtype.itemsize > 1:
            dtype = dtype.newbyteorder(">")
    if order is not None and order != tuple(range(len(chunk_shape))):
        arr = np.frombuffer(raw, dtype=dtype).reshape([chunk_shape[i] for i in order])
        return np.ascontiguousarray(arr.transpose(np.argsort(order)))
    return np.frombuffer(raw, dtype=dtype).reshape(chunk_shape)


def read_shard_index(raw: bytes, chunks_per_shard: tuple[int, ...], index_codecs: list[dict], index_location: str = "end") -> np.ndarray:
    """`sharding_indexed`: the (offset, nbytes) table of a shard file, shape chunks_per_shard + (2,), uint64 (all ones = empty)"""
    n = int(np.prod(chunks_per_shard))
    size = n * 16 + (4 if any(c["name"] == "crc32c" for c in index_codecs) else 0)
    blob = raw[-size:] if index_location == "end" else raw[:size]
    return np.frombuffer(blob[: n * 16], dtype="<u8").reshape(tuple(chunks_per_shard) + (2,))
