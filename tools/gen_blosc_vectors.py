"""tests/golden/blosc_vectors.npz — c-blosc frames produced by the REAL library, the pins of viscy_amd/data/codecs.py.

Runs only where a libblosc is loadable (the build container has /opt/conda/lib/libblosc.so.1.21.0; numcodecs, which the reference
uses, vendors the same c-blosc 1.x).  Every vector = (raw bytes, the frame libblosc made from them) for one combination of inner
codec, shuffle mode, item size, block size and length — including lengths that are not a multiple of the item size or of 8
items, multi-block frames with a short last block, incompressible (memcpy'd) input, and the configuration iohub writes
(zstd, level 1, bit-shuffle, 4-byte items)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = None
for cand in ("libblosc.so.1", "/opt/conda/lib/libblosc.so.1", "libblosc.so"):
    try:
        lib = C.CDLL(cand)
        break
    except OSError:
        pass
if lib is None:
    sys.exit("no libblosc to generate with")
lib.blosc_compress_ctx.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
lib.blosc_compress_ctx.restype = C.c_int
lib.blosc_decompress_ctx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
lib.blosc_decompress_ctx.restype = C.c_int
lib.blosc_get_version_string.restype = C.c_char_p


def compress(raw: bytes, cname: str, clevel: int, shuffle: int, typesize: int, blocksize: int) -> bytes:
    dst = C.create_string_buffer(len(raw) + 16 + 4096)
    n = lib.blosc_compress_ctx(clevel, shuffle, typesize, len(raw), raw, dst, len(dst), cname.encode(), blocksize, 1)
    assert n > 0, (cname, n)
    back = C.create_string_buffer(max(len(raw), 1))
    assert lib.blosc_decompress_ctx(dst.raw[:n], back, len(raw), 1) == len(raw) and back.raw[: len(raw)] == raw
    return dst.raw[:n]


rng = np.random.default_rng(7)


def payload(kind: str, n: int, typesize: int) -> bytes:
    if kind == "image":  # smooth float32 / uint16 image rows: what a microscopy chunk looks like
        x = np.cumsum(rng.normal(size=n // typesize + 8)).astype(np.float32 if typesize == 4 else np.float64)
        b = (x if typesize in (4, 8) else (x * 100).astype(np.uint16 if typesize == 2 else np.uint8)).tobytes()
        return b[:n]
    if kind == "noise":
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    return bytes(n)  # zeros


vec = {}
cases = []
for cname in ("zstd", "lz4", "zlib", "lz4hc"):
    for shuffle in (0, 1, 2):
        for typesize in (1, 2, 4, 8):
            cases.append((cname, 1 if cname != "lz4hc" else 4, shuffle, typesize, 0, 6000, "image"))
cases += [
    ("zstd", 1, 2, 4, 0, 96 * 96 * 4, "image"),        # iohub's configuration on one (1, 1, 1, 96, 96) float32 chunk
    ("zstd", 1, 2, 4, 0, 1000 * 4 + 3, "image"),       # length not a multiple of the item size
    ("zstd", 1, 2, 4, 8192, 30001, "image"),           # several blocks, short last block
    ("lz4", 5, 1, 4, 4096, 30001, "image"),            # split streams (lz4 + shuffle), several blocks
    ("lz4", 5, 1, 2, 0, 513 * 2, "image"),
    ("zstd", 3, 1, 4, 0, 16384, "noise"),              # incompressible: stored (memcpy flag or raw splits)
    ("lz4", 5, 0, 4, 0, 16384, "noise"),
    ("zstd", 1, 2, 4, 0, 65536, "zeros"),
    ("zstd", 1, 2, 4, 0, 20, "image"),                 # tiny buffer (fewer than 8 items: bit-shuffle skipped)
    ("zstd", 1, 2, 2, 0, 16 * 2 * 77, "image"),
    ("lz4", 1, 2, 8, 0, 8 * 4096, "image"),
]
for i, (cname, clevel, shuffle, typesize, blocksize, n, kind) in enumerate(cases):
    raw = payload(kind, n, typesize)
    frame = compress(raw, cname, clevel, shuffle, typesize, blocksize)
    key = f"{i:03d}_{cname}_l{clevel}_s{shuffle}_t{typesize}_b{blocksize}_n{n}_{kind}"
    vec[key + "__raw"] = np.frombuffer(raw, dtype=np.uint8)
    vec[key + "__frame"] = np.frombuffer(frame, dtype=np.uint8)
out = os.path.join(ROOT, "tests", "golden", "blosc_vectors.npz")
np.savez_compressed(out, libblosc_version=np.array(lib.blosc_get_version_string().decode()), **vec)
print(f"{len(cases)} vectors from libblosc {lib.blosc_get_version_string().decode()} -> {out} ({os.path.getsize(out) / 1e3:.0f} KB)")
