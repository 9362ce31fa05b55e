"""Build libvsx.so (hand-written gfx950 kernels + C-ABI) in-tree with hipcc.

    python -m viscy_amd.build            # incremental: recompiles only stale objects
hipcc cross-compiles for gfx950 without a GPU; the .so travels with the repo snapshot.
"""

from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.environ.get("VSX_CSRC", os.path.join(HERE, "csrc"))  # VSX_CSRC: build a variant from another source tree (same-box A/B of two commits)
BUILD = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libvsx.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


SF32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]  # whole-file scalar fp32 VALU (A/B variants only)


def build(verbose: bool = True, variant: str | None = None, sf32_files: tuple[str, ...] = (), defines: tuple[str, ...] = ()) -> str:
    """``variant`` (A/B builds, `VSX_LIB=viscy_amd/libvsx_<variant>.so` selects one at load time): objects go to
    ``_build_<variant>/``, the library to ``libvsx_<variant>.so``; ``sf32_files`` are compiled without packed-fp32 VALU
    instructions, ``defines`` are passed as -D to every file.  The shipped library is the plain ``build()``."""
    global BUILD, LIB
    if variant:
        BUILD, LIB = os.path.join(HERE, "_build_" + variant), os.path.join(HERE, f"libvsx_{variant}.so")
    else:
        BUILD, LIB = os.path.join(HERE, "_build"), os.path.join(HERE, "libvsx.so")
    os.makedirs(BUILD, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "vsx.h"))
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(BUILD, s[:-4] + ".o")
        objs.append(obj)
        if _stale(obj, [src] + hdrs):
            extra = (SF32 if s in sf32_files else []) + ["-D" + d for d in defines]
            jobs.append([hipcc, *FLAGS, *extra, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    _write_build_info()
    return LIB


def _write_build_info() -> None:
    """the GPU box receives a snapshot without .git: record HEAD (and whether the tree was dirty) next to the objects so that
    bench.py can put it into its JSON line"""
    import json

    root = os.path.dirname(HERE)
    if not os.path.isdir(os.path.join(root, ".git")):
        return
    try:
        sha = subprocess.run(["git", "-C", root, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=20).stdout.strip()
        dirty = bool(subprocess.run(["git", "-C", root, "status", "--porcelain", "--untracked-files=no"], capture_output=True,
                                    text=True, timeout=20).stdout.strip())
        with open(os.path.join(BUILD, "build_info.json"), "w") as f:
            json.dump({"git_sha": sha, "git_dirty": dirty}, f)
    except (OSError, subprocess.SubprocessError):
        pass


if __name__ == "__main__":
    # python -m viscy_amd.build [variant [sf32:file.hip,file.hip] [-DNAME ...]]
    var = sys.argv[1] if len(sys.argv) > 1 else None
    sf = tuple(f for a in sys.argv[2:] if a.startswith("sf32:") for f in a[5:].split(",") if f)
    dd = tuple(a[2:] for a in sys.argv[2:] if a.startswith("-D"))
    print(build(variant=var, sf32_files=sf, defines=dd))
    sys.exit(0)
