"""GPU-only: the affine LayerNorm backward launches of the bench step (stem / downsampling LayerNorms: dgamma, dbeta by atomics) under
`ln_ablk` = cap on workgroups; us per launch and GB/s on dy + x + dx.   python tools/perf_ln.py [ln_ablk=2048 ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viscy_amd import _lib as L  # noqa: E402
from viscy_amd import ops  # noqa: E402


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


caps = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("ln_ablk=")] or [512, 1024, 2048, 4096, 0]
dt = torch.bfloat16
for rows, C in ((512 * 64 * 64, 96), (512 * 32 * 32, 192), (512 * 16 * 16, 384), (512 * 8 * 8, 768)):
    x = torch.randn(rows, C, device="cuda").to(dt)
    dy = torch.randn(rows, C, device="cuda").to(dt)
    mean, rstd = torch.randn(rows, device="cuda"), torch.rand(rows, device="cuda") + 0.5
    gamma = torch.randn(C, device="cuda")
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    gb = 3 * rows * C * 2 / 1e9
    ref = None
    for cap in caps:
        assert L.lib().vsx_set_flag(b"ln_ablk", cap) == 0
        dg.zero_(); db.zero_()
        dx = ops.ln_bwd(dy, x, mean, rstd, gamma, None, dg, db, rows, C)
        if ref is None:
            ref = (dx.clone(), dg.clone(), db.clone())
        else:
            assert torch.equal(dx, ref[0])
            for a, b in ((dg, ref[1]), (db, ref[2])):
                assert (a - b).abs().max() <= 1e-3 * b.abs().max(), (a - b).abs().max()
        us = timeit(lambda: ops.ln_bwd(dy, x, mean, rstd, gamma, None, dg, db, rows, C))
        print(f"rows {rows:8d} C {C:4d} ln_ablk {cap:5d}: {us:8.1f} us  {gb / us * 1e6:6.0f} GB/s", flush=True)

# ---- lane packing (ln_pack): forward, affine-free backward and affine backward at the step's row shapes
if "pack" in sys.argv:
    assert L.lib().vsx_set_flag(b"ln_ablk", 512) == 0
    for rows, C in ((512 * 64 * 64, 96), (512 * 32 * 32, 192), (512 * 16 * 16, 384), (512 * 8 * 8, 768), (512 * 64 * 64, 224)):
        x = torch.randn(rows, C, device="cuda").to(dt)
        dy = torch.randn(rows, C, device="cuda").to(dt)
        gamma, beta = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
        dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        outs = {}
        for pk in (0, 1):
            assert L.lib().vsx_set_flag(b"ln_pack", pk) == 0
            y, mean, rstd = ops.ln_fwd(x, gamma, beta, rows, C)
            dx0 = ops.ln_bwd(dy, x, mean, rstd, None, None, None, None, rows, C)
            dg.zero_(); db.zero_()
            dx1 = ops.ln_bwd(dy, x, mean, rstd, gamma, None, dg, db, rows, C)
            outs[pk] = (y, mean, rstd, dx0, dx1, dg.clone(), db.clone())
            t_f = timeit(lambda: ops.ln_fwd(x, gamma, beta, rows, C))
            t_b = timeit(lambda: ops.ln_bwd(dy, x, mean, rstd, None, None, None, None, rows, C))
            t_a = timeit(lambda: ops.ln_bwd(dy, x, mean, rstd, gamma, None, dg, db, rows, C))
            gbf, gbb = 2 * rows * C * 2 / 1e9, 3 * rows * C * 2 / 1e9
            print(f"rows {rows:8d} C {C:4d} ln_pack {pk}: fwd {t_f:7.1f} us {gbf / t_f * 1e6:5.0f} GB/s | bwd {t_b:7.1f} us {gbb / t_b * 1e6:5.0f} GB/s | "
                  f"bwd affine {t_a:7.1f} us {gbb / t_a * 1e6:5.0f} GB/s", flush=True)
        for a, b in zip(outs[0][:5], outs[1][:5]):  # same arithmetic per row up to the summation order inside the row
            err = (a.float() - b.float()).abs().max().item()
            assert err <= 2e-2 * max(1.0, b.float().abs().max().item()), err
        for a, b in zip(outs[0][5:], outs[1][5:]):
            assert (a - b).abs().max() <= 2e-3 * b.abs().max(), (a - b).abs().max()
