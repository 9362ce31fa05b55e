import os, sys, torch
sys.path.insert(0, "/root/repo")
from viscy_amd import _lib as L, ops
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
dt = torch.bfloat16
L.lib().vsx_set_flag(b"ln_pack", 0)
for rows, C in ((512*64*64, 96), (512*32*32, 192), (512*16*16, 384), (512*64*64, 224)):
    x = torch.randn(rows, C, device="cuda").to(dt)
    gamma, beta = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    for fblk in (32768, 8192, 4096, 2048, 1024):
        L.lib().vsx_set_flag(b"ln_fblk", fblk)
        ta = timeit(lambda: ops.ln_fwd(x, gamma, beta, rows, C))
        tn = timeit(lambda: ops.ln_fwd(x, None, None, rows, C))
        tm = timeit(lambda: ops.ln_fwd(x, None, None, rows, C, need_mean=False))
        print(f"rows {rows} C {C} ln_fblk {fblk}: affine {ta:7.1f} | plain {tn:7.1f} | plain no mean {tm:7.1f} us", flush=True)
