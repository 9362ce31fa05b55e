"""Per-op micro-benchmarks at the bench shapes (B=32, 256x256, tiny). Prints us / GB/s / TFLOP/s."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viscy_amd import ops, _lib as L

dt = torch.bfloat16
dev = "cuda"
B = int(os.environ.get("B", 32))

def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us

def rnd(*s, dtype=dt): return torch.randn(*s, device=dev).to(dtype)

stages = [("s0", 64, 96), ("s1", 32, 192), ("s2", 16, 384), ("s3", 8, 768), ("d2", 64, 224)]
which = sys.argv[1:] or ["grn", "dw", "gemm", "ln", "loss"]
for name, hw, C in stages:
    M = B * hw * hw; N4 = 4 * C
    if "grn" in which:
        dz, h = rnd(M, N4), rnd(M, N4); s = torch.ones(B, N4, device=dev); t = torch.zeros(B, N4, device=dev); cs = torch.zeros(N4, device=dev)
        us = timeit(lambda: ops.grn_gelu_bwd(dz, h, s, t, cs, M, N4, hw * hw))
        print(f"grn_gelu_bwd {name} M={M} N={N4}: {us:8.1f} us  {3*M*N4*2/us/1e3:8.1f} GB/s")
    if "dw" in which:
        x, dy = rnd(M, C), rnd(M, C); w = torch.randn(49, C, device=dev); b = torch.randn(C, device=dev)
        us = timeit(lambda: ops.dwconv7_fwd(x, w, b, B, hw, hw, C)); print(f"dwconv7_fwd {name} C={C} hw={hw}: {us:8.1f} us  {2*M*C*2/us/1e3:8.1f} GB/s  {2*49*M*C/us/1e6:6.2f} TFLOP/s")
        us = timeit(lambda: ops.dwconv7_bwd_data(dy, w, x, B, hw, hw, C)); print(f"dwconv7_bwd_data {name}: {us:8.1f} us  {3*M*C*2/us/1e3:8.1f} GB/s")
        dw, db = torch.zeros(49, C, device=dev), torch.zeros(C, device=dev)
        us = timeit(lambda: ops.dwconv7_bwd_weight(dy, x, dw, db, B, hw, hw, C)); print(f"dwconv7_bwd_weight {name}: {us:8.1f} us  {2*M*C*2/us/1e3:8.1f} GB/s  {2*49*M*C/us/1e6:6.2f} TFLOP/s")
    if "ln" in which:
        x = rnd(M, C); g = torch.ones(C, device=dev); bb = torch.zeros(C, device=dev)
        us = timeit(lambda: ops.ln_fwd(x, g, bb, M, C)); print(f"ln_fwd affine {name} C={C}: {us:8.1f} us  {2*M*C*2/us/1e3:8.1f} GB/s")
        us = timeit(lambda: ops.ln_fwd(x, None, None, M, C, 1e-6, need_mean=False)); print(f"ln_fwd plain  {name} C={C}: {us:8.1f} us  {2*M*C*2/us/1e3:8.1f} GB/s")
        xh, _, rstd = ops.ln_fwd(x, None, None, M, C, 1e-6, need_mean=False); dyy = rnd(M, C)
        us = timeit(lambda: ops.ln_bwd(dyy, xh, None, rstd, None, None, None, None, M, C)); print(f"ln_bwd plain  {name} C={C}: {us:8.1f} us  {3*M*C*2/us/1e3:8.1f} GB/s")
    if "gemm" in which:
        xh, W1, hbuf = rnd(M, C), rnd(N4, C), torch.empty(M, N4, device=dev, dtype=dt)
        b1 = torch.zeros(N4, device=dev); colsq = torch.zeros(B, N4, device=dev)
        for epi, nm in [(L.EPI_BIAS, "bias"), (L.EPI_BIAS_GELU_SQ, "gelu_sq")]:
            us = timeit(lambda: ops.gemm("nt", xh, W1, hbuf, M, N4, C, C, C, N4, dtype=dt, epi=epi, bias=b1, red0=colsq, hw=hw*hw))
            print(f"gemm_nt fc1[{nm}] {name} M={M} N={N4} K={C}: {us:8.1f} us  {(M*C+M*N4)*2/us/1e3:8.1f} GB/s  {2*M*N4*C/us/1e6:7.1f} TFLOP/s")
        W2, out, res = rnd(C, N4), torch.empty(M, C, device=dev, dtype=dt), rnd(M, C)
        s = torch.ones(B, N4, device=dev); gb = torch.zeros(N4, device=dev); b2 = torch.zeros(C, device=dev)
        us = timeit(lambda: ops.gemm("nt", hbuf, W2, out, M, C, N4, N4, N4, C, dtype=dt, epi=L.EPI_BIAS_RES, bias=b2, res=res, ldr=C))
        print(f"gemm_nt fc2[plain] {name} M={M} N={C} K={N4}: {us:8.1f} us  {(M*N4+2*M*C)*2/us/1e3:8.1f} GB/s  {2*M*N4*C/us/1e6:7.1f} TFLOP/s")
        us = timeit(lambda: ops.gemm("nt", hbuf, W2, out, M, C, N4, N4, N4, C, dtype=dt, pro=L.PRO_GRN, grn_s=s, grn_b=gb, hw=hw*hw, epi=L.EPI_BIAS_RES, bias=b2, res=res, ldr=C))
        print(f"gemm_nt fc2[grn]   {name} M={M} N={C} K={N4}: {us:8.1f} us  {(M*N4+2*M*C)*2/us/1e3:8.1f} GB/s  {2*M*N4*C/us/1e6:7.1f} TFLOP/s")
        dW = torch.zeros(N4, C, device=dev)
        us = timeit(lambda: ops.gemm("tn", xh, hbuf, dW, M, N4, C, C, N4, C, dtype=dt))
        print(f"gemm_tn dW1 {name} M={M} N={N4} K={C}: {us:8.1f} us  {(M*C+M*N4)*2/us/1e3:8.1f} GB/s  {2*M*N4*C/us/1e6:7.1f} TFLOP/s")
        dW2 = torch.zeros(C, N4, device=dev)
        us = timeit(lambda: ops.gemm("tn", hbuf, out, dW2, M, C, N4, N4, C, N4, dtype=dt, pro=L.PRO_GRN, grn_s=s, grn_b=gb, hw=hw*hw))
        print(f"gemm_tn dW2[grn] {name} M={M} N={C} K={N4}: {us:8.1f} us  {(M*C+M*N4)*2/us/1e3:8.1f} GB/s  {2*M*N4*C/us/1e6:7.1f} TFLOP/s")
if "loss" in which:
    from viscy_amd.losses import MixedLoss
    p = torch.randn(B, 2, 5, 256, 256, device=dev, requires_grad=True); t = torch.rand(B, 2, 5, 256, 256, device=dev)
    crit = MixedLoss()
    def f():
        l = crit(p, t); l.backward()
    us = timeit(f, 5); print(f"MixedLoss fwd+bwd B={B}: {us:8.1f} us   ({B*2*5*256*256*4*2/1e6:.0f} MB of stacks)")
