"""The weight-gradient (TN) launches of the bench step, one by one: microseconds and GB/s of the algorithmic operand bytes.
B = 512 shapes of convnextv2_tiny.  FLAG=name, VALUES=a,b: one column per value of that vsx_set_flag knob (A/B of a kernel variant)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viscy_amd import _lib as L  # noqa: E402
from viscy_amd import ops  # noqa: E402

B = int(os.environ.get("B", 512))
REP = int(os.environ.get("REP", 10))
dt = torch.bfloat16
# (name, hw, C_x = N, C_y = K, per_sample)
CASES = [
    ("d2 dW2 per-sample (dout^T g)", 4096, 224, 896, True), ("d2 dW1 (dh^T xh)", 4096, 896, 224, False),
    ("s0 dW2 per-sample", 4096, 96, 384, True), ("s0 dW1", 4096, 384, 96, False),
    ("s1 dW2 per-sample", 1024, 192, 768, True), ("s1 dW1", 1024, 768, 192, False),
    ("s2 dW2 per-sample", 256, 384, 1536, True), ("s2 dW1", 256, 1536, 384, False),
    ("s3 dW1 (C = 768)", 64, 3072, 768, False), ("d1 dW1", 1024, 768, 192, False), ("d0 dW1", 256, 1536, 384, False),
    ("proj d2 (N = 224, K = 288)", 4096, 224, 288, False),
]
FLAG = os.environ.get("FLAG", "tn_rect")
FLAGS = [int(v) for v in os.environ.get("VALUES", "3").split(",")]
for name, hw, N, K, ps in CASES:
    M = B * hw
    X = torch.randn((M, N), device="cuda", dtype=dt)
    Y = torch.randn((M, K), device="cuda", dtype=dt)
    line = f"{name:32s} M={M:8d} N={N:5d} K={K:5d}"
    for fl in FLAGS:
        if L.lib().vsx_set_flag(FLAG.encode(), fl) != 0:
            raise SystemExit(f"unknown flag {FLAG}")
        out = torch.zeros((B if ps else 1, N, K), device="cuda")
        cs = torch.zeros((B if ps else 1, N), device="cuda")
        ts = []
        for it in range(REP + 2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.gemm("tn", Y, X, out, M, N, K, K, N, K, dtype=dt, hw=hw, colsum=cs, b_bstride=N * K if ps else 0)
            e1.record()
            torch.cuda.synchronize()
            if it >= 2:
                ts.append(e0.elapsed_time(e1) * 1e3)
        us = sorted(ts)[len(ts) // 2]
        gb = (M * (N + K) * 2) / 1e9
        line += f" | {FLAG}={fl}: {us:8.1f} us {gb / (us * 1e-6):7.0f} GB/s"
    print(line, flush=True)
    del X, Y
