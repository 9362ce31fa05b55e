"""Vendor-GEMM yardstick: `torch.matmul` (hipBLASLt / rocBLAS, bf16 operands, fp32 accumulation) on the
bench step's largest NT and TN shapes at B = 512, next to the hand-written kernels on the SAME buffers.

Tool only: nothing under viscy_amd/ calls a vendor GEMM.  It answers one question — how much of the distance between the
hand-written kernels and the MFMA / HBM roofs is the SHAPE (skinny K, wide outputs) and how much is the KERNEL.

    python tools/gemm_yardstick.py [--out gpurun_out/r05_gemm_yardstick.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viscy_amd import _lib as L  # noqa: E402
from viscy_amd import ops  # noqa: E402

dt, dev = torch.bfloat16, "cuda"


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/r05_gemm_yardstick.json")
    ap.add_argument("--batch", type=int, default=512)
    a = ap.parse_args()
    B = a.batch
    rows = []
    # (name, hw, C): the ConvNeXt-V2 block shapes of convnextv2_tiny at 256 x 256 (stage, pixels per sample, width)
    stages = [("s0", 4096, 96), ("s1", 1024, 192), ("s2", 256, 384), ("s3", 64, 768), ("d2", 4096, 224)]
    for name, hw, C in stages:
        M, N4 = B * hw, 4 * C
        x = torch.randn(M, C, device=dev).to(dt)
        W1 = torch.randn(N4, C, device=dev).to(dt)
        W2 = torch.randn(C, N4, device=dev).to(dt)
        h = torch.randn(M, N4, device=dev).to(dt)
        out4 = torch.empty(M, N4, device=dev, dtype=dt)
        out1 = torch.empty(M, C, device=dev, dtype=dt)
        # ---- NT, wide output (fc1 / dz): [M, C] x [4C, C]^T
        fl = 2.0 * M * N4 * C
        by = (M * C + M * N4 + N4 * C) * 2.0
        us_v = timeit(lambda: torch.matmul(x, W1.t(), out=out4))
        us_h = timeit(lambda: ops.gemm("nt", x, W1, out4, M, N4, C, C, C, N4, dtype=dt))
        rows.append(dict(kind="nt", name=f"{name} fc1-shaped", M=M, N=N4, K=C, vendor_us=us_v, hand_us=us_h, hand_kernel=L.lib().vsx_last_kernel().decode(),
                         gflop=fl / 1e9, gbytes=by / 1e9))
        # ---- NT, K-heavy (fc2 / dx): [M, 4C] x [C, 4C]^T
        by = (M * N4 + M * C + N4 * C) * 2.0
        us_v = timeit(lambda: torch.matmul(h, W2.t(), out=out1))
        us_h = timeit(lambda: ops.gemm("nt", h, W2, out1, M, C, N4, N4, N4, C, dtype=dt))
        rows.append(dict(kind="nt", name=f"{name} fc2-shaped", M=M, N=C, K=N4, vendor_us=us_v, hand_us=us_h, hand_kernel=L.lib().vsx_last_kernel().decode(),
                         gflop=fl / 1e9, gbytes=by / 1e9))
        # ---- TN, whole batch (dW1): [M, 4C]^T x [M, C] -> [4C, C]
        by = (M * N4 + M * C) * 2.0
        dW = torch.zeros(N4, C, device=dev)
        dWv = torch.empty(N4, C, device=dev, dtype=dt)
        us_v = timeit(lambda: torch.matmul(h.t(), x, out=dWv))
        us_h = timeit(lambda: ops.gemm("tn", x, h, dW, M, N4, C, C, N4, C, dtype=dt))
        rows.append(dict(kind="tn", name=f"{name} dW1", M=M, N=N4, K=C, vendor_us=us_v, hand_us=us_h, hand_kernel=L.lib().vsx_last_kernel().decode(),
                         gflop=fl / 1e9, gbytes=by / 1e9))
        # ---- TN, per sample (Q_b = dout_b^T g_b): B x ([hw, C]^T x [hw, 4C]) -> [B, C, 4C]
        if B * C * N4 * 4 <= (512 << 20):
            Qb = torch.empty(B, C, N4, device=dev)
            Qv = torch.empty(B, C, N4, device=dev, dtype=dt)
            cs = torch.zeros(B, C, device=dev)
            x3, h3 = x.view(B, hw, C), h.view(B, hw, N4)
            us_v = timeit(lambda: torch.bmm(x3.transpose(1, 2), h3, out=Qv))
            us_h = timeit(lambda: ops.gemm("tn", h, x, Qb, M, C, N4, N4, C, N4, dtype=dt, hw=hw, colsum=cs, b_bstride=C * N4))
            rows.append(dict(kind="tn_ps", name=f"{name} per-sample dout^T g", M=M, N=C, K=N4, vendor_us=us_v, hand_us=us_h,
                             hand_kernel=L.lib().vsx_last_kernel().decode(), gflop=fl / 1e9, gbytes=by / 1e9))
            del Qb, Qv
        del x, W1, W2, h, out4, out1
        torch.cuda.empty_cache()
    for r in rows:
        for who in ("vendor", "hand"):
            us = r[f"{who}_us"]
            r[f"{who}_tflops"] = round(r["gflop"] / us * 1e3, 1)  # GFLOP / us = 1000 TFLOP/s
            r[f"{who}_TBps"] = round(r["gbytes"] / us * 1e3, 3)
        r["hand_over_vendor"] = round(r["vendor_us"] / r["hand_us"], 3)
        print(f"{r['kind']:5s} {r['name']:28s} M={r['M']:8d} N={r['N']:5d} K={r['K']:5d} | vendor {r['vendor_us']:8.1f} us "
              f"{r['vendor_tflops']:7.1f} TF {r['vendor_TBps']:6.2f} TB/s | hand {r['hand_us']:8.1f} us {r['hand_tflops']:7.1f} TF "
              f"{r['hand_TBps']:6.2f} TB/s ({r['hand_kernel']}) | speed ratio hand/vendor {r['hand_over_vendor']:.2f}", flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump({"batch": B, "dtype": "bf16", "device": torch.cuda.get_device_name(0), "torch": torch.__version__,
                   "note": "median of 10 launches, HIP events; vendor = torch.matmul / torch.bmm (hipBLASLt / rocBLAS), bf16 output; "
                           "hand = viscy_amd.ops.gemm with no epilogue on the same operands (TN writes fp32)", "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
