// Weight-space helpers shared by their single-op entry points (optim.hip, mlp.hip) and by the task-list kernel
// vsx_weight_tasks (optim.hip): each body does the work of ONE 256-thread workgroup `blk` of the single-op launch.
// The step runs ~200 of these per weight refresh, 4 - 7 us each with the chip idle behind every one: a task list turns
// them into a handful of launches.
#pragma once
#include "vsx_common.h"

// tap order of the GEMM K axis: k = t_dst * Cs + c.  tapmode 0: t_dst = t_src; tapmode 1 (head
// Conv3d [.., kz, ky, kx] → (ky, kx, kz)): t_src = (kz*3 + ky)*3 + kx, t_dst = (ky*3 + kx)*3 + kz.
__device__ __forceinline__ int tap_dst(int t_src, int tapmode) {
  if (tapmode == 0) return t_src;
  int kx = t_src % 3, ky = (t_src / 3) % 3, kz = t_src / 9;
  return (ky * 3 + kx) * 3 + kz;
}

template <typename T>
__device__ __forceinline__ void wt_prep_weight(int blk, const float* __restrict__ src, T* __restrict__ dst,
                                               T* __restrict__ dstT, const float* __restrict__ gamma, int R, int Cs, int Tn,
                                               int tapmode) {
  const long total = (long)R * Cs * Tn;
  const long gid = (long)blk * 256 + threadIdx.x;
  if (gid >= total) return;
  const int t = (int)(gid % Tn);
  const long rc = gid / Tn;
  const int c = (int)(rc % Cs);
  const int r = (int)(rc / Cs);
  float v = src[gid];
  if (gamma) v *= gamma[c];
  const int K = Tn * Cs;
  const int k = tap_dst(t, tapmode) * Cs + c;
  const T o = from_f32<T>(v);
  if (dst) dst[(size_t)r * K + k] = o;
  if (dstT) dstT[(size_t)k * R + r] = o;
}

// out[r] = (b ? b[r] : 0) + Σ_c W[r][c] * v[c]          (fold LN beta into the fc1 bias); 4 rows per workgroup
__device__ __forceinline__ void wt_matvec(int blk, const float* __restrict__ W, const float* __restrict__ v,
                                          const float* __restrict__ b, float* __restrict__ out, int R, int C) {
  const int r = blk * 4 + (threadIdx.x >> 6);
  float a = 0.f;
  if (r < R)
    for (int c = threadIdx.x & 63; c < C; c += 64) a += W[(size_t)r * C + c] * v[c];
  a = wave_sum(a);
  if (r < R && (threadIdx.x & 63) == 0) out[r] = a + (b ? b[r] : 0.f);
}

// dst[j][i] (+)= src[i][j]   (fp32; depthwise weights [C][49] <-> [49][C])
__device__ __forceinline__ void wt_transpose_f32(int blk, const float* __restrict__ src, float* __restrict__ dst, int A, int Bn,
                                                 int accumulate) {
  const long gid = (long)blk * 256 + threadIdx.x;
  if (gid >= (long)A * Bn) return;
  const int j = (int)(gid % Bn), i = (int)(gid / Bn);
  const float v = src[gid];
  float* d = dst + (size_t)j * A + i;
  *d = accumulate ? *d + v : v;
}

// fragment-major LDS image of the fused GRN-MLP kernels (csrc/mlp.hip):
// piece (hs, hf, kk) lane (p, q): W1[hs*32 + hf*16 + p][kk*32 + q*8 .. +7]
// piece (hs, nf)     lane (p, q): W2[nf*16 + p][hs*32 + q*4 .. +3], W2[nf*16 + p][hs*32 + 16 + q*4 .. +3]
__device__ __forceinline__ void wt_mlp_pack(int blk, const bf16_t* __restrict__ W1, const bf16_t* __restrict__ W2,
                                            char* __restrict__ img, int C) {
  const int H4 = 4 * C, KK = C / 32, NF = C / 16, PPS = 2 * KK + NF;
  const long gid = (long)blk * 256 + threadIdx.x;
  const long total = (long)(H4 / 32) * PPS * 64;
  if (gid >= total) return;
  const int lane = (int)(gid & 63);
  const long piece = gid >> 6;
  const int hs = (int)(piece / PPS), pp = (int)(piece % PPS);
  const int p16 = lane & 15, kq = lane >> 4;
  uint4 v;
  if (pp < 2 * KK) {
    const int hf = pp / KK, kk = pp % KK;
    v = *reinterpret_cast<const uint4*>(W1 + (size_t)(hs * 32 + hf * 16 + p16) * C + kk * 32 + kq * 8);
  } else {
    const int nf = pp - 2 * KK;
    const bf16_t* r = W2 + (size_t)(nf * 16 + p16) * H4 + hs * 32 + kq * 4;
    const uint2 lo = *reinterpret_cast<const uint2*>(r), hi = *reinterpret_cast<const uint2*>(r + 16);
    v = make_uint4(lo.x, lo.y, hi.x, hi.y);
  }
  *reinterpret_cast<uint4*>(img + (size_t)gid * 16) = v;
}
__host__ __device__ inline long wt_mlp_pack_items(int C) { return (long)(4 * C / 32) * (2 * (C / 32) + C / 16) * 64; }

// gradient back-mapping: dparam[r][c][t] += g[r][tap_dst(t)*Cs + c] * (gamma ? gamma[c] : 1) + (u ? u[r]*beta[c] : 0);
// dgamma[c] += Σ_{r,t} g[r][k] * W[r][c][t]
// (u, beta): the folded bias b' = b + W·beta also depends on W  →  dW += u ⊗ beta with u = db'
// thread = one (c, t) column; a workgroup loops a chunk of rows so dgamma needs one atomic per thread.
// Grid: wt_unprep_bx(Cs, Tn) column blocks x cdiv(R, wt_unprep_rpb(R)) row chunks, flattened (x fastest).
__host__ __device__ inline int wt_unprep_rpb(int R) { const int r = (R + 63) / 64; return r < 8 ? 8 : r; }
__host__ __device__ inline int wt_unprep_bx(int Cs, int Tn) { return (Cs * Tn + 255) / 256; }
__device__ __forceinline__ void wt_unprep_grad(int blk, const float* __restrict__ g, float* __restrict__ dparam,
                                               const float* __restrict__ gamma, const float* __restrict__ W,
                                               float* __restrict__ dgamma, const float* __restrict__ u,
                                               const float* __restrict__ beta, const float* __restrict__ rowsub, int R, int Cs,
                                               int Tn, int tapmode) {
  const int nbx = wt_unprep_bx(Cs, Tn), rows_per_block = wt_unprep_rpb(R);
  const int col = (blk % nbx) * 256 + threadIdx.x;
  if (col >= Cs * Tn) return;
  const int t = col % Tn, c = col / Tn;
  const int K = Tn * Cs;
  const int k = tap_dst(t, tapmode) * Cs + c;
  const float gm = gamma ? gamma[c] : 1.f;
  const float bt = u ? beta[c] : 0.f;
  const int r0 = (blk / nbx) * rows_per_block;
  const int r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
  float dg = 0.f;
  for (int r = r0; r < r1; ++r) {
    const float gv = g[(size_t)r * K + k] - (rowsub ? rowsub[r] : 0.f);  // rowsub: the rank-1 term of a GEMM run on un-centred rows
    const size_t pi = ((size_t)r * Cs + c) * Tn + t;
    dparam[pi] += gv * gm + (u ? u[r] * bt : 0.f);
    if (dgamma) dg += gv * W[pi];
  }
  if (dgamma) atomicAdd(dgamma + c, dg);
}

// out[c] += Σ_r W[r][c] * u[r]                            (gradient of the folded LN beta)
__host__ __device__ inline int wt_matvec_t_rpb(int R) { const int r = (R + 31) / 32; return r < 16 ? 16 : r; }
__device__ __forceinline__ void wt_matvec_t(int blk, const float* __restrict__ W, const float* __restrict__ u,
                                            float* __restrict__ out, int R, int C) {
  const int nbx = (C + 255) / 256, rows_per_block = wt_matvec_t_rpb(R);
  const int c = (blk % nbx) * 256 + threadIdx.x;
  if (c >= C) return;
  const int r0 = (blk / nbx) * rows_per_block;
  const int r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
  float a = 0.f;
  for (int r = r0; r < r1; ++r) a += W[(size_t)r * C + c] * u[r];
  atomicAdd(out + c, a);
}

// out[n] += Σ_r ws[r][n].  Workgroup = 64 columns x 4 row slots over a 64-row slab; slots are combined in LDS, slabs with
// <= R/64 atomics per address.  Grid: cdiv(N, 64) column blocks x cdiv(R, 64) slabs, flattened (x fastest).
__device__ __forceinline__ void wt_reduce_rows(int blk, const float* __restrict__ ws, float* __restrict__ out, int R, int N) {
  __shared__ float red[4][64];
  const int nbx = (N + 63) / 64;
  const int nl = threadIdx.x & 63, slot = threadIdx.x >> 6;
  const int n = (blk % nbx) * 64 + nl;
  const int r0 = (blk / nbx) * 64;
  const int r1 = r0 + 64 < R ? r0 + 64 : R;
  float a = 0.f;
  if (n < N) {
#pragma unroll 4
    for (int r = r0 + slot; r < r1; r += 4) a += ws[(size_t)r * N + n];
  }
  red[slot][nl] = a;
  __syncthreads();
  if (slot == 0 && n < N) atomicAdd(out + n, red[0][nl] + red[1][nl] + red[2][nl] + red[3][nl]);
}
