// Channel LayerNorm, GRN statistics (K7) and the GRN/GELU backward pass.
// Rows are pixels of a channels-last tensor; a row's C channels are contiguous, so a row is
// reduced by an aligned group of G lanes of one wave64 with shuffle reductions (no LDS, no
// barriers); loads/stores are 16-byte vectors.
#include "vsx_common.h"
#include "wtasks.h"
#include "../../include/vsx.h"
extern int g_vsx_grn_stream;
extern int g_vsx_ggb_contig;
extern int g_vsx_ln_stream;

// ------------------------------------------------------------------ LayerNorm forward
template <typename T, int G, int CPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     int rows, int C, float eps, int iters, int nt) {
  // A lane group owns R rows per block and requests all of them before the first reduction: one row per group
  // (first version) meant one 16-byte load in flight per lane and 6 KB blocks -> 2.2 TB/s.
  constexpr int VN = VT<T>::N;
  constexpr int R = CPL == 1 ? 4 : (CPL == 2 ? 2 : 1);
  constexpr int RPI = 256 / G;  // rows per sweep of the block
  const int nch = C / VN;
  const int gl = threadIdx.x % G;
  float gam[CPL][VN], bet[CPL][VN];  // this lane's channels never change: the affine parameters live in registers
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = gl + i * G;
#pragma unroll
    for (int j = 0; j < VN; ++j) {
      gam[i][j] = (gamma && c < nch) ? gamma[c * VN + j] : 1.f;
      bet[i][j] = (gamma && c < nch) ? beta[c * VN + j] : 0.f;
    }
  }
  // a bounded grid sweeps the rows in interleaved windows (like ln_bwd, which streamed 1.9x faster than one short-lived
  // block per 64 rows): sweep `it` of the grid covers rows [it * gridDim.x * RPI * R, ...)
  for (int it = 0; it < iters; ++it) {
  const int row0 = (it * gridDim.x + blockIdx.x) * (RPI * R) + threadIdx.x / G;
  float v[R][CPL][VN];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = row0 + r * RPI;
    const T* xr = x + (size_t)(row < rows ? row : rows - 1) * C;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = gl + i * G;
      if (c < nch) {
        unpack<T>(ldvec_stream<T>(xr + c * VN, nt != 0), v[r][i]);
      } else {
#pragma unroll
        for (int j = 0; j < VN; ++j) v[r][i][j] = 0.f;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = row0 + r * RPI;
    // one reduction round: Σ(x - x0) and Σ(x - x0)² reduced together (two independent shuffle chains instead of two
    // dependent rounds); the shift x0 = first element of the row keeps the single-pass variance well conditioned
    const float x0 = __shfl(v[r][0][0], (threadIdx.x & 63) / G * G, 64);
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      if (gl + i * G < nch) {
#pragma unroll
        for (int j = 0; j < VN; ++j) {
          const float d = v[r][i][j] - x0;
          s += d;
          q = fmaf(d, d, q);
        }
      }
    }
    s = group_sum<G>(s) / (float)C;
    q = group_sum<G>(q) / (float)C;
    const float mean = x0 + s;
    const float rstd = rsqrtf(fmaxf(q - s * s, 0.f) + eps);
    if (row < rows) {  // (uniform per group; the shuffles above ran for every lane)
      T* yr = y + (size_t)row * C;
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int c = gl + i * G;
        if (c < nch) {
          float o[VN];
#pragma unroll
          for (int j = 0; j < VN; ++j) {
            const float xh = (v[r][i][j] - mean) * rstd;
            o[j] = gamma ? fmaf(xh, gam[i][j], bet[i][j]) : xh;
          }
          stvec<T>(yr + c * VN, pack<T>(o));
        }
      }
#ifndef LN_NO_RSTD
      if (gl == 0) {
        if (mean_out) mean_out[row] = mean;
        rstd_out[row] = rstd;
      }
#endif
    }
  }
  }  // sweeps
}

// ------------------------------------------------------------------ LayerNorm backward
// xhat = mean ? (x - mean) * rstd : x ;  g = dy * gamma ;  dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)) [+ add]
// AFF = false: no gamma and no dgamma / dbeta (the block LayerNorms, affine folded into fc1): the 3 x CPL x VN registers of the
// affine accumulators are not allocated (98 -> 70 registers at CPL = 1: 4 -> 7 waves per SIMD)
template <typename T, int G, int CPL, bool AFF>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const T* __restrict__ add,
                                                     T* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int rows, int C, int iters, int nt) {
  constexpr int VN = VT<T>::N;
  const bool g_nt = nt != 0;  // dy and x have no later reader: stream them past the caches
  extern __shared__ float red[];  // [2][C] when dgamma != nullptr
  const int nch = C / VN;
  const int gl = threadIdx.x % G;
  const int rslot = threadIdx.x / G;
  constexpr int RPI = 256 / G;
  if (AFF && dgamma) {
    for (int i = threadIdx.x; i < 2 * C; i += 256) red[i] = 0.f;
    __syncthreads();
  }
  constexpr int CA = AFF ? CPL : 1, VA = AFF ? VN : 1;
  float dg[CA][VA], db[CA][VA], gam[CA][VA];
  if constexpr (AFF) {
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      int c = gl + i * G;
#pragma unroll
      for (int j = 0; j < VN; ++j) {
        dg[i][j] = 0.f;
        db[i][j] = 0.f;
        gam[i][j] = (gamma && c < nch) ? gamma[c * VN + j] : 1.f;
      }
    }
  }
  typedef typename VT<T>::vec vec;
  constexpr int R = CPL == 1 ? 2 : 1;  // rows requested together per lane group (loads of both rows in flight)
  for (int it0 = 0; it0 < iters; it0 += R) {
    vec xr[R][CPL], dr[R][CPL], ar[R][CPL];
    int rowr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = ((it0 + r) * gridDim.x + blockIdx.x) * RPI + rslot;  // interleaved: the grid sweeps a contiguous window
      rowr[r] = (it0 + r < iters && row < rows) ? row : -1;
      const size_t ro = (size_t)(rowr[r] >= 0 ? row : 0) * C;
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int c = gl + i * G;
        if (c < nch) {
          xr[r][i] = ldvec_stream<T>(x + ro + c * VN, g_nt);
          dr[r][i] = ldvec_stream<T>(dy + ro + c * VN, g_nt);
          if (add) ar[r][i] = ldvec<T>(add + ro + c * VN);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = rowr[r];
      const bool live = row >= 0;  // uniform per lane group; dead groups still take part in the shuffles
      const float mu = (mean && live) ? mean[row] : 0.f;
      const float rs = live ? rstd[row] : 0.f;
      float xh[CPL][VN], g[CPL][VN];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int c = gl + i * G;
        if (c < nch && live) {
          float xv[VN], dv[VN];
          unpack<T>(xr[r][i], xv);
          unpack<T>(dr[r][i], dv);
#pragma unroll
          for (int j = 0; j < VN; ++j) {
            xh[i][j] = mean ? (xv[j] - mu) * rs : xv[j];
            if constexpr (AFF) {
              g[i][j] = dv[j] * gam[i][j];
              dg[i][j] += dv[j] * xh[i][j];
              db[i][j] += dv[j];
            } else {
              g[i][j] = dv[j];
            }
            s1 += g[i][j];
            s2 += g[i][j] * xh[i][j];
          }
        } else {
#pragma unroll
          for (int j = 0; j < VN; ++j) { xh[i][j] = 0.f; g[i][j] = 0.f; }
        }
      }
      s1 = group_sum<G>(s1) / (float)C;
      s2 = group_sum<G>(s2) / (float)C;
      if (live) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
          const int c = gl + i * G;
          if (c < nch) {
            float o[VN];
#pragma unroll
            for (int j = 0; j < VN; ++j) o[j] = rs * (g[i][j] - s1 - xh[i][j] * s2);
            if (add) {
              float a[VN];
              unpack<T>(ar[r][i], a);
#pragma unroll
              for (int j = 0; j < VN; ++j) o[j] += a[j];
            }
            stvec<T>(dx + (size_t)row * C + c * VN, pack<T>(o));
          }
        }
      }
    }
  }
  if constexpr (AFF) if (dgamma) {
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      int c = gl + i * G;
      if (c < nch) {
#pragma unroll
        for (int j = 0; j < VN; ++j) {
          atomicAdd(&red[c * VN + j], dg[i][j]);
          atomicAdd(&red[C + c * VN + j], db[i][j]);
        }
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256) {
      atomicAdd(dgamma + i, red[i]);
      atomicAdd(dbeta + i, red[C + i]);
    }
  }
}

extern int g_vsx_ln_fblk;
extern int g_vsx_ln_bblk;
extern int g_vsx_ln_ablk;
extern int g_vsx_ln_pack;
template <typename T, int G, int CPL>
static int ln_launch(bool fwd, const void* a0, const void* a1, void* out, float* mean, float* rstd,
                     const float* gamma, const float* beta, const void* add, float* dgamma, float* dbeta, int rows,
                     int C, float eps, hipStream_t s) {
  constexpr int RPI = 256 / G;
  if (fwd) {
    constexpr int R = CPL == 1 ? 4 : (CPL == 2 ? 2 : 1);  // rows per lane group (see the kernel)
    // <= ln_fblk workgroups, each sweeping `iters` windows.  With an affine every workgroup first fetches its gamma / beta
    // vectors — a memory round trip before its first row — so short-lived workgroups pay it once per 64 rows: an eighth of the
    // plain cap (C = 96 / 192 / 384 at B = 512: 258 -> 183, 115 -> 89, 65 -> 55 us; the plain pass is best at 8 192 .. 32 768)
    const int fcap = gamma ? (g_vsx_ln_fblk >= 8 ? g_vsx_ln_fblk / 8 : 1) : g_vsx_ln_fblk;
    int iters = vsx_cdiv(rows, RPI * R * fcap);
    if (iters < 1) iters = 1;
    hipLaunchKernelGGL((ln_fwd_kernel<T, G, CPL>), dim3(vsx_cdiv(rows, RPI * R * iters)), dim3(256), 0, s, (const T*)a0,
                       (T*)out, mean, rstd, gamma, beta, rows, C, eps, iters, g_vsx_ln_stream & 2);
  } else {
    // with an affine LayerNorm: <= 512 blocks → <= 512 same-address atomics on dgamma / dbeta; the block LayerNorms have no
    // affine here (folded into fc1) and no such limit — their cap is the flag ln_bblk
    // affine: every workgroup ends with one atomic per dgamma / dbeta element, and same-address atomics retire at ~40 ns each
    // (tools/perf_ln.py: 8 192 workgroups at C = 768 take 339 us for a 33 us stream), so the workgroup count is what the pass
    // can stream meanwhile: bytes / (4.5 TB/s x 80 ns), 256 .. 4 096 (C = 96 / 192 / 384 at B = 512: 336 -> 262, 171 -> 132,
    // 91 -> 81 us against the fixed 512 of rounds 1 - 5); ln_ablk != 0 forces a count
    int acap = g_vsx_ln_ablk;
    if (dgamma && acap <= 0) {
      const long cap = 3L * rows * C * (long)sizeof(T) / 360000L;
      acap = (int)(cap < 256 ? 256 : (cap > 4096 ? 4096 : cap));
    }
    int iters = vsx_cdiv(rows, RPI * (dgamma ? acap : g_vsx_ln_bblk));
    if (iters < 1) iters = 1;
    int grid = vsx_cdiv(rows, RPI * iters);
    size_t sh = dgamma ? 2 * (size_t)C * sizeof(float) : 0;
    if (gamma || dgamma)
      hipLaunchKernelGGL((ln_bwd_kernel<T, G, CPL, true>), dim3(grid), dim3(256), sh, s, (const T*)a0, (const T*)a1,
                         (const float*)mean, (const float*)rstd, gamma, (const T*)add, (T*)out, dgamma, dbeta, rows, C,
                         iters, g_vsx_ln_stream & 1);
    else
      hipLaunchKernelGGL((ln_bwd_kernel<T, G, CPL, false>), dim3(grid), dim3(256), sh, s, (const T*)a0, (const T*)a1,
                         (const float*)mean, (const float*)rstd, gamma, (const T*)add, (T*)out, dgamma, dbeta, rows, C,
                         iters, g_vsx_ln_stream & 1);
  }
  VSX_LAUNCH_CHECK();
  return 0;
}

template <typename T>
static int ln_dispatch(bool fwd, const void* a0, const void* a1, void* out, float* mean, float* rstd,
                       const float* gamma, const float* beta, const void* add, float* dgamma, float* dbeta, int rows,
                       int C, float eps, hipStream_t s) {
  const int nch = C / VT<T>::N;
#define LN_CASE(G, CPL) return ln_launch<T, G, CPL>(fwd, a0, a1, out, mean, rstd, gamma, beta, add, dgamma, dbeta, rows, C, eps, s)
  // ln_pack: rows of 3 x 2^k vectors on groups of 2^k lanes with three vectors each — every lane loads, where the next power of
  // two leaves a quarter of the lanes idle
  // (backward only, 24 and 48 vectors = C = 192 / 384 in bf16: 113 -> 94, 58 -> 48 us, with affine gradients 186 -> 129, 91 -> 85;
  // the forward and the 12- / 96-vector rows are slower packed: tools/perf_ln.py pack)
  if (g_vsx_ln_pack && !fwd) {
    if (nch == 24) LN_CASE(8, 3);
    if (nch == 48) LN_CASE(16, 3);
  }
  if (nch <= 4) LN_CASE(4, 1);
  if (nch <= 8) LN_CASE(8, 1);
  if (nch <= 16) LN_CASE(16, 1);
  if (nch <= 32) LN_CASE(32, 1);
  if (nch <= 64) LN_CASE(64, 1);
  if (nch <= 128) LN_CASE(64, 2);
  if (nch <= 192) LN_CASE(64, 3);
  if (nch <= 256) LN_CASE(64, 4);
#undef LN_CASE
  vsx_set_error("layernorm: C=%d too wide (max %d)", C, 256 * VT<T>::N);
  return 1;
}

/* K2/K4: timm LayerNorm2d / nn.LayerNorm over channels, eps 1e-6 (called at unext2.py:79 through
 * timm ConvNeXtStage / ConvNeXtBlock).  gamma == NULL → no affine (the block LN's affine is folded
 * into fc1 on the host side, see viscy_amd/unext2.py). */
extern "C" int32_t vsx_ln_fwd(const void* x, void* y, float* mean, float* rstd, const float* gamma, const float* beta,
                              int32_t rows, int32_t C, float eps, int32_t dtype, vsx_stream_t stream) {
  int vn = dtype == VSX_BF16 ? 8 : 4;
  VSX_CHECK(rows > 0 && C > 0 && C % vn == 0, "vsx_ln_fwd: rows=%d C=%d (C must be a multiple of %d)", rows, C, vn);
  VSX_CHECK(x && y && rstd, "vsx_ln_fwd: null pointer");
  VSX_CHECK((gamma == nullptr) == (beta == nullptr), "vsx_ln_fwd: gamma and beta must both be set or both NULL");
  hipStream_t s = (hipStream_t)stream;
  return dtype == VSX_BF16
             ? ln_dispatch<bf16_t>(true, x, nullptr, y, mean, rstd, gamma, beta, nullptr, nullptr, nullptr, rows, C, eps, s)
             : ln_dispatch<float>(true, x, nullptr, y, mean, rstd, gamma, beta, nullptr, nullptr, nullptr, rows, C, eps, s);
}

extern "C" int32_t vsx_ln_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                              const void* add, void* dx, float* dgamma, float* dbeta, int32_t rows, int32_t C,
                              int32_t dtype, vsx_stream_t stream) {
  int vn = dtype == VSX_BF16 ? 8 : 4;
  VSX_CHECK(rows > 0 && C > 0 && C % vn == 0, "vsx_ln_bwd: rows=%d C=%d (C must be a multiple of %d)", rows, C, vn);
  VSX_CHECK(dy && x && rstd && dx, "vsx_ln_bwd: null pointer");
  VSX_CHECK((dgamma == nullptr) == (dbeta == nullptr), "vsx_ln_bwd: dgamma and dbeta must both be set or both NULL");
  hipStream_t s = (hipStream_t)stream;
  return dtype == VSX_BF16 ? ln_dispatch<bf16_t>(false, dy, x, dx, (float*)mean, (float*)rstd, gamma, nullptr, add,
                                                 dgamma, dbeta, rows, C, 0.f, s)
                           : ln_dispatch<float>(false, dy, x, dx, (float*)mean, (float*)rstd, gamma, nullptr, add,
                                                dgamma, dbeta, rows, C, 0.f, s);
}

// ------------------------------------------------------------------ GRN statistics (fp32, [B, N])
// s[b,n] = 1 + gamma[n] * g / (mean_n g + eps),  g = sqrt(colsq[b,n])
__global__ __launch_bounds__(256) void grn_scale_kernel(const float* __restrict__ colsq, const float* __restrict__ gamma,
                                                        float* __restrict__ s, int N, float eps) {
  __shared__ float part[4];
  const int b = blockIdx.x;
  float acc = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) acc += sqrtf(colsq[(size_t)b * N + n]);
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  const float m = (part[0] + part[1] + part[2] + part[3]) / (float)N;
  for (int n = threadIdx.x; n < N; n += 256)
    s[(size_t)b * N + n] = 1.f + gamma[n] * sqrtf(colsq[(size_t)b * N + n]) / (m + eps);
}

// backward of the statistics path: P[b,n] = sum_hw dz * g_act ;  t[b,n] multiplies g_act in dG = dz*s + g_act*t
// One workgroup per sample writes t[b, :] and its dgamma contribution into a workspace row; dgamma / dbeta are then column
// sums over the samples through reduce_rows_kernel (<= nb / 64 atomics per address).  The first version added nb
// same-address atomics per channel straight onto dgamma / dbeta: 38 us per launch at nb = 512.
__global__ __launch_bounds__(256) void grn_bwd_stats_kernel(const float* __restrict__ colsq, const float* __restrict__ P,
                                                            const float* __restrict__ gamma, float* __restrict__ t,
                                                            float* __restrict__ wsg, int N, float eps) {
  __shared__ float part[2][4];
  const int b = blockIdx.x;
  float a0 = 0.f, a1 = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) {
    float g = sqrtf(colsq[(size_t)b * N + n]);
    a0 += g;
    a1 += gamma[n] * P[(size_t)b * N + n] * g;
  }
  a0 = wave_sum(a0);
  a1 = wave_sum(a1);
  if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = a0; part[1][threadIdx.x >> 6] = a1; }
  __syncthreads();
  const float m = (part[0][0] + part[0][1] + part[0][2] + part[0][3]) / (float)N;
  const float sdg = part[1][0] + part[1][1] + part[1][2] + part[1][3];
  const float inv = 1.f / (m + eps);
  for (int n = threadIdx.x; n < N; n += 256) {
    float g = sqrtf(colsq[(size_t)b * N + n]);
    float pn = P[(size_t)b * N + n];
    float dn = gamma[n] * pn;
    float dgv = dn * inv - sdg * inv * inv / (float)N;
    t[(size_t)b * N + n] = g > 0.f ? dgv / g : 0.f;
    wsg[(size_t)b * N + n] = pn * g * inv;
  }
}
__global__ void reduce_rows_kernel(const float* __restrict__ ws, float* __restrict__ out, int R, int N);

// det_reduce: a workgroup owns (one group, 64 columns); its 4 row slots add every fourth row of the group in ascending order and
// the four partial sums are combined in slot order — a fixed tree, the same bits in every run (1 024 rows per sample at 2048^2:
// one thread per column alone took milliseconds)
__global__ __launch_bounds__(256) void det_group_sum_kernel(const float* __restrict__ ws, int ld, int col0, float* __restrict__ out,
                                                            int groups, int rpg, int N) {
  __shared__ float part[4][64];
  const int g = blockIdx.y, cl = threadIdx.x & 63, slot = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + cl;
  float acc = 0.f;
  if (n < N) {
    const float* p = ws + (size_t)g * rpg * ld + col0 + n;
    for (int r = slot; r < rpg; r += 4) acc += p[(size_t)r * ld];
  }
  part[slot][cl] = acc;
  __syncthreads();
  if (slot == 0 && n < N) out[(size_t)g * N + n] += (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
}
int vsx_det_group_sum(const float* ws, int ld, int col0, float* out, int groups, int rows_per_group, int N, hipStream_t s) {
  hipLaunchKernelGGL(det_group_sum_kernel, dim3(vsx_cdiv(N, 64), groups), dim3(256), 0, s, ws, ld, col0, out, groups, rows_per_group, N);
  VSX_LAUNCH_CHECK();
  return 0;
}

/* K7: timm GlobalResponseNorm statistics.  colsq[b,n] = sum_hw gelu(h)^2 comes from the fc1 GEMM epilogue. */
extern "C" int32_t vsx_grn_scale(const float* colsq, const float* gamma, float* s, int32_t nb, int32_t N, float eps,
                                 vsx_stream_t stream) {
  VSX_CHECK(colsq && gamma && s && nb > 0 && N > 0, "vsx_grn_scale: bad arguments");
  hipLaunchKernelGGL(grn_scale_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, colsq, gamma, s, N, eps);
  VSX_LAUNCH_CHECK();
  return 0;
}
extern "C" int32_t vsx_grn_bwd_stats(const float* colsq, const float* P, const float* Sb, const float* gamma, float* t,
                                     float* dgamma, float* dbeta, float* rowst, int32_t nb, int32_t N, float eps,
                                     vsx_stream_t stream) {
  VSX_CHECK(colsq && P && gamma && t && rowst && nb > 0 && N > 0, "vsx_grn_bwd_stats: bad arguments");
  VSX_CHECK((Sb == nullptr) == (dbeta == nullptr), "vsx_grn_bwd_stats: Sb and dbeta come together");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(grn_bwd_stats_kernel, dim3(nb), dim3(256), 0, st, colsq, P, gamma, t, rowst, N, eps);
  if (dgamma == nullptr) {  // the caller folds rowst (and its Sb) into the gradients itself: VSX_WTASK_REDUCE_ROWS jobs
    VSX_CHECK(dbeta == nullptr, "vsx_grn_bwd_stats: dgamma == NULL (reductions left to the caller) needs dbeta == NULL");
    VSX_LAUNCH_CHECK();
    return 0;
  }
  dim3 rg(vsx_cdiv(N, 64), vsx_cdiv(nb, 64));
  hipLaunchKernelGGL(reduce_rows_kernel, rg, dim3(256), 0, st, (const float*)rowst, dgamma, nb, N);
  if (Sb) hipLaunchKernelGGL(reduce_rows_kernel, rg, dim3(256), 0, st, Sb, dbeta, nb, N);  // GRN beta gradient = sum_b sum_hw dz
  VSX_LAUNCH_CHECK();
  return 0;
}

extern int g_vsx_ggb_blocks;
// ------------------------------------------------------------------ GRN + GELU backward (pass 2)
// dh = (dz * s[b,n] + gelu(h) * t[b,n]) * gelu'(h), written over dz; colsum[n] += Σ_m dh
// Block = [256/tpr row slots][tpr column chunks]; every thread streams its column chunk down the
// rows it is dealt (16-byte loads, lanes along the contiguous channel axis, two rows in flight).
// Same-address global atomics serialise at ~0.2 us each on this chip, so the column sums go
// block → LDS → one workspace row per block → a second tiny kernel, never thousands of atomics
// onto one address.
template <typename T>
__global__ __launch_bounds__(256) void grn_gelu_bwd_kernel(T* __restrict__ dz, const T* __restrict__ h,
                                                           const float* __restrict__ s, const float* __restrict__ t,
                                                           float* __restrict__ ws, int M, int N, int hw, int tpr, int nt,
                                                           int rpb) {
  constexpr int VN = VT<T>::N;
  __shared__ float red[256 * VN];
  const int cl = threadIdx.x % tpr;
  const int cc = blockIdx.x * tpr + cl;
  const int slot = threadIdx.x / tpr, nslot = blockDim.x / tpr;
  const bool active = cc * VN < N;
  const int n = cc * VN;
  float cs[VN];
#pragma unroll
  for (int j = 0; j < VN; ++j) cs[j] = 0.f;
  if (active) {
    int bcur = -1;
    float sv[VN], tv[VN];
    // rpb > 0: the workgroup walks its own contiguous range of rpb rows (sequential pages, the sample index and with it
    // s / t change once per hw rows); rpb = 0: rows strided by the grid (the first version)
    const int mstep = rpb > 0 ? nslot : gridDim.y * nslot;
    const int mbeg = rpb > 0 ? blockIdx.y * rpb : blockIdx.y * nslot;
    const int mend = rpb > 0 ? (mbeg + rpb < M ? mbeg + rpb : M) : M;
    for (int m0 = mbeg + slot; m0 < mend; m0 += 2 * mstep) {
      const int m1 = m0 + mstep;
      const bool has1 = m1 < mend;
      typename VT<T>::vec d0 = ldvec<T>(dz + (size_t)m0 * N + n), h0 = ldvec_stream<T>(h + (size_t)m0 * N + n, (nt & 2) != 0);
      typename VT<T>::vec d1 = vzero<T>(), h1 = vzero<T>();
      if (has1) {
        d1 = ldvec<T>(dz + (size_t)m1 * N + n);
        h1 = ldvec_stream<T>(h + (size_t)m1 * N + n, (nt & 2) != 0);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u == 1 && !has1) break;
        const int m = u ? m1 : m0;
        const int b = m / hw;
        if (b != bcur) {
          bcur = b;
#pragma unroll
          for (int j = 0; j < VN; ++j) { sv[j] = s[(size_t)b * N + n + j]; tv[j] = t[(size_t)b * N + n + j]; }
        }
        float dv[VN], hv[VN], o[VN];
        unpack<T>(u ? d1 : d0, dv);
        unpack<T>(u ? h1 : h0, hv);
#pragma unroll
        for (int j = 0; j < VN; j += 2) {  // packed fp32 pairs (gelu_parts2)
          const vsx_v2f x = {hv[j], hv[j + 1]}, d2 = {dv[j], dv[j + 1]};
          const vsx_v2f s2 = {sv[j], sv[j + 1]}, t2 = {tv[j], tv[j + 1]};
          vsx_v2f cdf, pdf;
          gelu_parts2(x, cdf, pdf);
          const vsx_v2f gv = x * cdf, dgv = cdf + x * pdf;
          const vsx_v2f r = (d2 * s2 + gv * t2) * dgv;
          o[j] = round_to<T>(r.x);
          o[j + 1] = round_to<T>(r.y);
          cs[j] += o[j];
          cs[j + 1] += o[j + 1];
        }
        stvec_stream(dz + (size_t)m * N + n, pack<T>(o), (nt & 1) != 0);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < VN; ++j) red[(slot * tpr + cl) * VN + j] = cs[j];
  __syncthreads();
  if (slot == 0 && active) {
    float o[VN];
#pragma unroll
    for (int j = 0; j < VN; ++j) {
      float a = 0.f;
      for (int q = 0; q < nslot; ++q) a += red[(q * tpr + cl) * VN + j];
      o[j] = a;
    }
#pragma unroll
    for (int j = 0; j < VN; ++j) ws[(size_t)blockIdx.y * N + n + j] = o[j];
  }
}

// out[n] += Σ_r ws[r][n].  Block = 64 columns x 4 row slots over a 64-row slab (blockIdx.y); slots are
// combined in LDS, slabs with <= R/64 atomics per address.
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ ws, float* __restrict__ out, int R,
                                                          int N) {
  wt_reduce_rows(blockIdx.x + gridDim.x * blockIdx.y, ws, out, R, N);  // body: csrc/wtasks.h
}

extern "C" int32_t vsx_grn_gelu_bwd(void* dz, const void* h, const float* s, const float* t, float* colsum, float* ws,
                                    int32_t ws_rows, int32_t M, int32_t N, int32_t hw, int32_t dtype,
                                    vsx_stream_t stream) {
  int vn = dtype == VSX_BF16 ? 8 : 4;
  VSX_CHECK(dz && h && s && t && colsum && ws && ws_rows > 0 && M > 0 && N > 0 && hw > 0 && N % vn == 0,
            "vsx_grn_gelu_bwd: bad arguments");
  int ncc = N / vn;
  // column chunks per block = an exact divisor-like split of the row (no power-of-two padding: with tpr rounded up to 64 /
  // 128 / 256 a quarter of the lanes of every wave idled on the 4C = 384 / 768 / 1536 / 3072 rows)
  int gx = vsx_cdiv(ncc, 256);
  int tpr = vsx_cdiv(ncc, gx);
  int p2 = 1;
  while (p2 < ncc && p2 < 256) p2 <<= 1;
  if (ncc * 5 >= p2 * 4 && ncc <= 256) {  // >= 80 % of a power of two (4C = 896): full waves win (measured, tools/perf_ggb.py)
    gx = 1;
    tpr = p2;
  }
  int nslot = 256 / tpr;
  const int nthreads = tpr * nslot;
  int ngroups = vsx_cdiv(M, nslot);
  int gy = vsx_cdiv(g_vsx_ggb_blocks, gx);
  if (gy > vsx_cdiv(ngroups, 4)) gy = vsx_cdiv(ngroups, 4);  // >= 4 rows per thread
  if (gy > ws_rows) gy = ws_rows;
  if (gy < 1) gy = 1;
  int rpb = 0;
  if (g_vsx_ggb_contig) {
    rpb = vsx_cdiv(vsx_cdiv(M, gy), 2 * nslot) * 2 * nslot;
    gy = vsx_cdiv(M, rpb);
  }
  dim3 grid(gx, gy);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VSX_BF16)
    hipLaunchKernelGGL(grn_gelu_bwd_kernel<bf16_t>, grid, dim3(nthreads), 0, st, (bf16_t*)dz, (const bf16_t*)h, s, t, ws, M, N,
                       hw, tpr, g_vsx_grn_stream, rpb);
  else
    hipLaunchKernelGGL(grn_gelu_bwd_kernel<float>, grid, dim3(nthreads), 0, st, (float*)dz, (const float*)h, s, t, ws, M, N, hw,
                       tpr, g_vsx_grn_stream, rpb);
  VSX_LAUNCH_CHECK();
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(vsx_cdiv(N, 64), vsx_cdiv(gy, 64)), dim3(256), 0, st, ws, colsum, gy, N);
  VSX_LAUNCH_CHECK();
  return 0;
}

// GRN statistics and fc2 weight gradient from the per-sample products Q[b] = dout_b^T g_b — ONE pass over Q (round 2 read
// it twice: once per sample for P / S, once per channel for dW2: 1.3 TB/s of its own traffic, 2 ms per step).
// A workgroup owns (a group of QR_SB samples) x (a block of QR_CB channels) x (512 columns); a thread owns 4 columns:
//   per channel c it loads the QR_SB sample rows (16-byte loads, QR_SB in flight), accumulates
//     p[b]  += W2[c, j] * Q[b, c, j]        (-> P[b, j], atomics: one per channel block)
//     sm[b] += W2[c, j] * cs[b, c]          (-> S[b, j])
//     acc   += s[b, j] * Q[b, c, j]         (-> the group's partial of dW2[c, j], plain store into ws[group][c][j])
// and reduce_rows_kernel folds the groups' partials into dW2 (<= nb / (64 QR_SB) atomics per address).
// channel block per workgroup: round 5 measured (B = 512, us per launch at C = 96 / 192 / 224) 16-channel blocks 58 / 167 / 237, 32-channel
// blocks 50 / 111 / 156, 64-channel blocks 74 / 103 / 129 — the pass is bound by its P / S atomics (C / block adds per address), not
// by the workgroup count, until the blocks get too few: 32 below C = 192, 64 from there
constexpr int QR_SB = 8, QR_TH = 128;
template <typename T>
__global__ __launch_bounds__(QR_TH) void grn_q_reduce_kernel(const float* __restrict__ Q, const float* __restrict__ cs,
                                                             const T* __restrict__ W2, const float* __restrict__ s,
                                                             const float* __restrict__ beta, float* __restrict__ P,
                                                             float* __restrict__ S, float* __restrict__ ws,
                                                             float* __restrict__ db2, int nb, int C, int QR_CB) {
  const int N = 4 * C;  // a multiple of 4: every thread's 4 columns are all inside or all outside
  const int j = (blockIdx.x * QR_TH + threadIdx.x) * 4;
  const int g = blockIdx.y;
  const int b0 = g * QR_SB;
  const int c0 = blockIdx.z * QR_CB, c1 = min(C, c0 + QR_CB);
  const bool lead = blockIdx.x == 0 && threadIdx.x == 0;  // one thread per (group, channel block) owns the bias gradient
  if (j >= N && !lead) return;
  const bool live = j < N;
  const int jj = live ? j : 0;
  auto w4 = [&](int c) -> float4 {
    if constexpr (sizeof(T) == 2) {
      const uint2 u = *reinterpret_cast<const uint2*>(W2 + (size_t)c * N + jj);
      return make_float4(bf16_bits_to_f32(u.x & 0xffffu), bf16_bits_to_f32(u.x >> 16), bf16_bits_to_f32(u.y & 0xffffu),
                         bf16_bits_to_f32(u.y >> 16));
    } else {
      return *reinterpret_cast<const float4*>(W2 + (size_t)c * N + jj);
    }
  };
  float4 p[QR_SB], sm[QR_SB], sv[QR_SB];
#pragma unroll
  for (int u = 0; u < QR_SB; ++u) {
    p[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    sm[u] = p[u];
    const int b = b0 + u < nb ? b0 + u : nb - 1;
    sv[u] = *reinterpret_cast<const float4*>(s + (size_t)b * N + jj);
  }
  const float4 bt = *reinterpret_cast<const float4*>(beta + jj);
  for (int c = c0; c < c1; ++c) {
    float4 qv[QR_SB];
    float cc[QR_SB];
#pragma unroll
    for (int u = 0; u < QR_SB; ++u) {
      const int b = b0 + u < nb ? b0 + u : nb - 1;
      qv[u] = *reinterpret_cast<const float4*>(Q + ((size_t)b * C + c) * N + jj);
      cc[u] = b0 + u < nb ? cs[(size_t)b * C + c] : 0.f;
    }
    const float4 wv = w4(c);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float ct = 0.f;
#pragma unroll
    for (int u = 0; u < QR_SB; ++u) {
      if (b0 + u < nb) {
        p[u].x = fmaf(wv.x, qv[u].x, p[u].x); p[u].y = fmaf(wv.y, qv[u].y, p[u].y);
        p[u].z = fmaf(wv.z, qv[u].z, p[u].z); p[u].w = fmaf(wv.w, qv[u].w, p[u].w);
        sm[u].x = fmaf(wv.x, cc[u], sm[u].x); sm[u].y = fmaf(wv.y, cc[u], sm[u].y);
        sm[u].z = fmaf(wv.z, cc[u], sm[u].z); sm[u].w = fmaf(wv.w, cc[u], sm[u].w);
        acc.x = fmaf(sv[u].x, qv[u].x, acc.x); acc.y = fmaf(sv[u].y, qv[u].y, acc.y);
        acc.z = fmaf(sv[u].z, qv[u].z, acc.z); acc.w = fmaf(sv[u].w, qv[u].w, acc.w);
        ct += cc[u];
      }
    }
    if (live) {
      acc.x = fmaf(bt.x, ct, acc.x); acc.y = fmaf(bt.y, ct, acc.y); acc.z = fmaf(bt.z, ct, acc.z); acc.w = fmaf(bt.w, ct, acc.w);
      *reinterpret_cast<float4*>(ws + ((size_t)g * C + c) * N + j) = acc;
    }
    if (lead) atomicAdd(db2 + c, ct);
  }
  if (!live) return;
#pragma unroll
  for (int u = 0; u < QR_SB; ++u) {
    if (b0 + u < nb) {
      float* Pp = P + (size_t)(b0 + u) * N + j;
      float* Sp = S + (size_t)(b0 + u) * N + j;
      atomicAdd(Pp + 0, p[u].x); atomicAdd(Pp + 1, p[u].y); atomicAdd(Pp + 2, p[u].z); atomicAdd(Pp + 3, p[u].w);
      atomicAdd(Sp + 0, sm[u].x); atomicAdd(Sp + 1, sm[u].y); atomicAdd(Sp + 2, sm[u].z); atomicAdd(Sp + 3, sm[u].w);
    }
  }
}

extern "C" int64_t vsx_grn_q_reduce_ws_floats(int32_t nb, int32_t C) { return (int64_t)vsx_cdiv(nb, QR_SB) * C * 4 * C; }

extern "C" int32_t vsx_grn_q_reduce(const float* Q, const float* cs, const void* W2, const float* s, const float* beta, float* P,
                                    float* S, float* dW2, float* db2, float* ws, int64_t ws_floats, int32_t nb, int32_t C,
                                    int32_t dtype, vsx_stream_t stream) {
  VSX_CHECK(Q && cs && W2 && s && beta && P && S && dW2 && db2 && ws && nb > 0 && C > 0, "vsx_grn_q_reduce: bad arguments");
  const int G = vsx_cdiv(nb, QR_SB), N = 4 * C;
  VSX_CHECK(ws_floats >= (int64_t)G * C * N, "vsx_grn_q_reduce: workspace needs %ld floats (vsx_grn_q_reduce_ws_floats)",
            (long)((int64_t)G * C * N));
  VSX_CHECK((int64_t)C * N < (1ll << 31), "vsx_grn_q_reduce: C too large");
  hipStream_t st = (hipStream_t)stream;
  int cb = C >= 192 ? 64 : 32;
  // few samples (the 2048^2 gate shape: 8): 6 - 18 workgroups of 64-channel blocks ran 64 us per launch on an empty chip; halve the
  // block until the launch has ~256 workgroups (the atomics per address grow to C / 8 at most)
  while (cb > 8 && (long)vsx_cdiv(N, QR_TH * 4) * G * vsx_cdiv(C, cb) < 256) cb /= 2;
  dim3 grid(vsx_cdiv(N, QR_TH * 4), G, vsx_cdiv(C, cb));
  if (dtype == VSX_BF16)
    hipLaunchKernelGGL(grn_q_reduce_kernel<bf16_t>, grid, dim3(QR_TH), 0, st, Q, cs, (const bf16_t*)W2, s, beta, P, S, ws, db2, nb, C, cb);
  else
    hipLaunchKernelGGL(grn_q_reduce_kernel<float>, grid, dim3(QR_TH), 0, st, Q, cs, (const float*)W2, s, beta, P, S, ws, db2, nb, C, cb);
  // dW2[c, j] += sum over the groups' partials (the [C, 4C] matrix seen as one row of C * 4C columns)
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(vsx_cdiv((long)C * N, 64), vsx_cdiv(G, 64)), dim3(256), 0, st, (const float*)ws, dW2, G,
                     C * N);
  VSX_LAUNCH_CHECK();
  return 0;
}
