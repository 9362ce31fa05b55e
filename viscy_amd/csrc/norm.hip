// Channel LayerNorm, GRN statistics (K7) and the GRN/GELU backward pass.
// Rows are pixels of a channels-last tensor; a row's C channels are contiguous, so a row is
// reduced by an aligned group of G lanes of one wave64 with shuffle reductions (no LDS, no
// barriers); loads/stores are 16-byte vectors.
#include "vsx_common.h"
#include "../../include/vsx.h"

// ------------------------------------------------------------------ LayerNorm forward
template <typename T, int G, int CPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     int rows, int C, float eps) {
  constexpr int VN = VT<T>::N;
  const int nch = C / VN;
  const int row = blockIdx.x * (256 / G) + threadIdx.x / G;
  const int gl = threadIdx.x % G;
  if (row >= rows) return;  // whole groups exit together
  const T* xr = x + (size_t)row * C;
  float v[CPL][VN];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    int c = gl + i * G;
    if (c < nch) {
      unpack<T>(ldvec<T>(xr + c * VN), v[i]);
#pragma unroll
      for (int j = 0; j < VN; ++j) s += v[i][j];
    } else {
#pragma unroll
      for (int j = 0; j < VN; ++j) v[i][j] = 0.f;
    }
  }
  const float mean = group_sum<G>(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    int c = gl + i * G;
    if (c < nch) {
#pragma unroll
      for (int j = 0; j < VN; ++j) {
        float d = v[i][j] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(group_sum<G>(q) / (float)C + eps);
  T* yr = y + (size_t)row * C;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    int c = gl + i * G;
    if (c < nch) {
      float o[VN];
#pragma unroll
      for (int j = 0; j < VN; ++j) {
        float xh = (v[i][j] - mean) * rstd;
        o[j] = gamma ? xh * gamma[c * VN + j] + beta[c * VN + j] : xh;
      }
      stvec<T>(yr + c * VN, pack<T>(o));
    }
  }
  if (gl == 0) {
    if (mean_out) mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
}

// ------------------------------------------------------------------ LayerNorm backward
// xhat = mean ? (x - mean) * rstd : x ;  g = dy * gamma ;  dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)) [+ add]
template <typename T, int G, int CPL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const T* __restrict__ add,
                                                     T* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int rows, int C, int iters) {
  constexpr int VN = VT<T>::N;
  extern __shared__ float red[];  // [2][C] when dgamma != nullptr
  const int nch = C / VN;
  const int gl = threadIdx.x % G;
  const int rslot = threadIdx.x / G;
  constexpr int RPI = 256 / G;
  if (dgamma) {
    for (int i = threadIdx.x; i < 2 * C; i += 256) red[i] = 0.f;
    __syncthreads();
  }
  float dg[CPL][VN], db[CPL][VN], gam[CPL][VN];
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
  for (int it = 0; it < iters; ++it) {
    const int row = (blockIdx.x * iters + it) * RPI + rslot;
    if (row >= rows) break;
    const float mu = mean ? mean[row] : 0.f;
    const float rs = rstd[row];
    float xh[CPL][VN], g[CPL][VN];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      int c = gl + i * G;
      if (c < nch) {
        float xv[VN], dv[VN];
        unpack<T>(ldvec<T>(x + (size_t)row * C + c * VN), xv);
        unpack<T>(ldvec<T>(dy + (size_t)row * C + c * VN), dv);
#pragma unroll
        for (int j = 0; j < VN; ++j) {
          xh[i][j] = mean ? (xv[j] - mu) * rs : xv[j];
          g[i][j] = dv[j] * gam[i][j];
          s1 += g[i][j];
          s2 += g[i][j] * xh[i][j];
          dg[i][j] += dv[j] * xh[i][j];
          db[i][j] += dv[j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < VN; ++j) { xh[i][j] = 0.f; g[i][j] = 0.f; }
      }
    }
    s1 = group_sum<G>(s1) / (float)C;
    s2 = group_sum<G>(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      int c = gl + i * G;
      if (c < nch) {
        float o[VN];
#pragma unroll
        for (int j = 0; j < VN; ++j) o[j] = rs * (g[i][j] - s1 - xh[i][j] * s2);
        if (add) {
          float a[VN];
          unpack<T>(ldvec<T>(add + (size_t)row * C + c * VN), a);
#pragma unroll
          for (int j = 0; j < VN; ++j) o[j] += a[j];
        }
        stvec<T>(dx + (size_t)row * C + c * VN, pack<T>(o));
      }
    }
  }
  if (dgamma) {
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

template <typename T, int G, int CPL>
static int ln_launch(bool fwd, const void* a0, const void* a1, void* out, float* mean, float* rstd,
                     const float* gamma, const float* beta, const void* add, float* dgamma, float* dbeta, int rows,
                     int C, float eps, hipStream_t s) {
  constexpr int RPI = 256 / G;
  if (fwd) {
    hipLaunchKernelGGL((ln_fwd_kernel<T, G, CPL>), dim3(vsx_cdiv(rows, RPI)), dim3(256), 0, s, (const T*)a0, (T*)out,
                       mean, rstd, gamma, beta, rows, C, eps);
  } else {
    int iters = vsx_cdiv(rows, RPI * 2048);
    if (iters < 1) iters = 1;
    int grid = vsx_cdiv(rows, RPI * iters);
    size_t sh = dgamma ? 2 * (size_t)C * sizeof(float) : 0;
    hipLaunchKernelGGL((ln_bwd_kernel<T, G, CPL>), dim3(grid), dim3(256), sh, s, (const T*)a0, (const T*)a1,
                       (const float*)mean, (const float*)rstd, gamma, (const T*)add, (T*)out, dgamma, dbeta, rows, C,
                       iters);
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
__global__ __launch_bounds__(256) void grn_bwd_stats_kernel(const float* __restrict__ colsq, const float* __restrict__ P,
                                                            const float* __restrict__ gamma, float* __restrict__ t,
                                                            float* __restrict__ dgamma, int N, float eps) {
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
    atomicAdd(dgamma + n, pn * g * inv);
  }
}

/* K7: timm GlobalResponseNorm statistics.  colsq[b,n] = sum_hw gelu(h)^2 comes from the fc1 GEMM epilogue. */
extern "C" int32_t vsx_grn_scale(const float* colsq, const float* gamma, float* s, int32_t nb, int32_t N, float eps,
                                 vsx_stream_t stream) {
  VSX_CHECK(colsq && gamma && s && nb > 0 && N > 0, "vsx_grn_scale: bad arguments");
  hipLaunchKernelGGL(grn_scale_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, colsq, gamma, s, N, eps);
  VSX_LAUNCH_CHECK();
  return 0;
}
extern "C" int32_t vsx_grn_bwd_stats(const float* colsq, const float* P, const float* gamma, float* t, float* dgamma,
                                     int32_t nb, int32_t N, float eps, vsx_stream_t stream) {
  VSX_CHECK(colsq && P && gamma && t && dgamma && nb > 0 && N > 0, "vsx_grn_bwd_stats: bad arguments");
  hipLaunchKernelGGL(grn_bwd_stats_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, colsq, P, gamma, t, dgamma, N,
                     eps);
  VSX_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ GRN + GELU backward (pass 2)
// dh = (dz * s[b,n] + gelu(h) * t[b,n]) * gelu'(h), written over dz; colsum[n] += dh
template <typename T>
__global__ __launch_bounds__(256) void grn_gelu_bwd_kernel(T* __restrict__ dz, const T* __restrict__ h,
                                                           const float* __restrict__ s, const float* __restrict__ t,
                                                           float* __restrict__ colsum, int M, int N, int hw,
                                                           int rows_per_block, int tpr) {
  constexpr int VN = VT<T>::N;
  // tpr threads (a power of two <= 256) span the column chunks of a row; 256/tpr row slots per block
  const int cc = blockIdx.x * tpr + threadIdx.x % tpr;
  const int slot = threadIdx.x / tpr, nslot = 256 / tpr;
  if (cc * VN >= N) return;
  const int n = cc * VN;
  const int r0 = blockIdx.y * rows_per_block + slot;
  const int rend = blockIdx.y * rows_per_block + rows_per_block;
  const int r1 = rend < M ? rend : M;
  float cs[VN];
#pragma unroll
  for (int j = 0; j < VN; ++j) cs[j] = 0.f;
  int bcur = -1;
  float sv[VN], tv[VN];
  for (int m = r0; m < r1; m += nslot) {
    const int b = m / hw;
    if (b != bcur) {
      bcur = b;
#pragma unroll
      for (int j = 0; j < VN; ++j) { sv[j] = s[(size_t)b * N + n + j]; tv[j] = t[(size_t)b * N + n + j]; }
    }
    float dv[VN], hv[VN], o[VN];
    unpack<T>(ldvec<T>(dz + (size_t)m * N + n), dv);
    unpack<T>(ldvec<T>(h + (size_t)m * N + n), hv);
#pragma unroll
    for (int j = 0; j < VN; ++j) {
      float dgv = dv[j] * sv[j] + gelu_f(hv[j]) * tv[j];
      o[j] = round_to<T>(dgv * gelu_grad_f(hv[j]));
      cs[j] += o[j];
    }
    stvec<T>(dz + (size_t)m * N + n, pack<T>(o));
  }
#pragma unroll
  for (int j = 0; j < VN; ++j) atomicAdd(colsum + n + j, cs[j]);
}

extern "C" int32_t vsx_grn_gelu_bwd(void* dz, const void* h, const float* s, const float* t, float* colsum, int32_t M,
                                    int32_t N, int32_t hw, int32_t dtype, vsx_stream_t stream) {
  int vn = dtype == VSX_BF16 ? 8 : 4;
  VSX_CHECK(dz && h && s && t && colsum && M > 0 && N > 0 && hw > 0 && N % vn == 0, "vsx_grn_gelu_bwd: bad arguments");
  int ncc = N / vn;
  int tpr = 1;
  while (tpr < ncc && tpr < 256) tpr <<= 1;
  int gx = vsx_cdiv(ncc, tpr);
  int rpb = vsx_cdiv(M, vsx_cdiv(4096, gx));
  int min_rows = 16 * (256 / tpr);
  if (rpb < min_rows) rpb = min_rows;
  dim3 grid(gx, vsx_cdiv(M, rpb));
  if (dtype == VSX_BF16)
    hipLaunchKernelGGL(grn_gelu_bwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (bf16_t*)dz,
                       (const bf16_t*)h, s, t, colsum, M, N, hw, rpb, tpr);
  else
    hipLaunchKernelGGL(grn_gelu_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (float*)dz, (const float*)h,
                       s, t, colsum, M, N, hw, rpb, tpr);
  VSX_LAUNCH_CHECK();
  return 0;
}
