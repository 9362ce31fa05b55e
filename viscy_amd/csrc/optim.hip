// Parameter-space kernels: fused AdamW over the flat parameter buffer, and the
// per-step re-layout of fp32 master weights into GEMM operands (cast to the compute dtype, tap-major
// K order, transposed copy for the data-gradient GEMM, LayerNorm affine folded into fc1) plus the
// inverse mapping for gradients.  All are tiny HBM-bound passes over <=32 M parameters.
#include "vsx_common.h"
#include "wtasks.h"
#include "../../include/vsx.h"

// ------------------------------------------------------------------ AdamW (torch.optim.AdamW semantics)
// hyper = {lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2, grad_scale}
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    const float* __restrict__ hyper, long n) {
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], bc1 = hyper[5],
              bc2 = hyper[6], gs = hyper[7];
  const float step = lr / bc1, rbc2 = rsqrtf(bc2);
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * 256 * 4;
  for (; i + 3 < n; i += stride) {
    float4 pv = *reinterpret_cast<float4*>(p + i);
    float4 gv = *reinterpret_cast<const float4*>(g + i);
    float4 mv = *reinterpret_cast<float4*>(m + i);
    float4 vv = *reinterpret_cast<float4*>(v + i);
    float* pp = &pv.x;
    float* gp = &gv.x;
    float* mp = &mv.x;
    float* vp = &vv.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float gr = gp[j] * gs;
      float pj = pp[j] * (1.f - lr * wd);
      mp[j] = b1 * mp[j] + (1.f - b1) * gr;
      vp[j] = b2 * vp[j] + (1.f - b2) * gr * gr;
      pp[j] = pj - step * mp[j] / (sqrtf(vp[j]) * rbc2 + eps);
    }
    *reinterpret_cast<float4*>(p + i) = pv;
    *reinterpret_cast<float4*>(m + i) = mv;
    *reinterpret_cast<float4*>(v + i) = vv;
  }
  // tail (n not a multiple of 4): handled by the first thread of the last stride window
  if (i < n && i + 3 >= n) {
    for (long k = i; k < n; ++k) {
      float gr = g[k] * gs;
      float pj = p[k] * (1.f - lr * wd);
      float mk = b1 * m[k] + (1.f - b1) * gr;
      float vk = b2 * v[k] + (1.f - b2) * gr * gr;
      m[k] = mk;
      v[k] = vk;
      p[k] = pj - step * mk / (sqrtf(vk) * rbc2 + eps);
    }
  }
}

/* K26: torch.optim.AdamW step (viscy_utils/optimizers.py:50) on flat fp32 buffers; hyper is a
 * device array of 8 floats so the launch is hipGraph-replayable while lr / step change. */
extern "C" int32_t vsx_adamw(float* p, const float* g, float* m, float* v, const float* hyper, int64_t n,
                             vsx_stream_t stream) {
  VSX_CHECK(p && g && m && v && hyper && n > 0, "vsx_adamw: bad arguments");
  int grid = vsx_cdiv(n, 256L * 4 * 4);
  if (grid > 4096) grid = 4096;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, hyper, (long)n);
  VSX_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ fill (zero_grad, reduction targets, the loss's scalar block)
// The captured step carries no library launch: gradient zeroing, the zero arenas of the two passes and the loss's running
// max start value are ordinary kernel nodes of this library (they were ATen fills).
__global__ __launch_bounds__(256) void fill_f32_kernel(float* __restrict__ p, long n, float v) {
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) *reinterpret_cast<float4*>(p + i) = make_float4(v, v, v, v);
  else for (; i < n; ++i) p[i] = v;
}

extern "C" int32_t vsx_fill_f32(float* p, int64_t n, float value, vsx_stream_t stream) {
  VSX_CHECK(p != nullptr && n >= 0 && ((uintptr_t)p & 15) == 0, "vsx_fill_f32: null or unaligned pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(fill_f32_kernel, dim3(vsx_cdiv(n, 1024L)), dim3(256), 0, (hipStream_t)stream, p, (long)n, value);
  VSX_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ schedule + bias corrections on the device
// One thread turns (constants, step counter) into the 8 per-step scalars the AdamW launch reads and advances the
// counter.  The first version refreshed those scalars from a pinned host block with an asynchronous copy inside the
// captured step: the copy reads the block when the GPU gets there, the host — many replays ahead of a 140 ms step — had
// already overwritten it with a later step's lr / bias corrections (optim.py race).  Now nothing that changes
// per step lives on the host: a hipGraph replay is a pure function of device state.
// cfg is DOUBLE: beta2 = 0.999 as a float is 0.99900001287, which moved the bias correction 1 - beta2^t by 1.3e-5 (relative)
// at t = 1 against torch.optim.AdamW's host-double arithmetic; step counts stay exact beyond 2^24.
// cfg: {base_lr, beta1, beta2, eps, weight_decay, grad_scale, schedule (0 constant | 1 MONAI WarmupCosine),
//       warmup_steps, t_total, warmup_multiplier, cycles}
__global__ void adamw_advance_kernel(const double* __restrict__ cfg, int* __restrict__ step, float* __restrict__ hyper) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int t0 = *step;  // optimiser steps taken so far = index of this step in the LambdaLR schedule
  const double b1 = cfg[1], b2 = cfg[2];
  double lam = 1.0;
  if (cfg[6] > 0.5) {
    const double warm = cfg[7], total = cfg[8], mult = cfg[9], cycles = cfg[10];
    if ((double)t0 < warm) {
      lam = mult + (1.0 - mult) * ((double)t0 / fmax(1.0, warm));
    } else {
      const double progress = ((double)t0 - warm) / fmax(1.0, total - warm);
      lam = fmax(0.0, 0.5 * (1.0 + cos(3.14159265358979323846 * cycles * 2.0 * progress)));
    }
  }
  const int t = t0 + 1;
  hyper[0] = (float)((double)cfg[0] * lam);
  hyper[1] = (float)cfg[1];
  hyper[2] = (float)cfg[2];
  hyper[3] = (float)cfg[3];
  hyper[4] = (float)cfg[4];
  hyper[5] = (float)(1.0 - pow(b1, (double)t));
  hyper[6] = (float)(1.0 - pow(b2, (double)t));
  hyper[7] = (float)cfg[5];
  *step = t;
}

extern "C" int32_t vsx_adamw_advance(const double* cfg, int32_t* step, float* hyper, vsx_stream_t stream) {
  VSX_CHECK(cfg && step && hyper, "vsx_adamw_advance: bad arguments");
  hipLaunchKernelGGL(adamw_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, cfg, step, hyper);
  VSX_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ weight preparation
template <typename T>
__global__ __launch_bounds__(256) void prep_weight_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                          T* __restrict__ dstT, const float* __restrict__ gamma, int R,
                                                          int Cs, int Tn, int tapmode) {
  wt_prep_weight<T>(blockIdx.x, src, dst, dstT, gamma, R, Cs, Tn, tapmode);
}

/* src: fp32 parameter viewed as [R, Cs, Tn] (out, in-channels, taps) → dst [R, Tn*Cs] and/or
 * dstT [Tn*Cs, R] in `dtype`, optionally scaled per input channel by gamma[Cs] (LayerNorm fold). */
extern "C" int32_t vsx_prep_weight(const float* src, void* dst, void* dstT, const float* gamma, int32_t R, int32_t Cs,
                                   int32_t Tn, int32_t tapmode, int32_t dtype, vsx_stream_t stream) {
  VSX_CHECK(src && (dst || dstT) && R > 0 && Cs > 0 && Tn > 0, "vsx_prep_weight: bad arguments");
  VSX_CHECK(tapmode == 0 || (tapmode == 1 && Tn == 27), "vsx_prep_weight: tapmode 1 needs 27 taps");
  long total = (long)R * Cs * Tn;
  dim3 grid(vsx_cdiv(total, 256));
  if (dtype == VSX_BF16)
    hipLaunchKernelGGL(prep_weight_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst,
                       (bf16_t*)dstT, gamma, R, Cs, Tn, tapmode);
  else
    hipLaunchKernelGGL(prep_weight_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, src, (float*)dst,
                       (float*)dstT, gamma, R, Cs, Tn, tapmode);
  VSX_LAUNCH_CHECK();
  return 0;
}

// gradient back-mapping (body and grid shape: csrc/wtasks.h)
__global__ __launch_bounds__(256) void unprep_grad_kernel(const float* __restrict__ g, float* __restrict__ dparam,
                                                          const float* __restrict__ gamma, const float* __restrict__ W,
                                                          float* __restrict__ dgamma, const float* __restrict__ u,
                                                          const float* __restrict__ beta, const float* __restrict__ rowsub, int R,
                                                          int Cs, int Tn, int tapmode) {
  wt_unprep_grad(blockIdx.x, g, dparam, gamma, W, dgamma, u, beta, rowsub, R, Cs, Tn, tapmode);
}

extern "C" int32_t vsx_unprep_grad(const float* g, float* dparam, const float* gamma, const float* W, float* dgamma,
                                   const float* u, const float* beta, const float* rowsub, int32_t R, int32_t Cs, int32_t Tn,
                                   int32_t tapmode, vsx_stream_t stream) {
  VSX_CHECK(g && dparam && R > 0 && Cs > 0 && Tn > 0, "vsx_unprep_grad: bad arguments");
  VSX_CHECK((gamma == nullptr) == (dgamma == nullptr) && (gamma == nullptr || W != nullptr),
            "vsx_unprep_grad: gamma / dgamma / W must come together");
  VSX_CHECK((u == nullptr) == (beta == nullptr) && (u == nullptr || Tn == 1), "vsx_unprep_grad: u / beta need Tn == 1");
  dim3 grid(wt_unprep_bx(Cs, Tn) * vsx_cdiv(R, wt_unprep_rpb(R)));
  hipLaunchKernelGGL(unprep_grad_kernel, grid, dim3(256), 0, (hipStream_t)stream, g, dparam, gamma, W, dgamma, u, beta, rowsub,
                     R, Cs, Tn, tapmode);
  VSX_LAUNCH_CHECK();
  return 0;
}

// out[r] = (b ? b[r] : 0) + Σ_c W[r][c] * v[c]          (fold LN beta into the fc1 bias)
__global__ __launch_bounds__(256) void matvec_kernel(const float* __restrict__ W, const float* __restrict__ v,
                                                     const float* __restrict__ b, float* __restrict__ out, int R,
                                                     int C) {
  wt_matvec(blockIdx.x, W, v, b, out, R, C);
}
// out[c] += Σ_r W[r][c] * u[r]                            (gradient of the folded LN beta; body: csrc/wtasks.h)
__global__ __launch_bounds__(256) void matvec_t_kernel(const float* __restrict__ W, const float* __restrict__ u,
                                                       float* __restrict__ out, int R, int C) {
  wt_matvec_t(blockIdx.x, W, u, out, R, C);
}

extern "C" int32_t vsx_matvec(const float* W, const float* v, const float* b, float* out, int32_t R, int32_t C,
                              vsx_stream_t stream) {
  VSX_CHECK(W && v && out && R > 0 && C > 0, "vsx_matvec: bad arguments");
  hipLaunchKernelGGL(matvec_kernel, dim3(vsx_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, W, v, b, out, R, C);
  VSX_LAUNCH_CHECK();
  return 0;
}
extern "C" int32_t vsx_matvec_t_add(const float* W, const float* u, float* out, int32_t R, int32_t C,
                                    vsx_stream_t stream) {
  VSX_CHECK(W && u && out && R > 0 && C > 0, "vsx_matvec_t_add: bad arguments");
  dim3 grid(vsx_cdiv(C, 256) * vsx_cdiv(R, wt_matvec_t_rpb(R)));
  hipLaunchKernelGGL(matvec_t_kernel, grid, dim3(256), 0, (hipStream_t)stream, W, u, out, R, C);
  VSX_LAUNCH_CHECK();
  return 0;
}

// dst[j][i] (+)= src[i][j]   (fp32; depthwise weights [C][49] <-> [49][C])
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int A,
                                                            int Bn, int accumulate) {
  wt_transpose_f32(blockIdx.x, src, dst, A, Bn, accumulate);
}
extern "C" int32_t vsx_transpose_f32(const float* src, float* dst, int32_t A, int32_t Bn, int32_t accumulate,
                                     vsx_stream_t stream) {
  VSX_CHECK(src && dst && A > 0 && Bn > 0, "vsx_transpose_f32: bad arguments");
  hipLaunchKernelGGL(transpose_f32_kernel, dim3(vsx_cdiv((long)A * Bn, 256)), dim3(256), 0, (hipStream_t)stream, src, dst,
                     A, Bn, accumulate);
  VSX_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ weight-space task list
// One launch for a list of independent weight-space jobs (vsx_prep_weight / vsx_transpose_f32 / vsx_matvec / vsx_mlp_pack
// bodies, csrc/wtasks.h).  The list travels BY VALUE in the kernel arguments (a hipGraph node keeps it; no device-side table
// to keep alive): <= VSX_WTASK_MAX tasks per launch, workgroup -> task by a binary search over the prefix sums.
struct WTaskBatch {
  int n;
  int start[VSX_WTASK_MAX + 1];
  VsxWTask t[VSX_WTASK_MAX];
};
static_assert(sizeof(WTaskBatch) <= 4096, "the task list must fit the kernel-argument segment");

__global__ __launch_bounds__(256) void weight_tasks_kernel(const WTaskBatch b) {
  int lo = 0, hi = b.n;  // largest lo with start[lo] <= blockIdx.x
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (b.start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
  }
  const VsxWTask& t = b.t[lo];
  const int blk = (int)blockIdx.x - b.start[lo];
  switch (t.kind) {
    case VSX_WTASK_PREP:
      if (t.dtype == VSX_BF16)
        wt_prep_weight<bf16_t>(blk, (const float*)t.p0, (bf16_t*)t.p1, (bf16_t*)t.p2, (const float*)t.p3, t.i0, t.i1, t.i2, t.i3);
      else
        wt_prep_weight<float>(blk, (const float*)t.p0, (float*)t.p1, (float*)t.p2, (const float*)t.p3, t.i0, t.i1, t.i2, t.i3);
      break;
    case VSX_WTASK_TRANSPOSE: wt_transpose_f32(blk, (const float*)t.p0, (float*)t.p1, t.i0, t.i1, t.i2); break;
    case VSX_WTASK_MATVEC: wt_matvec(blk, (const float*)t.p0, (const float*)t.p3, (const float*)t.p2, (float*)t.p1, t.i0, t.i1); break;
    case VSX_WTASK_MLP_PACK: wt_mlp_pack(blk, (const bf16_t*)t.p0, (const bf16_t*)t.p3, (char*)t.p1, t.i0); break;
    case VSX_WTASK_UNPREP:
      wt_unprep_grad(blk, (const float*)t.p0, (float*)t.p1, (const float*)t.p3, (const float*)t.p4, (float*)t.p2,
                     (const float*)t.p5, (const float*)t.p6, (const float*)t.p7, t.i0, t.i1, t.i2, t.i3);
      break;
    case VSX_WTASK_MATVEC_T: wt_matvec_t(blk, (const float*)t.p0, (const float*)t.p3, (float*)t.p1, t.i0, t.i1); break;
    case VSX_WTASK_REDUCE_ROWS: wt_reduce_rows(blk, (const float*)t.p0, (float*)t.p1, t.i0, t.i1); break;
    default: break;
  }
}

/* n independent weight-space jobs in ceil(n / VSX_WTASK_MAX) launches.  Per kind (fields of VsxWTask, include/vsx.h):
 *   VSX_WTASK_PREP      = vsx_prep_weight(p0 src, p1 dst, p2 dstT, p3 gamma, i0 R, i1 Cs, i2 Tn, i3 tapmode, dtype)
 *   VSX_WTASK_TRANSPOSE = vsx_transpose_f32(p0 src, p1 dst, i0 A, i1 Bn, i2 accumulate)
 *   VSX_WTASK_MATVEC    = vsx_matvec(p0 W, p3 v, p2 b, p1 out, i0 R, i1 C)
 *   VSX_WTASK_MLP_PACK  = vsx_mlp_pack(p0 W1, p3 W2, p1 img, i0 C)
 *   VSX_WTASK_UNPREP    = vsx_unprep_grad(p0 g, p1 dparam, p3 gamma, p4 W, p2 dgamma, p5 u, p6 beta, p7 rowsub, i0 R, i1 Cs, i2 Tn, i3 tapmode)
 *   VSX_WTASK_MATVEC_T  = vsx_matvec_t_add(p0 W, p3 u, p1 out, i0 R, i1 C)
 *   VSX_WTASK_REDUCE_ROWS: p1 out[n] += sum over the i0 rows of p0 ws [i0, i1]   (column sums of a workspace of partials)
 * No task may read or accumulate into what another task of the same call writes. */
extern "C" int32_t vsx_weight_tasks(const VsxWTask* tasks, int32_t n, vsx_stream_t stream) {
  VSX_CHECK(tasks && n > 0, "vsx_weight_tasks: bad arguments");
  for (int i0 = 0; i0 < n; i0 += VSX_WTASK_MAX) {
    WTaskBatch b;
    b.n = n - i0 < VSX_WTASK_MAX ? n - i0 : VSX_WTASK_MAX;
    long total = 0;
    for (int j = 0; j < b.n; ++j) {
      const VsxWTask& t = tasks[i0 + j];
      long blocks = 0;
      switch (t.kind) {
        case VSX_WTASK_PREP:
          VSX_CHECK(t.p0 && (t.p1 || t.p2) && t.i0 > 0 && t.i1 > 0 && t.i2 > 0 && (t.i3 == 0 || (t.i3 == 1 && t.i2 == 27)) &&
                        (t.dtype == VSX_BF16 || t.dtype == VSX_F32), "vsx_weight_tasks: task %d (prep_weight): bad arguments", i0 + j);
          blocks = vsx_cdiv((long)t.i0 * t.i1 * t.i2, 256L);
          break;
        case VSX_WTASK_TRANSPOSE:
          VSX_CHECK(t.p0 && t.p1 && t.i0 > 0 && t.i1 > 0, "vsx_weight_tasks: task %d (transpose_f32): bad arguments", i0 + j);
          blocks = vsx_cdiv((long)t.i0 * t.i1, 256L);
          break;
        case VSX_WTASK_MATVEC:
          VSX_CHECK(t.p0 && t.p3 && t.p1 && t.i0 > 0 && t.i1 > 0, "vsx_weight_tasks: task %d (matvec): bad arguments", i0 + j);
          blocks = vsx_cdiv(t.i0, 4);
          break;
        case VSX_WTASK_MLP_PACK:
          VSX_CHECK(t.p0 && t.p3 && t.p1 && t.i0 > 0 && t.i0 % 32 == 0, "vsx_weight_tasks: task %d (mlp_pack): C must be a multiple of 32", i0 + j);
          blocks = vsx_cdiv(wt_mlp_pack_items(t.i0), 256L);
          break;
        case VSX_WTASK_UNPREP:
          VSX_CHECK(t.p0 && t.p1 && t.i0 > 0 && t.i1 > 0 && t.i2 > 0 && (t.p3 == nullptr) == (t.p2 == nullptr) &&
                        (t.p3 == nullptr || t.p4 != nullptr) && (t.p5 == nullptr) == (t.p6 == nullptr) && (t.p5 == nullptr || t.i2 == 1),
                    "vsx_weight_tasks: task %d (unprep_grad): bad arguments", i0 + j);
          blocks = (long)wt_unprep_bx(t.i1, t.i2) * vsx_cdiv(t.i0, wt_unprep_rpb(t.i0));
          break;
        case VSX_WTASK_MATVEC_T:
          VSX_CHECK(t.p0 && t.p3 && t.p1 && t.i0 > 0 && t.i1 > 0, "vsx_weight_tasks: task %d (matvec_t_add): bad arguments", i0 + j);
          blocks = (long)vsx_cdiv(t.i1, 256) * vsx_cdiv(t.i0, wt_matvec_t_rpb(t.i0));
          break;
        case VSX_WTASK_REDUCE_ROWS:
          VSX_CHECK(t.p0 && t.p1 && t.i0 > 0 && t.i1 > 0, "vsx_weight_tasks: task %d (reduce_rows): bad arguments", i0 + j);
          blocks = (long)vsx_cdiv(t.i1, 64) * vsx_cdiv(t.i0, 64);
          break;
        default:
          vsx_set_error("vsx_weight_tasks: task %d has unknown kind %d", i0 + j, t.kind);
          return 1;
      }
      b.start[j] = (int)total;
      b.t[j] = t;
      total += blocks;
      VSX_CHECK(total < (1l << 31), "vsx_weight_tasks: too many workgroups");
    }
    b.start[b.n] = (int)total;
    hipLaunchKernelGGL(weight_tasks_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, b);
    VSX_LAUNCH_CHECK();
  }
  return 0;
}

// head Conv3d data-gradient weights: dst[zp][c3][((ty*3+tx)*3 + j)*Cmid + o] =
//   W[o][c3][kz][2-ty][2-tx]  with kz = zp - (zs + j), zs = clamp(zp-2, 0, Zout-3), zero when kz ∉ [0,2]
template <typename T>
__global__ __launch_bounds__(256) void prep_head_dgrad_kernel(const float* __restrict__ W, T* __restrict__ dst, int Cmid,
                                                              int C3, int Zout) {
  const int Zin = Zout + 2;
  const int K = 27 * Cmid;
  const long total = (long)Zin * C3 * K;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int k = (int)(gid % K);
  const long r = gid / K;
  const int c3 = (int)(r % C3);
  const int zp = (int)(r / C3);
  const int o = k % Cmid;
  const int tj = k / Cmid;
  const int j = tj % 3, tap = tj / 3;
  const int ty = tap / 3, tx = tap % 3;
  int zs = zp - 2;
  if (zs < 0) zs = 0;
  if (zs > Zout - 3) zs = Zout - 3;
  const int kz = zp - (zs + j);
  float v = 0.f;
  if (kz >= 0 && kz <= 2) v = W[((((size_t)o * C3 + c3) * 3 + kz) * 3 + (2 - ty)) * 3 + (2 - tx)];
  dst[gid] = from_f32<T>(v);
}
extern "C" int32_t vsx_prep_head_dgrad(const float* W, void* dst, int32_t Cmid, int32_t C3, int32_t Zout, int32_t dtype,
                                       vsx_stream_t stream) {
  VSX_CHECK(W && dst && Cmid > 0 && C3 > 0 && Zout >= 3, "vsx_prep_head_dgrad: bad arguments (Zout=%d must be >= 3)", Zout);
  long total = (long)(Zout + 2) * C3 * 27 * Cmid;
  dim3 grid(vsx_cdiv(total, 256));
  if (dtype == VSX_BF16)
    hipLaunchKernelGGL(prep_head_dgrad_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, W, (bf16_t*)dst, Cmid, C3,
                       Zout);
  else
    hipLaunchKernelGGL(prep_head_dgrad_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, W, (float*)dst, Cmid, C3,
                       Zout);
  VSX_LAUNCH_CHECK();
  return 0;
}

// out[b][r][k] = T(W[r][k] * s[b][k])   (GRN scale folded into per-sample fc2 weights; one 16-byte vector per thread)
template <typename T>
__global__ __launch_bounds__(256) void scale_weight_samples_kernel(const float* __restrict__ W, const float* __restrict__ s,
                                                                   T* __restrict__ out, int B, int R, int K) {
  constexpr int VN = VT<T>::N;
  const int kv = K / VN;
  const long total = (long)B * R * kv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % kv);
    const long br = i / kv;
    const int r = (int)(br % R), b = (int)(br / R);
    float f[VN];
#pragma unroll
    for (int j = 0; j < VN; j += 4) {
      const float4 w = *reinterpret_cast<const float4*>(W + (size_t)r * K + c * VN + j);
      const float4 sv = *reinterpret_cast<const float4*>(s + (size_t)b * K + c * VN + j);
      f[j] = w.x * sv.x; f[j + 1] = w.y * sv.y; f[j + 2] = w.z * sv.z; f[j + 3] = w.w * sv.w;
    }
    stvec<T>(out + ((size_t)b * R + r) * K + c * VN, pack<T>(f));
  }
}
extern "C" int32_t vsx_scale_weight_samples(const float* W, const float* s, void* out, int32_t B, int32_t R, int32_t K,
                                            int32_t dtype, vsx_stream_t stream) {
  int vn = dtype == VSX_BF16 ? 8 : 4;
  VSX_CHECK(W && s && out && B > 0 && R > 0 && K > 0 && K % vn == 0, "vsx_scale_weight_samples: bad arguments (K=%d)", K);
  long total = (long)B * R * (K / vn);
  int g = vsx_cdiv(total, 256);
  if (g > 8192) g = 8192;
  if (dtype == VSX_BF16)
    hipLaunchKernelGGL(scale_weight_samples_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, W, s, (bf16_t*)out, B, R, K);
  else
    hipLaunchKernelGGL(scale_weight_samples_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, W, s, (float*)out, B, R, K);
  VSX_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ ConvNeXt-V1 layer scale folded into fc2 (DynaCLR trunk)
// timm ConvNeXtBlock: y = x + gamma * (fc2(h) ) = x + h (diag(gamma) W2)^T + gamma * b2: the block kernels run with
//   Ws[r, :] = gamma[r] * W[r, :],  bs[r] = gamma[r] * b[r]
// and the gradients of (Ws, bs) are unfolded:  dW += gamma[r] * dWs[r, :],  db += gamma * dbs,
//   dgamma[r] += sum_k dWs[r, k] W[r, k] + dbs[r] b[r].            One wave per row (K = 4C elements), all fp32.
__global__ __launch_bounds__(256) void layer_scale_fold_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                               const float* __restrict__ gamma, float* __restrict__ Ws,
                                                               float* __restrict__ bs, int R, int K) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  const float g = gamma[r];
  for (int k = lane; k < K; k += 64) Ws[(size_t)r * K + k] = g * W[(size_t)r * K + k];
  if (lane == 0) bs[r] = g * b[r];
}
__global__ __launch_bounds__(256) void layer_scale_unfold_kernel(const float* __restrict__ dWs, const float* __restrict__ dbs,
                                                                 const float* __restrict__ W, const float* __restrict__ b,
                                                                 const float* __restrict__ gamma, float* __restrict__ dW,
                                                                 float* __restrict__ db, float* __restrict__ dgamma, int R,
                                                                 int K) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  const float g = gamma[r];
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float d = dWs[(size_t)r * K + k];
    dW[(size_t)r * K + k] += g * d;
    acc = fmaf(d, W[(size_t)r * K + k], acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    db[r] += g * dbs[r];
    dgamma[r] += acc + dbs[r] * b[r];
  }
}
extern "C" int32_t vsx_layer_scale_fold(const float* W, const float* b, const float* gamma, float* Ws, float* bs, int32_t R,
                                        int32_t K, vsx_stream_t stream) {
  VSX_CHECK(W && b && gamma && Ws && bs && R > 0 && K > 0, "vsx_layer_scale_fold: bad arguments");
  hipLaunchKernelGGL(layer_scale_fold_kernel, dim3(vsx_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, W, b, gamma, Ws, bs, R, K);
  VSX_LAUNCH_CHECK();
  return 0;
}
extern "C" int32_t vsx_layer_scale_unfold(const float* dWs, const float* dbs, const float* W, const float* b, const float* gamma,
                                          float* dW, float* db, float* dgamma, int32_t R, int32_t K, vsx_stream_t stream) {
  VSX_CHECK(dWs && dbs && W && b && gamma && dW && db && dgamma && R > 0 && K > 0, "vsx_layer_scale_unfold: bad arguments");
  hipLaunchKernelGGL(layer_scale_unfold_kernel, dim3(vsx_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, dWs, dbs, W, b, gamma, dW,
                     db, dgamma, R, K);
  VSX_LAUNCH_CHECK();
  return 0;
}
