// viscy-transforms hot-path pieces and the predict-time Z blend:
// HBM-bound elementwise kernels over (B, C, Z, Y, X) fp32 stacks, 16-byte accesses along the
// contiguous X axis, per-sample parameters broadcast from small device arrays.
#include "vsx_common.h"
#include "../../include/vsx.h"

// y = (x - sub[b]) / (div[b] + 1e-8)
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        const float* __restrict__ sub, const float* __restrict__ dv,
                                                        long per_sample, long total) {
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (long)gridDim.x * 256 * 4) {
    const int b = (int)(i / per_sample);
    const float s = sub[b], d = dv[b] + 1e-8f;
    float4 v = *reinterpret_cast<const float4*>(x + i);
    v.x = (v.x - s) / d; v.y = (v.y - s) / d; v.z = (v.z - s) / d; v.w = (v.w - s) / d;
    *reinterpret_cast<float4*>(y + i) = v;
  }
}

// y = clamp(x, lo, hi) → 2 (x - lo) / (hi - lo + 1e-8) - 1
__global__ __launch_bounds__(256) void minmax_norm_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          const float* __restrict__ lo, const float* __restrict__ hi,
                                                          long per_sample, long total) {
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (long)gridDim.x * 256 * 4) {
    const int b = (int)(i / per_sample);
    const float l = lo[b], h = hi[b];
    const float inv = 1.f / (h - l + 1e-8f);
    float4 v = *reinterpret_cast<const float4*>(x + i);
    float* p = &v.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = 2.f * (fminf(fmaxf(p[j], l), h) - l) * inv - 1.f;
    *reinterpret_cast<float4*>(y + i) = v;
  }
}

__device__ __forceinline__ void atomic_minmax(float* mn, float* mx, float lo, float hi) {
  // order-preserving int mapping of floats
  if (lo >= 0.f) atomicMin(reinterpret_cast<int*>(mn), __float_as_int(lo));
  else atomicMax(reinterpret_cast<unsigned int*>(mn), __float_as_uint(lo));
  if (hi >= 0.f) atomicMax(reinterpret_cast<int*>(mx), __float_as_int(hi));
  else atomicMin(reinterpret_cast<unsigned int*>(mx), __float_as_uint(hi));
}

// per-sample min / max (mn pre-set to +inf, mx to -inf); gridDim.y = B, <= 64 blocks per sample
__global__ __launch_bounds__(256) void sample_minmax_kernel(const float* __restrict__ x, float* __restrict__ mn,
                                                            float* __restrict__ mx, long per_sample) {
  const int b = blockIdx.y;
  const float* xs = x + (size_t)b * per_sample;
  float lo = INFINITY, hi = -INFINITY;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < per_sample; i += (long)gridDim.x * 256 * 4) {
    float4 v = *reinterpret_cast<const float4*>(xs + i);
    lo = fminf(fminf(lo, v.x), fminf(v.y, fminf(v.z, v.w)));
    hi = fmaxf(fmaxf(hi, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
  }
  lo = -wave_max(-lo);
  hi = wave_max(hi);
  if ((threadIdx.x & 63) == 0 && hi >= lo) atomic_minmax(mn + b, mx + b, lo, hi);
}

// per-sample sum and sum of squares in double (MONAI AdjustContrast(retain_stats=True) wants mean / unbiased std of a sample
// before and after the gamma curve); sums[2b], sums[2b + 1] pre-set to 0; gridDim.y = B
__global__ __launch_bounds__(256) void sample_moments_kernel(const float* __restrict__ x, double* __restrict__ sums, long per_sample) {
  const int b = blockIdx.y;
  const float* xs = x + (size_t)b * per_sample;
  double s1 = 0.0, s2 = 0.0;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < per_sample; i += (long)gridDim.x * 256 * 4) {
    const float4 v = *reinterpret_cast<const float4*>(xs + i);
    s1 += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
    s2 += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o);
    s2 += __shfl_xor(s2, o);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(sums + 2 * b, s1);
    atomicAdd(sums + 2 * b + 1, s2);
  }
}

// fused intensity augmentation (BatchedRandAdjustContrast → BatchedRandScaleIntensity → BatchedRandGaussianNoise):
//   gamma[b] > 0 : x = ((x - m)/(r + 1e-7))^gamma * r + m     (MONAI AdjustContrast, m = min, r = max - min)
//                  invert bit 0 (invert_image=True): the curve runs on v = -x (m = -max, same r); bit 1: ... and the result is
//                  negated back (bit 0 alone hands the un-negated curve to the retain_stats pass of the host)
//   x *= (1 + factor[b])
//   nstd[b] >= 0 (applied): x += nmean + noise[i mod per_sample] * nstd[b]   (one field shared by the batch)
__global__ __launch_bounds__(256) void intensity_aug_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            const float* __restrict__ mn, const float* __restrict__ mx,
                                                            const float* __restrict__ gamma, const float* __restrict__ factor,
                                                            const float* __restrict__ noise, const float* __restrict__ nstd,
                                                            float nmean, int invert, long per_sample, long total) {
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (long)gridDim.x * 256 * 4) {
    const int b = (int)(i / per_sample);
    const long off = i - (long)b * per_sample;
    float4 v = *reinterpret_cast<const float4*>(x + i);
    float* p = &v.x;
    const float g = gamma ? gamma[b] : 0.f;
    if (g > 0.f) {
      const float sg = (invert & 1) ? -1.f : 1.f, so = (invert & 2) ? -1.f : 1.f;
      const float m = (invert & 1) ? -mx[b] : mn[b], r = mx[b] - mn[b];
      const float inv = 1.f / (r + 1e-7f);
#pragma unroll
      for (int j = 0; j < 4; ++j) p[j] = so * (__powf((sg * p[j] - m) * inv, g) * r + m);
    }
    if (factor) {
      const float f = 1.f + factor[b];
#pragma unroll
      for (int j = 0; j < 4; ++j) p[j] *= f;
    }
    if (noise && nstd[b] >= 0.f) {
      const float4 nz = *reinterpret_cast<const float4*>(noise + off);
      const float s = nstd[b];
      p[0] += nmean + nz.x * s; p[1] += nmean + nz.y * s; p[2] += nmean + nz.z * s; p[3] += nmean + nz.w * s;
    }
    *reinterpret_cast<float4*>(y + i) = v;
  }
}

// out[b,c,z,y,x] = old * (f_z - 1) / f_z + new / f_z    (prediction_writer._blend_in, Z feathering)
__global__ __launch_bounds__(256) void blend_in_kernel(const float* __restrict__ oldp, const float* __restrict__ newp,
                                                       float* __restrict__ out, const float* __restrict__ fz, int Z, long plane,
                                                       long total) {
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (long)gridDim.x * 256 * 4) {
    const int z = (int)((i / plane) % Z);
    const float f = fz[z];
    const float a = (f - 1.f) / f, bq = 1.f / f;
    const float4 o = *reinterpret_cast<const float4*>(oldp + i);
    const float4 n = *reinterpret_cast<const float4*>(newp + i);
    *reinterpret_cast<float4*>(out + i) = make_float4(o.x * a + n.x * bq, o.y * a + n.y * bq, o.z * a + n.z * bq, o.w * a + n.w * bq);
  }
}

static int ew_grid(long total) {
  int g = vsx_cdiv(total, 256L * 4 * 4);
  return g > 8192 ? 8192 : (g < 1 ? 1 : g);
}

/* K17 NormalizeSampled.__call__ (viscy_transforms/_normalize.py:72-80) on a (B, ...) fp32 batch with (B,) statistics. */
extern "C" int32_t vsx_normalize(const float* x, float* y, const float* sub, const float* div, int32_t B, int64_t per_sample,
                                 vsx_stream_t stream) {
  VSX_CHECK(x && y && sub && div && B > 0 && per_sample > 0 && per_sample % 4 == 0, "vsx_normalize: bad arguments (per-sample size must be a multiple of 4)");
  long total = (long)B * per_sample;
  hipLaunchKernelGGL(normalize_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, sub, div, (long)per_sample, total);
  VSX_LAUNCH_CHECK();
  return 0;
}
/* MinMaxSampled.__call__ (viscy_transforms/_normalize.py:124-134). */
extern "C" int32_t vsx_minmax_norm(const float* x, float* y, const float* lo, const float* hi, int32_t B, int64_t per_sample,
                                   vsx_stream_t stream) {
  VSX_CHECK(x && y && lo && hi && B > 0 && per_sample > 0 && per_sample % 4 == 0, "vsx_minmax_norm: bad arguments");
  long total = (long)B * per_sample;
  hipLaunchKernelGGL(minmax_norm_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, lo, hi, (long)per_sample, total);
  VSX_LAUNCH_CHECK();
  return 0;
}
/* per-sample min / max needed by MONAI AdjustContrast (K19); mn/mx must be pre-set to +inf / -inf. */
extern "C" int32_t vsx_sample_minmax(const float* x, float* mn, float* mx, int32_t B, int64_t per_sample, vsx_stream_t stream) {
  VSX_CHECK(x && mn && mx && B > 0 && per_sample > 0 && per_sample % 4 == 0, "vsx_sample_minmax: bad arguments");
  int gx = vsx_cdiv(per_sample, 256L * 4 * 8);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(sample_minmax_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, x, mn, mx, (long)per_sample);
  VSX_LAUNCH_CHECK();
  return 0;
}
/* per-sample sum / sum of squares (double; sums[B][2] pre-set to 0): the statistics AdjustContrast(retain_stats=True) restores. */
extern "C" int32_t vsx_sample_moments(const float* x, double* sums, int32_t B, int64_t per_sample, vsx_stream_t stream) {
  VSX_CHECK(x && sums && B > 0 && per_sample > 0 && per_sample % 4 == 0, "vsx_sample_moments: bad arguments");
  int gx = vsx_cdiv(per_sample, 256L * 4 * 8);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(sample_moments_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, x, sums, (long)per_sample);
  VSX_LAUNCH_CHECK();
  return 0;
}
/* K19-K21 fused: BatchedRandAdjustContrast (_adjust_contrast.py:54-86) → BatchedRandScaleIntensity
 * (_scale_intensity.py:59-77) → BatchedRandGaussianNoise (_noise.py:158-204) with injected per-sample parameters;
 * any of gamma / factor / noise may be NULL (stage skipped); gamma[b] <= 0 or nstd[b] < 0 = sample not selected;
 * invert: 0, or 3 = AdjustContrast(invert_image=True) (1 = the curve of -x without negating back: retain_stats path). */
extern "C" int32_t vsx_intensity_aug(const float* x, float* y, const float* mn, const float* mx, const float* gamma,
                                     const float* factor, const float* noise, const float* nstd, float nmean, int32_t invert,
                                     int32_t B, int64_t per_sample, vsx_stream_t stream) {
  VSX_CHECK(x && y && B > 0 && per_sample > 0 && per_sample % 4 == 0, "vsx_intensity_aug: bad arguments");
  VSX_CHECK(invert >= 0 && invert <= 3, "vsx_intensity_aug: invert is a 2-bit mask");
  VSX_CHECK(!gamma || (mn && mx), "vsx_intensity_aug: gamma needs per-sample min/max");
  VSX_CHECK((noise == nullptr) == (nstd == nullptr), "vsx_intensity_aug: noise and nstd come together");
  long total = (long)B * per_sample;
  hipLaunchKernelGGL(intensity_aug_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, mn, mx, gamma, factor,
                     noise, nstd, nmean, (int)invert, (long)per_sample, total);
  VSX_LAUNCH_CHECK();
  return 0;
}
/* K25 _blend_in (viscy_utils/callbacks/prediction_writer.py:74-111): fz[Z] = per-slice blend factor. */
extern "C" int32_t vsx_blend_in(const float* oldp, const float* newp, float* out, const float* fz, int32_t Z, int64_t plane,
                                int64_t total, vsx_stream_t stream) {
  VSX_CHECK(oldp && newp && out && fz && Z > 0 && plane > 0 && plane % 4 == 0 && total % 4 == 0, "vsx_blend_in: bad arguments");
  hipLaunchKernelGGL(blend_in_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, oldp, newp, out, fz, Z, (long)plane,
                     (long)total);
  VSX_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ K18: batched 3-D affine warp
// y[b,c,z,y,x] = trilinear( x[b,c], Minv[b] · (x,y,z,1) ), zero padding outside the volume.
// Minv [B][3][4] maps OUTPUT voxel coordinates (x, y, z) to INPUT voxel coordinates (x, y, z).
// Only the output region of interest [z0,z0+Do) x [y0,y0+Ho) x [x0,x0+Wo) is produced (the recipes follow the warp with a
// centre crop that keeps ~30 % of the voxels: viscy_data/hcs.py:694-695 chain, BatchedCenterSpatialCrop _crop.py:164-187).
// A workgroup owns a 32 x 8 output tile of one z-slice: under an in-plane rotation its input footprint is a compact
// ~30 x 30 patch per tap plane, so the 8 gathers per voxel hit L1 lines shared by the whole tile (a 256 x 1 row of outputs
// would sweep up to 180 input rows).  The matrix is wave-uniform (scalar registers).
// out-of-volume source coordinates as torch's grid_sample treats them (kornia warp_affine3d hands padding_mode and align_corners
// through, _affine.py:33-47): 1 "border" = clamp to [0, n-1]; 2 "reflection" with align_corners=True = mirror about 0 and n-1, then
// clamp; 3 (round 5) "reflection" with align_corners=False — what kornia's RandomAffine3D passes by default and the reference never
// overrides — = mirror about -0.5 and n-0.5 (the voxel EDGES), then clamp
__device__ __forceinline__ float warp_pad_coord(float v, int n, int pad) {
  if (pad == 2) {
    const float span = (float)(n - 1);
    if (span <= 0.f) return 0.f;
    v = fabsf(v);
    const float extra = fmodf(v, span);
    const int flips = (int)floorf(v / span);
    v = (flips & 1) ? span - extra : extra;
  } else if (pad == 3) {
    const float span = (float)n;
    v = fabsf(v + 0.5f);
    const float extra = fmodf(v, span);
    const int flips = (int)floorf(v / span);
    v = ((flips & 1) ? span - extra : extra) - 0.5f;
  }
  return fminf(fmaxf(v, 0.f), (float)(n - 1));
}

__global__ __launch_bounds__(256) void warp_affine3d_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            const float* __restrict__ Minv, int C, int D, int H, int W, int z0,
                                                            int y0, int x0, int Do, int Ho, int Wo, int tiles_x, int mode) {
  const int nearest = mode & 1, pad = mode >> 1;
  const int b = blockIdx.z, oz = blockIdx.y;
  const int ty_ = blockIdx.x / tiles_x, tx_ = blockIdx.x - ty_ * tiles_x;
  const int ox = tx_ * 32 + (threadIdx.x & 31), oy = ty_ * 8 + (threadIdx.x >> 5);
  if (ox >= Wo || oy >= Ho) return;
  const float* m = Minv + (size_t)b * 12;
  const float fx_ = (float)(ox + x0), fy_ = (float)(oy + y0), fz_ = (float)(oz + z0);
  float sx = m[0] * fx_ + m[1] * fy_ + m[2] * fz_ + m[3];
  float sy = m[4] * fx_ + m[5] * fy_ + m[6] * fz_ + m[7];
  float sz = m[8] * fx_ + m[9] * fy_ + m[10] * fz_ + m[11];
  if (pad) {
    sx = warp_pad_coord(sx, W, pad);
    sy = warp_pad_coord(sy, H, pad);
    sz = warp_pad_coord(sz, D, pad);
  }
  const size_t vol = (size_t)D * H * W, ovol = (size_t)Do * Ho * Wo;
  const float* xb = x + (size_t)b * C * vol;
  float* yb = y + (size_t)b * C * ovol + ((size_t)oz * Ho + oy) * Wo + ox;
  if (nearest) {
    const int ix = (int)nearbyintf(sx), iy = (int)nearbyintf(sy), iz = (int)nearbyintf(sz);
    const bool in = ix >= 0 && ix < W && iy >= 0 && iy < H && iz >= 0 && iz < D;
    const size_t off = in ? ((size_t)iz * H + iy) * W + ix : 0;
    for (int c = 0; c < C; ++c) yb[c * ovol] = in ? xb[c * vol + off] : 0.f;
    return;
  }
  const float fx = floorf(sx), fy = floorf(sy), fz = floorf(sz);
  const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
  const float tx = sx - fx, ty = sy - fy, tz = sz - fz;
  // per-corner weights, zeroed outside the volume; clamped addresses keep every load legal and unconditional
  const bool vx0 = ix >= 0 && ix < W, vx1 = ix + 1 >= 0 && ix + 1 < W;
  const bool vy0 = iy >= 0 && iy < H, vy1 = iy + 1 >= 0 && iy + 1 < H;
  const bool vz0 = iz >= 0 && iz < D, vz1 = iz + 1 >= 0 && iz + 1 < D;
  const float wx0 = vx0 ? 1.f - tx : 0.f, wx1 = vx1 ? tx : 0.f;
  const float wy0 = vy0 ? 1.f - ty : 0.f, wy1 = vy1 ? ty : 0.f;
  const float wz0 = vz0 ? 1.f - tz : 0.f, wz1 = vz1 ? tz : 0.f;
  const int cx0 = min(max(ix, 0), W - 1), cx1 = min(max(ix + 1, 0), W - 1);
  const int cy0 = min(max(iy, 0), H - 1), cy1 = min(max(iy + 1, 0), H - 1);
  const int cz0 = min(max(iz, 0), D - 1), cz1 = min(max(iz + 1, 0), D - 1);
  const size_t r00 = ((size_t)cz0 * H + cy0) * W, r01 = ((size_t)cz0 * H + cy1) * W;
  const size_t r10 = ((size_t)cz1 * H + cy0) * W, r11 = ((size_t)cz1 * H + cy1) * W;
  // same association as the reference expression: sum over (dz, dy, dx) of wz*wy*wx * v, accumulated in that order
  for (int c = 0; c < C; ++c) {
    const float* xc = xb + c * vol;
    const float a000 = xc[r00 + cx0], a001 = xc[r00 + cx1], a010 = xc[r01 + cx0], a011 = xc[r01 + cx1];
    const float a100 = xc[r10 + cx0], a101 = xc[r10 + cx1], a110 = xc[r11 + cx0], a111 = xc[r11 + cx1];
    float v = 0.f;
    v += (wx0 * wy0 * wz0) * a000;
    v += (wx1 * wy0 * wz0) * a001;
    v += (wx0 * wy1 * wz0) * a010;
    v += (wx1 * wy1 * wz0) * a011;
    v += (wx0 * wy0 * wz1) * a100;
    v += (wx1 * wy0 * wz1) * a101;
    v += (wx0 * wy1 * wz1) * a110;
    v += (wx1 * wy1 * wz1) * a111;
    yb[c * ovol] = v;
  }
}

// ------------------------------------------------------------------ K22: one axis of a separable filter
// y[i] = Σ_t taps[b][t] · x[i + (t - r)·stride]   (zero outside the axis; taps per sample; r = (k-1)/2)
__global__ __launch_bounds__(256) void conv1d_axis_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          const float* __restrict__ taps, int k, long per_sample, long stride,
                                                          int L, long total) {
  const int r = (k - 1) / 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int b = (int)(i / per_sample);
    const int pos = (int)((i / stride) % L);
    const float* tb = taps + (size_t)b * k;
    float acc = 0.f;
    for (int t = 0; t < k; ++t) {
      const int p = pos + t - r;
      if (p >= 0 && p < L) acc = fmaf(tb[t], x[i + (long)(t - r) * stride], acc);
    }
    y[i] = acc;
  }
}

/* K18 kornia warp_affine3d as used by BatchedRandAffined (viscy_transforms/_affine.py:33-47,358-393): trilinear (or
 * nearest) resampling; Minv[B][3][4] = output-voxel → input-voxel coordinates (x, y, z order).
 * mode: bit 0 = nearest; bits 1-2 = padding_mode (0 "zeros", 1 "border", 2 "reflection" about the voxel centres 0 / n - 1,
 * 3 "reflection" about the voxel edges -0.5 / n - 0.5 = grid_sample with align_corners=False; _affine.py:102-108).
 * vsx_warp_affine3d_roi produces only the output window [z0,z0+Do) x [y0,y0+Ho) x [x0,x0+Wo) of the full (D,H,W) frame:
 * the warp fused with the BatchedCenterSpatialCrop that follows it in the recipes (_crop.py:164-187). */
extern "C" int32_t vsx_warp_affine3d_roi(const float* x, float* y, const float* Minv, int32_t B, int32_t C, int32_t D,
                                         int32_t H, int32_t W, int32_t z0, int32_t y0, int32_t x0, int32_t Do, int32_t Ho,
                                         int32_t Wo, int32_t mode, vsx_stream_t stream) {
  VSX_CHECK(x && y && Minv && B > 0 && C > 0 && D > 0 && H > 0 && W > 0 && mode >= 0 && mode <= 7, "vsx_warp_affine3d: bad arguments");
  VSX_CHECK(Do > 0 && Ho > 0 && Wo > 0 && z0 >= 0 && y0 >= 0 && x0 >= 0 && z0 + Do <= D && y0 + Ho <= H && x0 + Wo <= W,
            "vsx_warp_affine3d: output window (%d,%d,%d)+(%d,%d,%d) outside the (%d,%d,%d) frame", z0, y0, x0, Do, Ho, Wo, D, H, W);
  VSX_CHECK(B <= 65535 && Do <= 65535, "vsx_warp_affine3d: B and the output depth must be <= 65535");
  const int tiles_x = vsx_cdiv(Wo, 32), tiles_y = vsx_cdiv(Ho, 8);
  hipLaunchKernelGGL(warp_affine3d_kernel, dim3(tiles_x * tiles_y, Do, B), dim3(256), 0, (hipStream_t)stream, x, y, Minv, C, D, H, W,
                     z0, y0, x0, Do, Ho, Wo, tiles_x, mode);
  VSX_LAUNCH_CHECK();
  return 0;
}
extern "C" int32_t vsx_warp_affine3d(const float* x, float* y, const float* Minv, int32_t B, int32_t C, int32_t D, int32_t H,
                                     int32_t W, int32_t mode, vsx_stream_t stream) {
  return vsx_warp_affine3d_roi(x, y, Minv, B, C, D, H, W, 0, 0, 0, D, H, W, mode, stream);
}
/* K22 one pass of kornia filter3d's separable form as used by BatchedRandGaussianSmooth
 * (viscy_transforms/_gaussian_smooth.py:141-167): per-sample 1-D taps along the axis with element stride `stride`
 * and length L, constant (zero) border. */
extern "C" int32_t vsx_conv1d_axis(const float* x, float* y, const float* taps, int32_t k, int32_t B, int64_t per_sample,
                                   int64_t stride, int32_t L, vsx_stream_t stream) {
  VSX_CHECK(x && y && taps && k > 0 && (k & 1) && B > 0 && per_sample > 0 && stride > 0 && L > 0, "vsx_conv1d_axis: bad arguments");
  long total = (long)B * per_sample;
  int g = vsx_cdiv(total, 256);
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(conv1d_axis_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, x, y, taps, k, (long)per_sample,
                     (long)stride, L, total);
  VSX_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ K23: BatchedRandWeightedCropd (_crop.py:263-386)
// (1) w2[b, y, x] = max(sum_{c, z} w[b, c, z, y, x], 0)
__global__ __launch_bounds__(256) void weight_map_yx_kernel(const float* __restrict__ w, float* __restrict__ out, int B, int CZ,
                                                            long plane4) {
  const long total = (long)B * plane4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / plane4, p = i - b * plane4;
    const float4* src = reinterpret_cast<const float4*>(w) + (size_t)b * CZ * plane4 + p;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < CZ; ++k) {
      const float4 v = src[(size_t)k * plane4];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
    reinterpret_cast<float4*>(out)[i] = a;
  }
}
// (2) window sums, separable: rows (window cx along x) then columns (window cy along y); double running sums
__global__ __launch_bounds__(256) void box_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int X, int cx) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const float* a = in + (size_t)r * X;
  float* o = out + (size_t)r * (X - cx + 1);
  double s = 0.0;
  for (int x = 0; x < cx; ++x) s += a[x];
  o[0] = (float)s;
  for (int x = 1; x + cx <= X; ++x) {
    s += (double)a[x + cx - 1] - (double)a[x - 1];
    o[x] = (float)s;
  }
}
__global__ __launch_bounds__(256) void box_cols_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int Y, int vx, int cy) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;  // (b, x)
  if (i >= (long)B * vx) return;
  const int b = (int)(i / vx), x = (int)(i % vx);
  const float* a = in + (size_t)b * Y * vx + x;
  float* o = out + (size_t)b * (Y - cy + 1) * vx + x;
  double s = 0.0;
  for (int y = 0; y < cy; ++y) s += a[(size_t)y * vx];
  o[0] = (float)s;
  for (int y = 1; y + cy <= Y; ++y) {
    s += (double)a[(size_t)(y + cy - 1) * vx] - (double)a[(size_t)(y - 1) * vx];
    o[(size_t)y * vx] = (float)s;
  }
}
// (3) inverse-CDF draw per sample: smallest i with sum_{j <= i} wp[b, j] > u[b] * total  (uniform over n when total == 0)
__global__ __launch_bounds__(256) void sample_index_kernel(const float* __restrict__ wp, const float* __restrict__ u, int* __restrict__ idx,
                                                           long n) {
  __shared__ double part[256];
  const int b = blockIdx.x, t = threadIdx.x;
  const float* a = wp + (size_t)b * n;
  const long chunk = (n + 255) / 256;
  const long lo = t * chunk, hi = lo + chunk < n ? lo + chunk : n;
  double s = 0.0;
  for (long i = lo; i < hi; ++i) s += a[i];
  part[t] = s;
  __syncthreads();
  if (t == 0) {
    double total = 0.0;
    for (int k = 0; k < 256; ++k) total += part[k];
    long pick;
    if (total <= 0.0) {
      pick = (long)((double)u[b] * (double)n);
    } else {
      const double target = (double)u[b] * total;
      double c = 0.0;
      int k = 0;
      while (k < 255 && c + part[k] <= target) c += part[k++];
      long i = k * chunk;
      const long e = i + chunk < n ? i + chunk : n;
      pick = e - 1;
      for (; i < e; ++i) {
        c += a[i];
        if (c > target) { pick = i; break; }
      }
    }
    if (pick >= n) pick = n - 1;
    if (pick < 0) pick = 0;
    idx[b] = (int)pick;
  }
}
// (4) crop gather: y[b, c, z, yy, xx] = x[b, c, z0[b] + z, y0[b] + yy, x0[b] + xx]
__global__ __launch_bounds__(256) void crop3d_kernel(const float* __restrict__ x, float* __restrict__ y, const int* __restrict__ starts,
                                                     int B, int C, int Z, int Y, int X, int cz, int cy, int cx) {
  const long total = (long)B * C * cz * cy * cx;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int xx = (int)(i % cx);
    long r = i / cx;
    const int yy = (int)(r % cy); r /= cy;
    const int zz = (int)(r % cz); r /= cz;
    const int c = (int)(r % C);
    const int b = (int)(r / C);
    const int z0 = starts[3 * b], y0 = starts[3 * b + 1], x0 = starts[3 * b + 2];
    y[i] = x[((((size_t)b * C + c) * Z + z0 + zz) * Y + y0 + yy) * X + x0 + xx];
  }
}

/* K23 pieces of viscy_transforms.BatchedRandWeightedCropd (_crop.py:263-386): the pooled window weights the reference
 * computes with sum(dim=(1,2)).clamp(min=0) + F.avg_pool2d((cy, cx), stride 1) (here: window SUMS — the same distribution),
 * an inverse-CDF draw per sample from caller-supplied uniforms (torch.multinomial's stream cannot be reproduced), and the
 * crop gather.  wpool: [B, (Y-cy+1)*(X-cx+1)]; tmp: B*Y*(X-cx+1) + B*Y*X floats of caller-owned scratch. */
extern "C" int32_t vsx_crop_weights(const float* w, float* wpool, float* tmp, int32_t B, int32_t CZ, int32_t Y, int32_t X,
                                    int32_t cy, int32_t cx, vsx_stream_t stream) {
  VSX_CHECK(w && wpool && tmp && B > 0 && CZ > 0 && Y >= cy && X >= cx && cy > 0 && cx > 0 && ((long)Y * X) % 4 == 0,
            "vsx_crop_weights: bad arguments (Y*X must be a multiple of 4, crop inside the image)");
  float* w2 = tmp;                              // [B, Y, X]
  float* rows = tmp + (size_t)B * Y * X;        // [B, Y, X - cx + 1]
  const long plane4 = (long)Y * X / 4;
  int g = vsx_cdiv((long)B * plane4, 256);
  if (g > 32768) g = 32768;
  hipLaunchKernelGGL(weight_map_yx_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, w, w2, B, CZ, plane4);
  hipLaunchKernelGGL(box_rows_kernel, dim3(vsx_cdiv((long)B * Y, 256)), dim3(256), 0, (hipStream_t)stream, w2, rows, B * Y, X, cx);
  hipLaunchKernelGGL(box_cols_kernel, dim3(vsx_cdiv((long)B * (X - cx + 1), 256)), dim3(256), 0, (hipStream_t)stream, rows, wpool, B, Y,
                     X - cx + 1, cy);
  VSX_LAUNCH_CHECK();
  return 0;
}
extern "C" int32_t vsx_sample_index(const float* wpool, const float* u, int32_t* idx, int32_t B, int64_t n, vsx_stream_t stream) {
  VSX_CHECK(wpool && u && idx && B > 0 && n > 0, "vsx_sample_index: bad arguments");
  hipLaunchKernelGGL(sample_index_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, wpool, u, idx, (long)n);
  VSX_LAUNCH_CHECK();
  return 0;
}
extern "C" int32_t vsx_crop3d(const float* x, float* y, const int32_t* starts, int32_t B, int32_t C, int32_t Z, int32_t Y, int32_t X,
                              int32_t cz, int32_t cy, int32_t cx, vsx_stream_t stream) {
  VSX_CHECK(x && y && starts && B > 0 && C > 0 && cz > 0 && cy > 0 && cx > 0 && cz <= Z && cy <= Y && cx <= X, "vsx_crop3d: bad arguments");
  long total = (long)B * C * cz * cy * cx;
  int g = vsx_cdiv(total, 256);
  if (g > 65536) g = 65536;
  hipLaunchKernelGGL(crop3d_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, x, y, starts, B, C, Z, Y, X, cz, cy, cx);
  VSX_LAUNCH_CHECK();
  return 0;
}
