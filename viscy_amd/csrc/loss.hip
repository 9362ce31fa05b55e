// MixedLoss = a1*L1 + a2*MSE + a3*(1 - MS-SSIM-2.5D)  (SURVEY §2.1 K15/K16), forward and backward.
// Reference: viscy_utils/losses/mixed_loss.py:42-69, viscy_utils/evaluation/metrics.py:174-349.
//
// The reference evaluates five (D,11,11) uniform-window means per scale as dense bf16 conv3d
// (605 MAC per voxel per conv).  A box filter is separable: here each 32x32 output tile stages the
// depth-summed terms of its 42x42 input footprint in LDS, then takes 11-tap running sums along X
// and Y — the kernel is HBM-bound (each input element is read ~1.7x, coalesced along X).
// Rounding points follow the reference bit-for-bit in intent: p, t, p*p, t*t, p*t are rounded to
// bf16 before summation, the window mean is bf16(k)·Σ rounded to bf16, everything after is fp32;
// in the backward the gradients w.r.t. the window means and the transposed-box-filter outputs are
// rounded to bf16 exactly where autograd casts them in the reference.
#include "vsx_common.h"
#include "../../include/vsx.h"

#define ST 32
#define SI 42
#define SLD 43

__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
  if (v >= 0.f)
    atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else
    atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// ------------------------------------------------------------------ pooling + max + L1/L2 sums
// one thread per 2x2 input quad (edge threads cover odd leftovers for max / L1 only)
__global__ __launch_bounds__(256) void loss_pool_kernel(const float* __restrict__ P, const float* __restrict__ T,
                                                        float* __restrict__ Po, float* __restrict__ To,
                                                        float* __restrict__ tmax, float* __restrict__ l1sum,
                                                        float* __restrict__ l2sum, int planes, int H, int W) {
  __shared__ float sh[4];
  const int Ho = H / 2, Wo = W / 2;
  const int Hc = (H + 1) / 2, Wc = (W + 1) / 2;
  const long total = (long)planes * Hc * Wc;
  float mx = -INFINITY, a1 = 0.f, a2 = 0.f;
  for (long gid = (long)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (long)gridDim.x * 256) {
    const int xq = (int)(gid % Wc);
    const long r = gid / Wc;
    const int yq = (int)(r % Hc);
    const long pl = r / Hc;
    float sp = 0.f, st = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int y = 2 * yq + i, x = 2 * xq + j;
        if (y < H && x < W) {
          const size_t off = ((size_t)pl * H + y) * W + x;
          const float p = P[off], t = T[off];
          sp += p;
          st += t;
          mx = fmaxf(mx, t);
          const float d = p - t;
          a1 += fabsf(d);
          a2 += d * d;
        }
      }
    if (Po && yq < Ho && xq < Wo) {
      const size_t oo = ((size_t)pl * Ho + yq) * Wo + xq;
      Po[oo] = 0.25f * sp;
      To[oo] = 0.25f * st;
    }
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0 && mx > -INFINITY) atomic_max_float(tmax, mx);
  if (l1sum) {
    float s1 = block_sum_256(a1, sh);
    float s2 = block_sum_256(a2, sh);
    if (threadIdx.x == 0) {
      atomicAdd(l1sum, s1);
      atomicAdd(l2sum, s2);
    }
  }
}

// ------------------------------------------------------------------ SSIM tile kernel (forward sums / backward d-mu maps)
struct SsimPix {
  float ssim, cs, dmx, dmxx, dmxy;
};

// XCD-aware tile order for the 3-D grids of the SSIM kernels: workgroups are dealt round-robin to the 8 XCDs in
// launch order (x fastest); remap so that each XCD owns a contiguous range of tiles = halo-sharing neighbours hit ONE L2
struct TileId { int x, y, z; };
__device__ __forceinline__ TileId xcd_tile() {
  const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
  unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  if ((total & 7u) == 0u) L = (L & 7u) * (total >> 3) + (L >> 3);
  TileId t;
  t.x = (int)(L % gx);
  t.y = (int)((L / gx) % gy);
  t.z = (int)(L / (gx * gy));
  return t;
}

template <bool BWD>
__device__ __forceinline__ SsimPix ssim_pixel(float mx, float my, float mxx, float myy, float mxy, float c1, float c2,
                                              float gs, float gc) {
  SsimPix o;
  const float sx = mxx - mx * mx, sy = myy - my * my, sxy = mxy - mx * my;
  const float A = 2.f * sxy + c2, Bd = sx + sy + c2;
  const float cs = A / Bd;
  const float num = 2.f * mx * my + c1, den = mx * mx + my * my + c1;
  const float L = num / den;
  o.cs = cs;
  o.ssim = L * cs;
  if (BWD) {
    const float dcs = gc + gs * L;
    const float dL = gs * cs;
    const float dsxy = dcs * 2.f / Bd;
    const float dsx = -dcs * cs / Bd;
    o.dmxx = dsx;
    o.dmxy = dsxy;
    o.dmx = dsx * (-2.f * mx) + dsxy * (-my) + dL * (2.f * my / den - num * 2.f * mx / (den * den));
  }
  return o;
}

template <bool BWD>
__global__ __launch_bounds__(256) void ssim_tile_kernel(const float* __restrict__ P, const float* __restrict__ T,
                                                        const float* __restrict__ tmax_p, int C, int D, int H, int W,
                                                        float* __restrict__ sum_ssim, float* __restrict__ sum_cs,
                                                        const float* __restrict__ coef, float* __restrict__ dmu, int last) {
  // The five window means are produced one quantity at a time through ONE pair of LDS planes (13 KB instead of 63 KB:
  // the kernel ran at 2 workgroups per CU and was latency-bound); the plane sums of this thread's halo pixels and the
  // finished means of its output pixels wait in registers.
  __shared__ float S[SI][SLD];
  __shared__ float R[SI][ST];
  __shared__ float sh[4];
  const TileId tl = xcd_tile();
  const int bc = tl.z;
  const int b = bc / C;
  const int oy0 = tl.y * ST, ox0 = tl.x * ST;
  const int Ho = H - 10, Wo = W - 10;
  const float kb = round_bf16(1.0f / (float)(D * 121));
  const float dr = tmax_p[0];
  const float c1 = (0.01f * dr) * (0.01f * dr), c2 = (0.03f * dr) * (0.03f * dr);

  constexpr int NIN = (SI * SI + 255) / 256;   // halo pixels per thread
  constexpr int NH = (SI * ST + 255) / 256;    // horizontal-pass outputs per thread
  constexpr int NOUT = (ST * ST + 255) / 256;  // output pixels per thread
  // depth-summed terms of this thread's halo pixels.  The depth loop is OUTSIDE the pixel loop: the 2 x NIN loads of a
  // slice are independent and issued together — with the loop nest the other way round (pixel outer, depth inner) the kernel
  // waited for memory NIN x D times per tile and ran at 1.5 TB/s, latency-bound (rocprofv3: 1.75 ms per full-resolution pass)
  float sq[NIN][5];
  size_t hoff[NIN];
  bool hin[NIN];
#pragma unroll
  for (int i = 0; i < NIN; ++i) {
    const int idx = threadIdx.x + i * 256;
    const int iy = idx / SI, ix = idx - iy * SI;
    const int gy = oy0 + iy, gx = ox0 + ix;
    hin[i] = idx < SI * SI && gy < H && gx < W;
    hoff[i] = ((size_t)bc * D * H + gy) * W + gx;
#pragma unroll
    for (int q = 0; q < 5; ++q) sq[i][q] = 0.f;
  }
  const size_t zstride = (size_t)H * W;
  for (int z = 0; z < D; ++z) {
    float pv[NIN], tv[NIN];
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      pv[i] = hin[i] ? P[hoff[i] + z * zstride] : 0.f;
      tv[i] = hin[i] ? T[hoff[i] + z * zstride] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const float p = pv[i], t = tv[i];
      sq[i][0] += round_bf16(p);
      sq[i][1] += round_bf16(t);
      sq[i][2] += round_bf16(p * p);
      sq[i][3] += round_bf16(t * t);
      sq[i][4] += round_bf16(p * t);
    }
  }
  float m[NOUT][5];
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    if (q) __syncthreads();  // previous quantity's vertical pass has finished reading R (and S)
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const int idx = threadIdx.x + i * 256;
      if (idx < SI * SI) S[idx / SI][idx % SI] = sq[i][q];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int idx = threadIdx.x + i * 256;
      if (idx < SI * ST) {
        const int iy = idx / ST, ox = idx - iy * ST;
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) a += S[iy][ox + k];
        R[iy][ox] = a;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
      const int idx = threadIdx.x + i * 256;
      const int oy = idx / ST, ox = idx - oy * ST;
      float a = 0.f;
      if (idx < ST * ST) {
#pragma unroll
        for (int k = 0; k < 11; ++k) a += R[oy + k][ox];
      }
      m[i][q] = round_bf16(kb * a);
    }
  }
  float acc_s = 0.f, acc_c = 0.f;
  float gs = 0.f, gc = 0.f;
  // coef == nullptr: the UNSCALED gradient field of this scale (the loss weights the SSIM map of the last scale and the
  // contrast map of the others, by one factor per sample that is known only after every scale has been summed:
  // ssim_bwd_in_kernel applies it) — lets the value pass and the map-gradient pass be one pass
  if (BWD) { gs = coef ? coef[2 * b] : (last ? 1.f : 0.f); gc = coef ? coef[2 * b + 1] : (last ? 0.f : 1.f); }
#pragma unroll
  for (int i = 0; i < NOUT; ++i) {
    const int idx = threadIdx.x + i * 256;
    const int oy = idx / ST, ox = idx - oy * ST;
    const int gy = oy0 + oy, gx = ox0 + ox;
    if (idx >= ST * ST || gy >= Ho || gx >= Wo) continue;
    SsimPix px = ssim_pixel<BWD>(m[i][0], m[i][1], m[i][2], m[i][3], m[i][4], c1, c2, gs, gc);
    if (BWD) {
      const size_t plane = (size_t)Ho * Wo;
      const size_t nbc = (size_t)gridDim.z;
      const size_t o = (size_t)bc * plane + (size_t)gy * Wo + gx;
      // the reference casts these gradients to bf16 AFTER the upstream factor is in them, and the three terms they feed
      // (G_x + 2 p G_xx + t G_xy) nearly cancel: rounding the unscaled field instead moves the final gradient by up to 8 %
      // of its maximum (measured) — bf16 noise of the same size as the reference's own, but not the SAME noise.  The
      // unscaled field is therefore kept in fp32 and rounded by ssim_bwd_in_kernel once the factor is applied.
      dmu[o] = coef ? round_bf16(px.dmx) : px.dmx;
      dmu[nbc * plane + o] = coef ? round_bf16(px.dmxx) : px.dmxx;
      dmu[2 * nbc * plane + o] = coef ? round_bf16(px.dmxy) : px.dmxy;
    }
    acc_s += px.ssim;
    acc_c += px.cs;
  }
  if (sum_ssim != nullptr) {
    float s = block_sum_256(acc_s, sh);
    float c = block_sum_256(acc_c, sh);
    if (threadIdx.x == 0) {
      atomicAdd(sum_ssim + b, s);
      atomicAdd(sum_cs + b, c);
    }
  }
}

// ------------------------------------------------------------------ backward: transposed box filter + chain rule to the stack
// G_q(y,x) = bf16( kb * Σ_{oy∈[y-10,y], ox∈[x-10,x]} dmu_q(oy,ox) );  dP(z,y,x) = G_x + 2 p G_xx + t G_xy
//            + 0.25 * dPnext(z, y/2, x/2)  + l1c * sign(p - t) + l2c * 2 (p - t)
__global__ __launch_bounds__(256) void ssim_bwd_in_kernel(const float* __restrict__ P, const float* __restrict__ T,
                                                          const float* __restrict__ dmu, const float* __restrict__ dPn,
                                                          float* __restrict__ dP, int D, int H, int W, float l1c_,
                                                          float l2c_, const float* __restrict__ gout_p, int has_ssim,
                                                          const float* __restrict__ coef, int C, int last) {
  __shared__ float S[3][SI][SLD];
  __shared__ float R[3][SI][ST];
  const TileId tl = xcd_tile();
  const int bc = tl.z;
  const int iy0 = tl.y * ST, ix0 = tl.x * ST;
  const int Ho = H - 10, Wo = W - 10;
  const float kb = round_bf16(1.0f / (float)(D * 121));
  const float gsc = gout_p ? gout_p[0] : 1.f;
  const float l1c = l1c_ * gsc, l2c = l2c_ * gsc;
  if (has_ssim) {
    const size_t plane = (size_t)Ho * Wo;
    const size_t nbc = (size_t)gridDim.z;
    // dmu of an unscaled field (see ssim_tile_kernel): this sample's factor for the map the loss uses at this scale
    const float gsample = coef ? coef[2 * (bc / C) + (last ? 0 : 1)] : 1.f;
    constexpr int NIN = (SI * SI + 255) / 256;
    float hv[NIN][3];
#pragma unroll
    for (int i = 0; i < NIN; ++i) {  // every halo load of the thread in flight before the first LDS store
      const int idx = threadIdx.x + i * 256;
      const int iy = idx / SI, ix = idx - iy * SI;
      const int oy = iy0 - 10 + iy, ox = ix0 - 10 + ix;
      float v0 = 0.f, v1 = 0.f, v2 = 0.f;
      if (idx < SI * SI && oy >= 0 && oy < Ho && ox >= 0 && ox < Wo) {
        const size_t o = (size_t)bc * plane + (size_t)oy * Wo + ox;
        v0 = dmu[o];
        v1 = dmu[nbc * plane + o];
        v2 = dmu[2 * nbc * plane + o];
      }
      hv[i][0] = v0; hv[i][1] = v1; hv[i][2] = v2;
    }
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const int idx = threadIdx.x + i * 256;
      if (idx < SI * SI) {
        const int iy = idx / SI, ix = idx - iy * SI;
        float v0 = hv[i][0], v1 = hv[i][1], v2 = hv[i][2];
        if (coef) { v0 = round_bf16(v0 * gsample); v1 = round_bf16(v1 * gsample); v2 = round_bf16(v2 * gsample); }
        S[0][iy][ix] = v0; S[1][iy][ix] = v1; S[2][iy][ix] = v2;
      }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < SI * ST; idx += 256) {
      const int iy = idx / ST, x = idx - iy * ST;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) a += S[q][iy][x + k];
        R[q][iy][x] = a;
      }
    }
    __syncthreads();
  }
  const int Hn = H / 2, Wn = W / 2;
  constexpr int NOUT = (ST * ST + 255) / 256;
  float G[NOUT][3];
  size_t off0[NOUT], offn[NOUT];
  bool live[NOUT], pooled[NOUT];
#pragma unroll
  for (int i = 0; i < NOUT; ++i) {
    const int idx = threadIdx.x + i * 256;
    const int y = idx / ST, x = idx - y * ST;
    const int gy = iy0 + y, gx = ix0 + x;
    live[i] = idx < ST * ST && gy < H && gx < W;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      float a = 0.f;
      if (has_ssim && live[i]) {
#pragma unroll
        for (int k = 0; k < 11; ++k) a += R[q][y + k][x];
      }
      G[i][q] = round_bf16(kb * a);
    }
    pooled[i] = live[i] && dPn != nullptr && (gy >> 1) < Hn && (gx >> 1) < Wn;
    off0[i] = ((size_t)bc * D * H + gy) * W + gx;
    offn[i] = ((size_t)bc * D * Hn + (gy >> 1)) * Wn + (gx >> 1);
  }
  // depth loop outside the pixel loop: the loads of a slice are issued together (see ssim_tile_kernel)
  const size_t zs = (size_t)H * W, zsn = (size_t)Hn * Wn;
  for (int z = 0; z < D; ++z) {
    float pv[NOUT], tv[NOUT], nv[NOUT];
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
      pv[i] = live[i] ? P[off0[i] + z * zs] : 0.f;
      tv[i] = live[i] ? T[off0[i] + z * zs] : 0.f;
      nv[i] = pooled[i] ? dPn[offn[i] + z * zsn] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
      const float p = pv[i], t = tv[i];
      float g = G[i][0] + 2.f * p * G[i][1] + t * G[i][2];
      if (pooled[i]) g += 0.25f * nv[i];
      const float d = p - t;
      if (l1c != 0.f) g += d > 0.f ? l1c : (d < 0.f ? -l1c : 0.f);
      if (l2c != 0.f) g += 2.f * l2c * d;
      if (live[i]) dP[off0[i] + z * zs] = g;
    }
  }
}

// ------------------------------------------------------------------ scalar finalisation
// vals[s][b] = clamp(mean cs (s<4) or mean ssim (s=4), 1e-4); ms = mean_b Π_s vals^beta_s
// loss = a1*l1 + a2*l2 + a3*(1 - ms);  coef[s][b] = {d loss / d ssim_pixel, d loss / d cs_pixel} (already / Npix_s)
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ sum_ssim,
                                                            const float* __restrict__ sum_cs,
                                                            const float* __restrict__ l1sum, const float* __restrict__ l2sum,
                                                            const float* __restrict__ npix, float nelem, int B, int nscale,
                                                            float a1, float a2, float a3,
                                                            const float* __restrict__ gout_p, float* __restrict__ loss,
                                                            float* __restrict__ coef, float* __restrict__ ms_out) {
  // one thread per batch sample (strided), block reduction of the per-sample MS-SSIM products
  __shared__ float sh[4];
  const float betas[5] = {0.0448f, 0.2856f, 0.3001f, 0.2363f, 0.1333f};
  const float gout = gout_p ? gout_p[0] : 1.f;
  float acc = 0.f;
  if (a3 != 0.f) {
    for (int b = threadIdx.x; b < B; b += 256) {
      float v[5];
      bool clamped[5];
      float prod = 1.f;
      for (int s = 0; s < nscale; ++s) {
        float raw = (s == nscale - 1 ? sum_ssim[s * B + b] : sum_cs[s * B + b]) / npix[s];
        clamped[s] = raw < 1e-4f;
        v[s] = clamped[s] ? 1e-4f : raw;
        prod *= powf(v[s], betas[s]);
      }
      acc += prod;
      for (int s = 0; s < nscale; ++s) {
        float g = clamped[s] ? 0.f : -a3 * gout / (float)B * betas[s] * prod / v[s] / npix[s];
        coef[(s * B + b) * 2 + 0] = (s == nscale - 1) ? g : 0.f;
        coef[(s * B + b) * 2 + 1] = (s == nscale - 1) ? 0.f : g;
      }
    }
  }
  const float tot = block_sum_256(acc, sh);
  if (threadIdx.x == 0) {
    const float ms_mean = a3 != 0.f ? tot / (float)B : 0.f;
    float l = 0.f;
    if (a1 != 0.f) l += a1 * l1sum[0] / nelem;
    if (a2 != 0.f) l += a2 * l2sum[0] / nelem;
    if (a3 != 0.f) l += a3 * (1.f - ms_mean);
    loss[0] = l;
    if (ms_out) ms_out[0] = ms_mean;
  }
}

/* avg_pool3d(·,(1,2,2)) of preds/target (metrics.py:340-341), target.max() of the INPUT planes
 * (metrics.py:298, data_range) and the L1 / L2 sums of mixed_loss.py:58-63 in one pass.
 * P/T: [planes, H, W] fp32; Po/To (may be NULL): [planes, H/2, W/2]; tmax must be pre-set to -inf. */
extern "C" int32_t vsx_loss_pool(const float* P, const float* T, float* Po, float* To, float* tmax, float* l1sum,
                                 float* l2sum, int32_t planes, int32_t H, int32_t W, vsx_stream_t stream) {
  VSX_CHECK(P && T && tmax && planes > 0 && H > 0 && W > 0, "vsx_loss_pool: bad arguments");
  VSX_CHECK((Po == nullptr) == (To == nullptr) && (l1sum == nullptr) == (l2sum == nullptr), "vsx_loss_pool: pointer pairs");
  long total = (long)planes * ((H + 1) / 2) * ((W + 1) / 2);
  int nblk = vsx_cdiv(total, 256);
  if (nblk > 2048) nblk = 2048;  // grid-stride: bounds the same-address atomics (max / L1 / L2 sums) to 2048 per launch
  hipLaunchKernelGGL(loss_pool_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, P, T, Po, To, tmax,
                     l1sum, l2sum, planes, H, W);
  VSX_LAUNCH_CHECK();
  return 0;
}

/* one scale of ssim_25d (metrics.py:272-305): per-sample sums of the SSIM and contrast-sensitivity maps.
 * P/T: [B, C, D, H, W] fp32 (this scale); sum_ssim / sum_cs: [B] (+=). */
extern "C" int32_t vsx_ssim_scale_fwd(const float* P, const float* T, const float* tmax, float* sum_ssim, float* sum_cs,
                                      int32_t B, int32_t C, int32_t D, int32_t H, int32_t W, vsx_stream_t stream) {
  VSX_CHECK(P && T && tmax && sum_ssim && sum_cs, "vsx_ssim_scale_fwd: null pointer");
  VSX_CHECK(H >= 11 && W >= 11 && D >= 1, "vsx_ssim_scale_fwd: plane %dx%d smaller than the 11x11 window", H, W);
  dim3 grid(vsx_cdiv(W - 10, ST), vsx_cdiv(H - 10, ST), B * C);
  hipLaunchKernelGGL(ssim_tile_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, P, T, tmax, C, D, H, W, sum_ssim,
                     sum_cs, (const float*)nullptr, (float*)nullptr, 0);
  VSX_LAUNCH_CHECK();
  return 0;
}

/* backward of one scale: coef [B][2] = upstream gradient per SSIM / CS map pixel; dmu scratch
 * [3][B*C][H-10][W-10]; dPnext (may be NULL): gradient w.r.t. the pooled stack of the next scale;
 * dP: [B, C, D, H, W] gradient w.r.t. this scale's preds (written, not accumulated). */
extern "C" int32_t vsx_ssim_scale_bwd(const float* P, const float* T, const float* tmax, const float* coef, float* dmu,
                                      const float* dPnext, float* dP, int32_t B, int32_t C, int32_t D, int32_t H,
                                      int32_t W, float l1c, float l2c, const float* gout, int32_t has_ssim,
                                      vsx_stream_t stream) {
  VSX_CHECK(P && T && dP, "vsx_ssim_scale_bwd: null pointer");
  if (has_ssim) {
    VSX_CHECK(tmax && coef && dmu && H >= 11 && W >= 11, "vsx_ssim_scale_bwd: SSIM term needs tmax/coef/dmu and a >=11 plane");
    dim3 g1(vsx_cdiv(W - 10, ST), vsx_cdiv(H - 10, ST), B * C);
    hipLaunchKernelGGL(ssim_tile_kernel<true>, g1, dim3(256), 0, (hipStream_t)stream, P, T, tmax, C, D, H, W,
                       (float*)nullptr, (float*)nullptr, coef, dmu, 0);
    VSX_LAUNCH_CHECK();
  }
  dim3 g2(vsx_cdiv(W, ST), vsx_cdiv(H, ST), B * C);
  hipLaunchKernelGGL(ssim_bwd_in_kernel, g2, dim3(256), 0, (hipStream_t)stream, P, T, dmu, dPnext, dP, D, H, W, l1c, l2c,
                     gout, has_ssim, (const float*)nullptr, C, 0);
  VSX_LAUNCH_CHECK();
  return 0;
}

/* Training forward of one scale: the sums of vsx_ssim_scale_fwd AND, in the same pass over the stack, the gradient field
 * w.r.t. the window means for a unit upstream gradient on the map this scale contributes (SSIM for the last scale,
 * contrast for the others): dmu [3][B*C][H-10][W-10], kept until the backward.  The per-sample factor (known once all
 * scales are summed: vsx_loss_finalize) is applied by vsx_ssim_scale_bwd_in — one pass over the full-resolution stack less
 * than vsx_ssim_scale_fwd + vsx_ssim_scale_bwd.  The field is stored unrounded (fp32); vsx_ssim_scale_bwd_in rounds it to
 * bf16 after the factor, where the reference's autograd casts it. */
extern "C" int32_t vsx_ssim_scale_fwd_dmu(const float* P, const float* T, const float* tmax, float* sum_ssim, float* sum_cs,
                                          float* dmu, int32_t B, int32_t C, int32_t D, int32_t H, int32_t W, int32_t last,
                                          vsx_stream_t stream) {
  VSX_CHECK(P && T && tmax && sum_ssim && sum_cs && dmu, "vsx_ssim_scale_fwd_dmu: null pointer");
  VSX_CHECK(H >= 11 && W >= 11 && D >= 1, "vsx_ssim_scale_fwd_dmu: plane %dx%d smaller than the 11x11 window", H, W);
  dim3 grid(vsx_cdiv(W - 10, ST), vsx_cdiv(H - 10, ST), B * C);
  hipLaunchKernelGGL(ssim_tile_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, P, T, tmax, C, D, H, W, sum_ssim,
                     sum_cs, (const float*)nullptr, dmu, last ? 1 : 0);
  VSX_LAUNCH_CHECK();
  return 0;
}

/* Backward of one scale from a stored unscaled field: dP = transposed box filter of coef_b * dmu + chain to the stack +
 * 0.25 * dPnext + L1 / L2 terms (as vsx_ssim_scale_bwd).  coef [B][2] from vsx_loss_finalize for this scale. */
extern "C" int32_t vsx_ssim_scale_bwd_in(const float* P, const float* T, const float* dmu, const float* coef,
                                         const float* dPnext, float* dP, int32_t B, int32_t C, int32_t D, int32_t H, int32_t W,
                                         float l1c, float l2c, const float* gout, int32_t has_ssim, int32_t last,
                                         vsx_stream_t stream) {
  VSX_CHECK(P && T && dP && B > 0 && C > 0, "vsx_ssim_scale_bwd_in: null pointer");
  if (has_ssim) VSX_CHECK(dmu && coef && H >= 11 && W >= 11, "vsx_ssim_scale_bwd_in: SSIM term needs dmu / coef and a >=11 plane");
  dim3 g2(vsx_cdiv(W, ST), vsx_cdiv(H, ST), B * C);
  hipLaunchKernelGGL(ssim_bwd_in_kernel, g2, dim3(256), 0, (hipStream_t)stream, P, T, dmu, dPnext, dP, D, H, W, l1c, l2c,
                     gout, has_ssim, coef, C, last ? 1 : 0);
  VSX_LAUNCH_CHECK();
  return 0;
}

/* ms_ssim_25d combination (metrics.py:326-349, clamp=True) + MixedLoss weights (mixed_loss.py:56-69).
 * sums: [nscale][B]; npix: [nscale] (= C*(H_s-10)*(W_s-10)); coef out: [nscale][B][2]. */
extern "C" int32_t vsx_loss_finalize(const float* sum_ssim, const float* sum_cs, const float* l1sum, const float* l2sum,
                                     const float* npix, float nelem, int32_t B, int32_t nscale, float a1, float a2,
                                     float a3, const float* gout, float* loss, float* coef, float* ms_out,
                                     vsx_stream_t stream) {
  VSX_CHECK(loss && B > 0 && nscale >= 1 && nscale <= 5, "vsx_loss_finalize: bad arguments");
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sum_ssim, sum_cs, l1sum, l2sum,
                     npix, nelem, B, nscale, a1, a2, a3, gout, loss, coef, ms_out);
  VSX_LAUNCH_CHECK();
  return 0;
}
