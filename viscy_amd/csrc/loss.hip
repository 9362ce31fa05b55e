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
      asm volatile("" : "+v"(m[i][q]));  // keep the sums out of the bounds-checked branch below (see ssim_tile_fused_kernel)
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

// ------------------------------------------------------------------ training forward of one scale in ONE pass over the stack
// ssim_tile_kernel<true> (sums + unscaled gradient field) that also does what loss_pool_kernel did in a pass of its own:
// the 2x2 average pooling into the NEXT scale's stacks, that scale's data range (max of the pooled target) and, at full
// resolution, the L1 / L2 sums.  A thread owns whole 2x2 quads of the tile's 42x42 footprint (21x21 quads, one per thread
// of a 512-thread workgroup), so pooling is register arithmetic on values the window sums need anyway; a tile pools / sums the 32x32 pixels at
// its origin — the last tile of a row / column everything up to the edge (at most 42: H - 10 <= 32 * tiles) — so that every
// input pixel is counted exactly once.  
// Same-address atomics serialise at ~10 ns each: 65 536 workgroups adding into ONE float cost more than the pass itself
// (measured: 8.5 ms instead of 2.6).  The one-pass kernel's scalar results (data range, L1 / L2 sums) are therefore SLOTTED:
// VSX_LOSS_SLOTS partial values, VSX_LOSS_SLOT_STRIDE floats (one 128-byte line) apart, the workgroup picks slot = id % slots;
// readers combine the slots (64 loads + a wave reduction).
#define LSLOTS VSX_LOSS_SLOTS
#define LSTRIDE VSX_LOSS_SLOT_STRIDE
// Buffer addressing for the loss kernels: a scalar plane descriptor + a 32-bit lane byte offset; the hardware range check
// returns 0 / drops the store for lanes whose offset is VSX_BUF_OOB (outside the image), so no access needs a branch, a select
// or a 64-bit lane pointer.
#define VSX_BUF_OOB 0x80000000u
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const float* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_ld(__amdgpu_buffer_rsrc_t r, uint32_t voff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0));
}
typedef uint32_t vsx_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 buf_ld2(__amdgpu_buffer_rsrc_t r, uint32_t voff) {
  const vsx_u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, 0, 0);
  return make_float2(__uint_as_float(w.x), __uint_as_float(w.y));
}
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, uint32_t voff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)voff, 0, 0);
}
#define FT 512  // threads per workgroup of the one-pass kernel: one quad of the footprint and two outputs per thread
__device__ __forceinline__ float block_sum_ft(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float a = 0.f;
#pragma unroll
  for (int w = 0; w < FT / 64; ++w) a += sh[w];
  return a;
}
// VEC: even W and 8-byte aligned stacks — the two pixels of a quad row are one 8-byte load (a wave reads whole lines).
template <bool VEC>
__global__ __launch_bounds__(FT, 8) void ssim_tile_fused_kernel(const float* __restrict__ P, const float* __restrict__ T,
                                                                 const float* __restrict__ tmax_p, int C, int D, int H, int W,
                                                                 float* __restrict__ sum_ssim, float* __restrict__ sum_cs,
                                                                 float* __restrict__ dmu, int last, float* __restrict__ Po,
                                                                 float* __restrict__ To, float* __restrict__ tmax_next,
                                                                 float* __restrict__ l1sum, float* __restrict__ l2sum) {
  __shared__ float S[SI][SLD];
  __shared__ float R[SI][ST];
  __shared__ float sh[FT / 64];
  const TileId tl = xcd_tile();
  const int bc = tl.z;
  const int b = bc / C;
  const int oy0 = tl.y * ST, ox0 = tl.x * ST;
  const int Ho = H - 10, Wo = W - 10;
  const float kb = round_bf16(1.0f / (float)(D * 121));
  const float dr = wave_max(tmax_p[(threadIdx.x & 63) * LSTRIDE]);  // slotted data range (LSLOTS == 64 == one wave)
  const float c1 = (0.01f * dr) * (0.01f * dr), c2 = (0.03f * dr) * (0.03f * dr);
  const bool last_y = tl.y == (int)gridDim.y - 1, last_x = tl.x == (int)gridDim.x - 1;

  constexpr int QN = SI / 2;                      // quads per footprint side
  constexpr int NQR = (QN * QN + FT - 1) / FT;      // quad rounds per thread
  constexpr int NIN = NQR * 4;                    // footprint pixels per thread
  constexpr int NOUT = (ST * ST + FT - 1) / FT;
  static_assert(NOUT == 2 && FT == 16 * ST && SI * (ST / 4) <= FT, "row-pair mapping of the vertical pass");
  float sq[NIN][5];
  uint32_t hoff[NIN], poff[NQR];  // byte offsets into a plane (VSX_BUF_OOB outside the image / for quads this tile does not pool)
  bool own[NQR], pool[NQR];
  const int Hn = H / 2, Wn = W / 2;
#pragma unroll
  for (int r = 0; r < NQR; ++r) {
    const int qi = threadIdx.x + r * FT;
    const int qy = qi / QN, qx = qi - qy * QN;
    const bool valid = qi < QN * QN;
    own[r] = valid && (qy < ST / 2 || last_y) && (qx < ST / 2 || last_x);
    const int py = (oy0 >> 1) + qy, px = (ox0 >> 1) + qx;
    pool[r] = own[r] && Po != nullptr && py < Hn && px < Wn;
    poff[r] = pool[r] ? (uint32_t)(py * Wn + px) * 4u : VSX_BUF_OOB;
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) {
      const int gy = oy0 + 2 * qy + (ab >> 1), gx = ox0 + 2 * qx + (ab & 1);
      hoff[r * 4 + ab] = valid && gy < H && gx < W ? (uint32_t)(gy * W + gx) * 4u : VSX_BUF_OOB;
#pragma unroll
      for (int q = 0; q < 5; ++q) sq[r * 4 + ab][q] = 0.f;
    }
  }
  const size_t zs = (size_t)H * W, zsn = (size_t)Hn * Wn;
  const float* Pg = P + (size_t)bc * D * zs;
  const float* Tg = T + (size_t)bc * D * zs;
  float* Pog = Po ? Po + (size_t)bc * D * zsn : dmu;  // no pooling: zero-sized descriptors, every store dropped
  float* Tog = To ? To + (size_t)bc * D * zsn : dmu;
  const size_t pbytes = Po ? zsn * 4 : 0;
  float mx = -INFINITY, a1 = 0.f, a2 = 0.f;
  const bool do_l1 = l1sum != nullptr;
  auto plane = [&](int z) {
    float pv[NIN], tv[NIN];
    const __amdgpu_buffer_rsrc_t rp = buf_rsrc(Pg + z * zs, zs * 4), rt = buf_rsrc(Tg + z * zs, zs * 4);
#pragma unroll
    for (int i = 0; i < NIN; i += 2) {
      if constexpr (VEC) {  // even W: the two pixels of a quad row are inside or outside together
        const float2 vp = buf_ld2(rp, hoff[i]), vt = buf_ld2(rt, hoff[i]);
        pv[i] = vp.x; pv[i + 1] = vp.y;
        tv[i] = vt.x; tv[i + 1] = vt.y;
      } else {
        pv[i] = buf_ld(rp, hoff[i]); pv[i + 1] = buf_ld(rp, hoff[i + 1]);
        tv[i] = buf_ld(rt, hoff[i]); tv[i + 1] = buf_ld(rt, hoff[i + 1]);
      }
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
    const __amdgpu_buffer_rsrc_t wp = buf_rsrc(Pog + z * zsn, pbytes), wt = buf_rsrc(Tog + z * zsn, pbytes);
#pragma unroll
    for (int r = 0; r < NQR; ++r) {
      // same summation order as loss_pool_kernel: (0,0), (0,1), (1,0), (1,1); quads this tile does not pool: stores dropped
      const float sp = ((pv[r * 4] + pv[r * 4 + 1]) + pv[r * 4 + 2]) + pv[r * 4 + 3];
      const float st = ((tv[r * 4] + tv[r * 4 + 1]) + tv[r * 4 + 2]) + tv[r * 4 + 3];
      buf_st(wp, poff[r], 0.25f * sp);
      buf_st(wt, poff[r], 0.25f * st);
      if (pool[r]) mx = fmaxf(mx, 0.25f * st);
      if (do_l1 && own[r]) {  // pixels outside the image were read as p = t = 0
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
          const float d = pv[r * 4 + ab] - tv[r * 4 + ab];
          a1 += fabsf(d);
          a2 += d * d;
        }
      }
    }
  };
  // a rolled depth loop: with every plane's loads in flight at once (compile-time depth) the kernel needs 94 registers and runs
  // 2 workgroups per CU — measured 2.6 ms per forward against 2.1 ms for this loop at 3 workgroups per CU
  for (int z = 0; z < D; ++z) plane(z);
  float m[NOUT][5];
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    if (q) __syncthreads();
#pragma unroll
    for (int r = 0; r < NQR; ++r) {
      const int qi = threadIdx.x + r * FT;
      const int qy = qi / QN, qx = qi - qy * QN;
      if (qi < QN * QN) {
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) S[2 * qy + (ab >> 1)][2 * qx + (ab & 1)] = sq[r * 4 + ab][q];
      }
    }
    __syncthreads();
    // 11-tap sums from registers: a thread reads the 14 (12) values four (two) neighbouring outputs share once instead of
    // 11 per output — the LDS pipe was as busy as HBM (55 -> 26 reads per thread and quantity); same summation order
    if (threadIdx.x < SI * (ST / 4)) {
      const int iy = threadIdx.x / (ST / 4), sg = threadIdx.x % (ST / 4);
      float v[14];
#pragma unroll
      for (int k = 0; k < 14; ++k) v[k] = S[iy][sg * 4 + k];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) a += v[j + k];
        R[iy][sg * 4 + j] = a;
      }
    }
    __syncthreads();
    {
      const int ox = threadIdx.x & (ST - 1), pr = threadIdx.x / ST;  // outputs (2 pr, ox), (2 pr + 1, ox)
      float v[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) v[k] = R[2 * pr + k][ox];
#pragma unroll
      for (int i = 0; i < NOUT; ++i) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) a += v[i + k];
        m[i][q] = round_bf16(kb * a);
        // pin the mean here: hipcc otherwise sinks the 11-term sums of all five quantities into the bounds-checked pixel
        // formula below and keeps the raw LDS values alive until then (spills at any useful occupancy)
        asm volatile("" : "+v"(m[i][q]));
      }
    }
  }
  float acc_s = 0.f, acc_c = 0.f;
  const float gs = last ? 1.f : 0.f, gc = last ? 0.f : 1.f;
  const size_t plane_o = (size_t)Ho * Wo, nbc = (size_t)gridDim.z;
#pragma unroll
  for (int i = 0; i < NOUT; ++i) {
    const int oy = 2 * (threadIdx.x / ST) + i, ox = threadIdx.x & (ST - 1);
    const int gy = oy0 + oy, gx = ox0 + ox;
    if (gy >= Ho || gx >= Wo) continue;
    SsimPix px = ssim_pixel<true>(m[i][0], m[i][1], m[i][2], m[i][3], m[i][4], c1, c2, gs, gc);
    const uint32_t o = (uint32_t)(gy * Wo + gx) * 4u;  // unscaled field, fp32 (see ssim_tile_kernel)
    buf_st(buf_rsrc(dmu + (size_t)bc * plane_o, plane_o * 4), o, px.dmx);
    buf_st(buf_rsrc(dmu + (nbc + bc) * plane_o, plane_o * 4), o, px.dmxx);
    buf_st(buf_rsrc(dmu + (2 * nbc + bc) * plane_o, plane_o * 4), o, px.dmxy);
    acc_s += px.ssim;
    acc_c += px.cs;
  }
  {
    const float s = block_sum_ft(acc_s, sh);
    const float c = block_sum_ft(acc_c, sh);
    if (threadIdx.x == 0) {
      atomicAdd(sum_ssim + b, s);
      atomicAdd(sum_cs + b, c);
    }
  }
  const unsigned slot = (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) % LSLOTS;
  if (tmax_next != nullptr) {
    mx = wave_max(mx);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
      for (int w = 1; w < FT / 64; ++w) mx = fmaxf(mx, sh[w]);
      if (mx > -INFINITY) atomic_max_float(tmax_next + slot * LSTRIDE, mx);
    }
  }
  if (do_l1) {
    const float s1 = block_sum_ft(a1, sh);
    const float s2 = block_sum_ft(a2, sh);
    if (threadIdx.x == 0) {
      atomicAdd(l1sum + slot * LSTRIDE, s1);
      atomicAdd(l2sum + slot * LSTRIDE, s2);
    }
  }
}

// max of a stack (the data range of scale 0 for the one-pass training forward above)
__global__ __launch_bounds__(256) void loss_tmax_kernel(const float* __restrict__ T, long n, float* __restrict__ tmax) {
  float mx = -INFINITY;
  // 16-byte loads from the first aligned element on; the (at most 3) elements before it and the tail go to workgroup 0
  const long head = (long)((16 - (reinterpret_cast<uintptr_t>(T) & 15)) & 15) >> 2;
  const long h = head < n ? head : n;
  const long n4 = (n - h) >> 2;
  const float4* T4 = reinterpret_cast<const float4*>(T + h);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = T4[i];
    mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
  }
  if (blockIdx.x == 0) {
    if ((long)threadIdx.x < h) mx = fmaxf(mx, T[threadIdx.x]);
    const long t0 = h + (n4 << 2);
    if (t0 + (long)threadIdx.x < n) mx = fmaxf(mx, T[t0 + threadIdx.x]);
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0 && mx > -INFINITY) atomic_max_float(tmax + ((blockIdx.x * 4 + (threadIdx.x >> 6)) % LSLOTS) * LSTRIDE, mx);
}

// ------------------------------------------------------------------ backward: transposed box filter + chain rule to the stack
// G_q(y,x) = bf16( kb * Σ_{oy∈[y-10,y], ox∈[x-10,x]} dmu_q(oy,ox) );  dP(z,y,x) = G_x + 2 p G_xx + t G_xy
//            + 0.25 * dPnext(z, y/2, x/2)  + l1c * sign(p - t) + l2c * 2 (p - t)
// Occupancy is what hides the latency here (measured: preloading all planes at 116 registers / 38 KB of LDS = 4 workgroups
// per CU ran 1.28 ms at full resolution, the 64-register version with three S / R plane pairs 1.79 ms): the three gradient
// fields go through ONE S / R pair one after the other (13 KB), the depth loop keeps one plane of loads in flight ahead of
// the plane it is finishing, and every global access is a BUFFER access — a scalar plane descriptor + a 32-bit lane offset
// (flat addressing kept 19 64-bit lane pointers alive: 108 registers), whose hardware range check returns 0 / drops the
// store for the out-of-image lanes (offset VSX_BUF_OOB), so there is no branch or select around any access.
__global__ __launch_bounds__(256, 6) void ssim_bwd_in_kernel(const float* __restrict__ P, const float* __restrict__ T,
                                                             const float* __restrict__ dmu, const float* __restrict__ dPn,
                                                             float* __restrict__ dP, int D, int H, int W, float l1c_,
                                                             float l2c_, const float* __restrict__ gout_p, int has_ssim,
                                                             const float* __restrict__ coef, int C, int last) {
  __shared__ float S[SI][SLD];
  __shared__ float R[SI][ST];
  const TileId tl = xcd_tile();
  const int bc = tl.z;
  const int iy0 = tl.y * ST, ix0 = tl.x * ST;
  const int Ho = H - 10, Wo = W - 10;
  const int Hn = H / 2, Wn = W / 2;
  constexpr int NOUT = (ST * ST + 255) / 256;
  static_assert(NOUT == 4 && 256 / ST * NOUT == ST, "column-segment mapping of the vertical pass");
  // thread (x = tid % 32, tid / 32) owns the NOUT consecutive rows of column x whose 11-tap column sums share their reads
  const int tx = threadIdx.x & (ST - 1), ty0 = NOUT * (threadIdx.x / ST);
  uint32_t off0[NOUT], offn[NOUT];  // byte offsets into a plane
  bool live[NOUT];
#pragma unroll
  for (int i = 0; i < NOUT; ++i) {
    const int gy = iy0 + ty0 + i, gx = ix0 + tx;
    live[i] = gy < H && gx < W;
    off0[i] = live[i] ? (uint32_t)(gy * W + gx) * 4u : VSX_BUF_OOB;
    offn[i] = live[i] && (gy >> 1) < Hn && (gx >> 1) < Wn ? (uint32_t)((gy >> 1) * Wn + (gx >> 1)) * 4u : VSX_BUF_OOB;
  }
  const size_t zs = (size_t)H * W, zsn = (size_t)Hn * Wn;
  const float* Pg = P + (size_t)bc * D * zs;
  const float* Tg = T + (size_t)bc * D * zs;
  float* dPg = dP + (size_t)bc * D * zs;
  const float* Ng = dPn ? dPn + (size_t)bc * D * zsn : P;
  const size_t nbytes = dPn ? zsn * 4 : 0;  // no next scale: every pooled load is out of range = 0
  float pv[NOUT], tv[NOUT], nv[NOUT];
  auto fetch = [&](int z) {
    const __amdgpu_buffer_rsrc_t rp = buf_rsrc(Pg + z * zs, zs * 4), rt = buf_rsrc(Tg + z * zs, zs * 4),
                                 rn = buf_rsrc(Ng + z * zsn, nbytes);
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
      pv[i] = buf_ld(rp, off0[i]);
      tv[i] = buf_ld(rt, off0[i]);
      nv[i] = buf_ld(rn, offn[i]);
    }
  };
  fetch(0);  // in flight across the field's halo loads and the LDS passes
  const float kb = round_bf16(1.0f / (float)(D * 121));
  const float gsc = gout_p ? gout_p[0] : 1.f;
  const float l1c = l1c_ * gsc, l2c = l2c_ * gsc;
  float G[NOUT][3];
#pragma unroll
  for (int i = 0; i < NOUT; ++i) G[i][0] = G[i][1] = G[i][2] = 0.f;
  if (has_ssim) {
    const size_t plane = (size_t)Ho * Wo;
    const size_t nbc = (size_t)gridDim.z;
    // dmu of an unscaled field (see ssim_tile_kernel): this sample's factor for the map the loss uses at this scale
    const float gsample = coef ? coef[2 * (bc / C) + (last ? 0 : 1)] : 1.f;
    constexpr int NIN = (SI * SI + 255) / 256;
    float hv[NIN];
    uint32_t ho[NIN];
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const int idx = threadIdx.x + i * 256;
      const int iy = idx / SI, ix = idx - iy * SI;
      const int oy = iy0 - 10 + iy, ox = ix0 - 10 + ix;
      ho[i] = idx < SI * SI && oy >= 0 && oy < Ho && ox >= 0 && ox < Wo ? (uint32_t)(oy * Wo + ox) * 4u : VSX_BUF_OOB;
    }
    auto halo = [&](int q) {  // one field's halo, all loads in flight together
      const __amdgpu_buffer_rsrc_t rd = buf_rsrc(dmu + ((size_t)q * nbc + bc) * plane, plane * 4);
#pragma unroll
      for (int i = 0; i < NIN; ++i) hv[i] = buf_ld(rd, ho[i]);
    };
    halo(0);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      if (q) __syncthreads();  // the previous field's column sums have been read
#pragma unroll
      for (int i = 0; i < NIN; ++i) {
        const int idx = threadIdx.x + i * 256;
        if (idx < SI * SI) S[idx / SI][idx % SI] = coef ? round_bf16(hv[i] * gsample) : hv[i];
      }
      if (q < 2) halo(q + 1);  // the next field travels during this one's two LDS passes
      __syncthreads();
      // 11-tap sums from registers (see ssim_tile_fused_kernel): 14 reads per 4 outputs instead of 44, same summation order
      for (int item = threadIdx.x; item < SI * (ST / 4); item += 256) {
        const int iy = item / (ST / 4), sg = item % (ST / 4);
        float v[14];
#pragma unroll
        for (int k = 0; k < 14; ++k) v[k] = S[iy][sg * 4 + k];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float a = 0.f;
#pragma unroll
          for (int k = 0; k < 11; ++k) a += v[j + k];
          R[iy][sg * 4 + j] = a;
        }
      }
      __syncthreads();
      float v[NOUT + 10];
#pragma unroll
      for (int k = 0; k < NOUT + 10; ++k) v[k] = R[ty0 + k][tx];
#pragma unroll
      for (int i = 0; i < NOUT; ++i) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) a += v[i + k];
        G[i][q] = round_bf16(kb * (live[i] ? a : 0.f));
      }
    }
  }
  for (int z = 0; z < D; ++z) {
    float pc[NOUT], tc[NOUT], nc[NOUT];
#pragma unroll
    for (int i = 0; i < NOUT; ++i) { pc[i] = pv[i]; tc[i] = tv[i]; nc[i] = nv[i]; }
    if (z + 1 < D) fetch(z + 1);
    const __amdgpu_buffer_rsrc_t rw = buf_rsrc(dPg + z * zs, zs * 4);
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
      const float p = pc[i], t = tc[i];
      float g = G[i][0] + 2.f * p * G[i][1] + t * G[i][2];
      g += 0.25f * nc[i];  // 0 where there is no pooled pixel (out-of-range load)
      const float d = p - t;
      if (l1c != 0.f) g += d > 0.f ? l1c : (d < 0.f ? -l1c : 0.f);
      if (l2c != 0.f) g += 2.f * l2c * d;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(g), rw, (int)off0[i], 0, 0);  // dropped outside the image
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
                                                            int sum_slots, float a1, float a2, float a3,
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
    float l = 0.f, s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < (sum_slots > 1 ? sum_slots : 1); ++i) {  // slotted partial sums of the one-pass forward (or plain scalars)
      if (a1 != 0.f) s1 += l1sum[i * LSTRIDE];
      if (a2 != 0.f) s2 += l2sum[i * LSTRIDE];
    }
    if (a1 != 0.f) l += a1 * s1 / nelem;
    if (a2 != 0.f) l += a2 * s2 / nelem;
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

/* max over a stack of n floats into the SLOTTED tmax (VSX_LOSS_SLOTS x VSX_LOSS_SLOT_STRIDE floats, pre-set to -inf): metrics.py:298 data_range of scale 0 for the one-pass
 * training forward below (the other scales' ranges come out of that pass). */
extern "C" int32_t vsx_loss_tmax(const float* T, int64_t n, float* tmax, vsx_stream_t stream) {
  VSX_CHECK(T && tmax && n > 0, "vsx_loss_tmax: bad arguments");
  long nblk = vsx_cdiv((long)(n >> 2) + 1, 256L * 8);
  if (nblk > 4096) nblk = 4096;
  hipLaunchKernelGGL(loss_tmax_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, T, (long)n, tmax);
  VSX_LAUNCH_CHECK();
  return 0;
}

/* vsx_ssim_scale_fwd_dmu + the vsx_loss_pool pass of the NEXT scale in one pass over this scale's stacks: also writes
 * Po / To [B, C, D, H/2, W/2] (avg_pool3d (1,2,2), metrics.py:340-341; NULL for the last scale), tmax_next[0] = max(To)
 * (pre-set to -inf; NULL with Po) and, when l1sum != NULL, adds the L1 / L2 sums of this scale (mixed_loss.py:58-63).
 * tmax = this scale's data range, already final (vsx_loss_tmax for scale 0, the previous scale's launch otherwise).
 * tmax, tmax_next, l1sum, l2sum are SLOTTED accumulators (VSX_LOSS_SLOTS partial values, VSX_LOSS_SLOT_STRIDE floats apart;
 * the range is their max, the sums their sum: vsx_loss_finalize(sum_slots = VSX_LOSS_SLOTS)). */
extern "C" int32_t vsx_ssim_scale_fwd_fused(const float* P, const float* T, const float* tmax, float* sum_ssim, float* sum_cs,
                                            float* dmu, float* Po, float* To, float* tmax_next, float* l1sum, float* l2sum,
                                            int32_t B, int32_t C, int32_t D, int32_t H, int32_t W, int32_t last,
                                            vsx_stream_t stream) {
  VSX_CHECK(P && T && tmax && sum_ssim && sum_cs && dmu, "vsx_ssim_scale_fwd_fused: null pointer");
  VSX_CHECK(H >= 11 && W >= 11 && D >= 1, "vsx_ssim_scale_fwd_fused: plane %dx%d smaller than the 11x11 window", H, W);
  VSX_CHECK((Po == nullptr) == (To == nullptr) && (Po == nullptr) == (tmax_next == nullptr) && (l1sum == nullptr) == (l2sum == nullptr),
            "vsx_ssim_scale_fwd_fused: pointer groups (Po, To, tmax_next) / (l1sum, l2sum)");
  VSX_CHECK((long)42 * W < (1l << 31), "vsx_ssim_scale_fwd_fused: row length");
  dim3 grid(vsx_cdiv(W - 10, ST), vsx_cdiv(H - 10, ST), B * C);
  const bool vec = (W & 1) == 0 && ((reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(T)) & 7) == 0;
#define VSX_FUSED_LAUNCH(VEC_)                                                                                           \
  hipLaunchKernelGGL((ssim_tile_fused_kernel<VEC_>), grid, dim3(FT), 0, (hipStream_t)stream, P, T, tmax, C, D, H, W,   \
                     sum_ssim, sum_cs, dmu, last ? 1 : 0, Po, To, tmax_next, l1sum, l2sum)
  if (vec) VSX_FUSED_LAUNCH(true); else VSX_FUSED_LAUNCH(false);
#undef VSX_FUSED_LAUNCH
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
                                     const float* npix, float nelem, int32_t B, int32_t nscale, int32_t sum_slots, float a1,
                                     float a2, float a3, const float* gout, float* loss, float* coef, float* ms_out,
                                     vsx_stream_t stream) {
  VSX_CHECK(loss && B > 0 && nscale >= 1 && nscale <= 5, "vsx_loss_finalize: bad arguments");
  VSX_CHECK(sum_slots == 0 || sum_slots == 1 || sum_slots == LSLOTS, "vsx_loss_finalize: sum_slots must be 0 / 1 (scalars) or VSX_LOSS_SLOTS");
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sum_ssim, sum_cs, l1sum, l2sum,
                     npix, nelem, B, nscale, sum_slots, a1, a2, a3, gout, loss, coef, ms_out);
  VSX_LAUNCH_CHECK();
  return 0;
}
