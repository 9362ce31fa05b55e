// PixelToVoxelHead 3x3x3 convolution (MONAI Convolution conv, viscy_models/components/heads.py:607-616)
// as DIRECT LDS-tiled MFMA kernels for the production shape (bf16, 8 -> 32 channels, 5 output planes).
//
// The z-batched implicit GEMMs of gemm.hip (VSX_A_CONV3) gather every 3x3 tap from global memory: PMC showed 19.4 GB
// of fabric reads per step for the data gradient alone (unique input 0.67 GB) and 3.7 + 2.1 + 1.3 ms per step for the
// three passes.  Here a workgroup stages one spatial tile (+ halo) in LDS once and every tap is an LDS read:
//   forward : U[p, z, n]      = b[n] + sum_{dy,dx,dz,c} hin[p + (dy-1, dx-1), z + dz, c] * W[n][(dy,dx,dz), c]   (+ IN statistics)
//   wgrad   : dW[n][(t), c]  += sum_{p,z} dU[p, z, n] * hin[p + d(t), z + dz(t), c]                 (+ db[n] via a ones column)
//   dgrad   : dhin[q, z', c]  = sum_{ey,ex,z,n} dU[q + (ey-1, ex-1), z, n] * W[n][(2-ey, 2-ex, z'-z), c]
// Layouts (channels-last, as everywhere): hin [B*H2*W2, 7*8], U / dU [B*H2*W2, 5*32], W [32][27*8] with
// k = ((dy*3 + dx)*3 + dz)*8 + c  (the VSX_A_CONV3 weight layout, so both paths share the prepared weights).
#include "vsx_common.h"
#include "../../include/vsx.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define HC_C3 8
#define HC_CMID 32
#define HC_ZO 5
#define HC_D7 7
#define HC_K 216  // 27 * 8

__device__ __forceinline__ f32x4 hc_mfma(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 hc_zero8() {
  union { uint4 u; bf16x8 v; } t;
  t.u = make_uint4(0u, 0u, 0u, 0u);
  return t.v;
}
// 8 contraction values for one column out of a row-major LDS tile: two transposing 8-byte reads (rows r and r+16 of a
// 32-row step); a0/a1 are THIS lane's addresses (see gemm.hip lds_frag_mn_bf16 for the slot mapping)
__device__ __forceinline__ bf16x8 hc_tr(const char* a0, const char* a1) {
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a1));
  union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
  u.s.lo = lo;
  u.s.hi = hi;
  return u.v;
}

// ------------------------------------------------------------------------------------------------ forward
// workgroup = 16x16 output pixels of one sample, all 5 planes; wave w owns pixel rows 4w..4w+3 (4 pixel fragments).
// MFMA roles: A = weights (rows = output channel), B = pixels -> a lane ends up with 4 consecutive channels of one pixel.
constexpr int HF_PS = HC_D7 * HC_C3 * 2;  // 112 B per halo pixel: 16 consecutive pixels hit 16 distinct 16-byte bank groups
// Round 5: a workgroup is 16 x 8 pixels (a wave owns 2 pixel rows) and the outputs leave through a wave-private staging row.  The
// first version (16 x 16 pixels, z outer) stored 8 bytes per lane straight from the accumulator layout: 32-byte runs at a
// 320-byte pixel stride, 2.7 TB/s for a pass that is 3/4 stores.  With the loops turned (pixel row outer, plane inner: the same
// fragment reads and MFMAs) all 5 planes x 32 channels of 16 consecutive pixels — 5 120 contiguous bytes — are parked in LDS and
// leave as five fully coalesced 1 KiB store instructions.  The smaller tile pays for the staging (20 + 21 KB: still three
// workgroups per CU by registers).
constexpr int HF_TY = 8;                           // pixel rows per workgroup
constexpr int HF_SR = HC_ZO * HC_CMID * 2 + 16;    // staging bytes per pixel (+16: the 16 lanes of a store group hit 16 bank pairs)
__global__ __launch_bounds__(256, 3) void head_conv_fwd_kernel(const bf16_t* __restrict__ hin, const bf16_t* __restrict__ Wc,
                                                            const float* __restrict__ bias, bf16_t* __restrict__ U,
                                                            float* __restrict__ ssum, float* __restrict__ ssq, int H2, int W2,
                                                            float* __restrict__ det_ws) {
  __shared__ __attribute__((aligned(16))) char tile[(HF_TY + 2) * 18 * HF_PS];
  __shared__ __attribute__((aligned(16))) char stage[4 * 16 * HF_SR];
  __shared__ float red[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p16 = lane & 15, kq = lane >> 4;
  const int b = blockIdx.z, ty0 = blockIdx.y * HF_TY, tx0 = blockIdx.x * 16;
  const size_t img = (size_t)b * H2 * W2;
  {
    // staging: every thread's loads first, then its LDS stores.  Written as `load; store` in one strided loop the compiler
    // kept a real loop with `s_waitcnt vmcnt(0)` in front of each store — one memory round trip per iteration, 9–15 per tile
    // (rocprofv3: the data-gradient kernel ran at 1.6 TB/s with its MFMAs idle 3/4 of the time)
    constexpr int NPIX = (HF_TY + 2) * 18;
    constexpr int NST = (NPIX * HC_D7 + 255) / 256;
    uint4 sv[NST];
#pragma unroll
    for (int it = 0; it < NST; ++it) {
      const int c = tid + it * 256;
      const int pix = c / HC_D7, zc = c - pix * HC_D7;
      const int py = pix / 18, px = pix - py * 18;
      const int y = ty0 + py - 1, x = tx0 + px - 1;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (c < NPIX * HC_D7 && y >= 0 && y < H2 && x >= 0 && x < W2)
        v = *reinterpret_cast<const uint4*>(hin + (img + (size_t)y * W2 + x) * (HC_D7 * HC_C3) + zc * HC_C3);
      sv[it] = v;
    }
#pragma unroll
    for (int it = 0; it < NST; ++it) {
      const int c = tid + it * 256;
      if (c < NPIX * HC_D7) {
        const int pix = c / HC_D7, zc = c - pix * HC_D7;
        const uint4 v = sv[it];
        *reinterpret_cast<uint4*>(tile + pix * HF_PS + zc * 16) = v;
      }
    }
  }
  // weights: 2 channel fragments x 7 K-steps (4 taps each; tap 27 does not exist -> zero), resident in registers
  bf16x8 wf[2][7];
  int toff[7];
#pragma unroll
  for (int kk = 0; kk < 7; ++kk) {
    const int t = kk * 4 + kq;
    const int tt = t < 27 ? t : 26;
    const int dyx = tt / 3, dz = tt - dyx * 3;
    const int dy = dyx / 3, dx = dyx - dy * 3;
    toff[kk] = (dy * 18 + dx) * HF_PS + dz * 16;
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
      wf[nf][kk] = t < 27 ? *reinterpret_cast<const bf16x8*>(Wc + (size_t)(nf * 16 + p16) * HC_K + t * 8) : hc_zero8();
  }
  float bs[2][4], s1[2][4], s2[2][4];
#pragma unroll
  for (int nf = 0; nf < 2; ++nf)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bs[nf][r] = bias ? bias[nf * 16 + kq * 4 + r] : 0.f;
      s1[nf][r] = 0.f;
      s2[nf][r] = 0.f;
    }
  __syncthreads();
  char* st = stage + wave * (16 * HF_SR);
#pragma unroll 1
  for (int mf = 0; mf < 2; ++mf) {
    const int pbase = ((wave * 2 + mf) * 18 + p16) * HF_PS;
#pragma unroll
    for (int z = 0; z < HC_ZO; ++z) {
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int kk = 0; kk < 7; ++kk) {
        const bf16x8 pf = *reinterpret_cast<const bf16x8*>(tile + pbase + toff[kk] + z * 16);
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) acc[nf] = hc_mfma(wf[nf][kk], pf, acc[nf]);
      }
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        float c[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = acc[nf][r] + bs[nf][r];
        uint2 o;
        o.x = f32x2_to_bf16x2_bits(c[0], c[1]);
        o.y = f32x2_to_bf16x2_bits(c[2], c[3]);
        *reinterpret_cast<uint2*>(st + p16 * HF_SR + (z * HC_CMID + nf * 16 + kq * 4) * 2) = o;
        // InstanceNorm statistics of the STORED (rounded) value, like VSX_EPI_BIAS_STATS
        const float q0 = __uint_as_float(o.x << 16), q1 = __uint_as_float(o.x & 0xffff0000u);
        const float q2 = __uint_as_float(o.y << 16), q3 = __uint_as_float(o.y & 0xffff0000u);
        s1[nf][0] += q0; s1[nf][1] += q1; s1[nf][2] += q2; s1[nf][3] += q3;
        s2[nf][0] += q0 * q0; s2[nf][1] += q1 * q1; s2[nf][2] += q2 * q2; s2[nf][3] += q3 * q3;
      }
    }
    // the row of 16 pixels x 320 bytes is contiguous in U: 320 16-byte chunks, 64 per store instruction
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    char* dst = reinterpret_cast<char*>(U + (img + (size_t)(ty0 + wave * 2 + mf) * W2 + tx0) * (HC_ZO * HC_CMID));
#pragma unroll
    for (int it = 0; it < 5; ++it) {
      const int c = lane + it * 64;
      const int px = c / 20, ch = c - px * 20;
      const uint4 v = *reinterpret_cast<const uint4*>(st + px * HF_SR + ch * 16);
      *reinterpret_cast<uint4*>(dst + c * 16) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // lanes of one kq group (16 pixels) -> one partial per channel; waves -> LDS; one atomic per (sample, channel) per workgroup
#pragma unroll
  for (int nf = 0; nf < 2; ++nf)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float a = group_sum<16>(s1[nf][r]), q = group_sum<16>(s2[nf][r]);
      if (p16 == 0) {
        red[wave][nf * 16 + kq * 4 + r] = a;
        red[wave][32 + nf * 16 + kq * 4 + r] = q;
      }
    }
  __syncthreads();
  if (tid < 64) {
    const float v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    if (det_ws) {  // det_reduce: one row of 64 partials per workgroup, added up in tile order by vsx_det_group_sum
      det_ws[((size_t)b * gridDim.y * gridDim.x + (size_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 + tid] = v;
    } else {
      atomicAdd((tid < 32 ? ssum : ssq) + (size_t)b * HC_CMID + (tid & 31), v);
    }
  }
}

// Round 6: the same pass as a persistent kernel.  One tile per workgroup put the tile's phases in a row — 5 halo loads per thread,
// wait, LDS stores, barrier, 140 MFMAs per wave, stores — with three workgroups per CU to overlap them: 3.5 TB/s of its bytes at
// 0.27 ms of MFMA and as much LDS time per launch.  Here a workgroup walks a contiguous range of tiles (x fastest, so a tile shares
// its halo columns with the one before it and its halo rows with the tile 8 x W2 / 16 steps back), keeps the weight fragments in
// registers for the whole launch and requests the NEXT tile's halo (5 x 16 bytes per thread) before it starts on the current one;
// the per-tile statistics leave as before (one atomic per (sample, channel) and tile, or the tile's det_reduce row).
// Two workgroups per CU: the unrolled plane loop with the next tile's five chunks in flight wants 230 registers (at the three
// workgroups of the one-tile kernel: 70 spilled, 1.67 ms; with the plane loop rolled: 10 spilled, 1.12 ms); measured at B = 512:
// 1 022 -> 945 us, 1 034 -> 968 us on a second box.
constexpr int HCF_WPE = 2;
__global__ __launch_bounds__(256, HCF_WPE) void head_conv_fwd_persist_kernel(const bf16_t* __restrict__ hin, const bf16_t* __restrict__ Wc,
                                                                    const float* __restrict__ bias, bf16_t* __restrict__ U,
                                                                    float* __restrict__ ssum, float* __restrict__ ssq, int H2,
                                                                    int W2, float* __restrict__ det_ws, int ntiles,
                                                                    int tiles_per_wg) {
  __shared__ __attribute__((aligned(16))) char tile[(HF_TY + 2) * 18 * HF_PS];
  __shared__ __attribute__((aligned(16))) char stage[4 * 16 * HF_SR];
  __shared__ float red[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p16 = lane & 15, kq = lane >> 4;
  const int tiles_x = W2 / 16, tiles_y = H2 / HF_TY, tiles_img = tiles_x * tiles_y;
  constexpr int NPIX = (HF_TY + 2) * 18;
  constexpr int NST = (NPIX * HC_D7 + 255) / 256;
  // this thread's chunks of a halo tile never change: (pixel, plane) slots and their LDS addresses
  int slot[NST];  // py | px << 8 | (plane + 1) << 16 (0: no chunk)
#pragma unroll
  for (int it = 0; it < NST; ++it) {
    const int c = tid + it * 256;
    const int pix = c / HC_D7;
    const int py = pix / 18;
    slot[it] = c < NPIX * HC_D7 ? (py | (pix - py * 18) << 8 | (c - pix * HC_D7 + 1) << 16) : 0;
  }
  auto issue = [&](int t, uint4 (&sv)[NST]) {
    const int b = t / tiles_img, r = t - b * tiles_img;
    const int ty0 = (r / tiles_x) * HF_TY, tx0 = (r % tiles_x) * 16;
    const size_t img = (size_t)b * H2 * W2;
#pragma unroll
    for (int it = 0; it < NST; ++it) {
      const int y = ty0 + (slot[it] & 0xff) - 1, x = tx0 + ((slot[it] >> 8) & 0xff) - 1, zc = (slot[it] >> 16) - 1;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (zc >= 0 && y >= 0 && y < H2 && x >= 0 && x < W2)
        v = *reinterpret_cast<const uint4*>(hin + (img + (size_t)y * W2 + x) * (HC_D7 * HC_C3) + zc * HC_C3);
      sv[it] = v;
    }
  };
  // weights: 2 channel fragments x 7 K-steps (4 taps each; tap 27 does not exist -> zero), resident in registers
  bf16x8 wf[2][7];
  int toff[7];
#pragma unroll
  for (int kk = 0; kk < 7; ++kk) {
    const int t = kk * 4 + kq;
    const int tt = t < 27 ? t : 26;
    const int dyx = tt / 3, dz = tt - dyx * 3;
    const int dy = dyx / 3, dx = dyx - dy * 3;
    toff[kk] = (dy * 18 + dx) * HF_PS + dz * 16;
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
      wf[nf][kk] = t < 27 ? *reinterpret_cast<const bf16x8*>(Wc + (size_t)(nf * 16 + p16) * HC_K + t * 8) : hc_zero8();
  }
  float bs[2][4];
#pragma unroll
  for (int nf = 0; nf < 2; ++nf)
#pragma unroll
    for (int r = 0; r < 4; ++r) bs[nf][r] = bias ? bias[nf * 16 + kq * 4 + r] : 0.f;
  const int t_begin = blockIdx.x * tiles_per_wg, t_end = min(ntiles, t_begin + tiles_per_wg);
  if (t_begin >= t_end) return;
  uint4 sv[NST];
  issue(t_begin, sv);
  char* st = stage + wave * (16 * HF_SR);
  for (int t = t_begin; t < t_end; ++t) {
    const int b = t / tiles_img, rr = t - b * tiles_img;
    const int tyi = rr / tiles_x, txi = rr - tyi * tiles_x;
    const int ty0 = tyi * HF_TY, tx0 = txi * 16;
    const size_t img = (size_t)b * H2 * W2;
    // (every wave is done with the previous tile: it passed the statistics barrier at the end of the last step)
#pragma unroll
    for (int it = 0; it < NST; ++it)
      if (slot[it]) *reinterpret_cast<uint4*>(tile + ((slot[it] & 0xff) * 18 + ((slot[it] >> 8) & 0xff)) * HF_PS + ((slot[it] >> 16) - 1) * 16) = sv[it];
    __syncthreads();
    if (t + 1 < t_end) issue(t + 1, sv);
    float s1[2][4], s2[2][4];
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s1[nf][r] = 0.f;
        s2[nf][r] = 0.f;
      }
#pragma unroll 1
    for (int mf = 0; mf < 2; ++mf) {
      const int pbase = ((wave * 2 + mf) * 18 + p16) * HF_PS;
#pragma unroll
      for (int z = 0; z < HC_ZO; ++z) {
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kk = 0; kk < 7; ++kk) {
          const bf16x8 pf = *reinterpret_cast<const bf16x8*>(tile + pbase + toff[kk] + z * 16);
#pragma unroll
          for (int nf = 0; nf < 2; ++nf) acc[nf] = hc_mfma(wf[nf][kk], pf, acc[nf]);
        }
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
          float c[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) c[r] = acc[nf][r] + bs[nf][r];
          uint2 o;
          o.x = f32x2_to_bf16x2_bits(c[0], c[1]);
          o.y = f32x2_to_bf16x2_bits(c[2], c[3]);
          *reinterpret_cast<uint2*>(st + p16 * HF_SR + (z * HC_CMID + nf * 16 + kq * 4) * 2) = o;
          // InstanceNorm statistics of the STORED (rounded) value, like VSX_EPI_BIAS_STATS
          const float q0 = __uint_as_float(o.x << 16), q1 = __uint_as_float(o.x & 0xffff0000u);
          const float q2 = __uint_as_float(o.y << 16), q3 = __uint_as_float(o.y & 0xffff0000u);
          s1[nf][0] += q0; s1[nf][1] += q1; s1[nf][2] += q2; s1[nf][3] += q3;
          s2[nf][0] += q0 * q0; s2[nf][1] += q1 * q1; s2[nf][2] += q2 * q2; s2[nf][3] += q3 * q3;
        }
      }
      // the row of 16 pixels x 320 bytes is contiguous in U: 320 16-byte chunks, 64 per store instruction
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      char* dst = reinterpret_cast<char*>(U + (img + (size_t)(ty0 + wave * 2 + mf) * W2 + tx0) * (HC_ZO * HC_CMID));
#pragma unroll
      for (int it = 0; it < 5; ++it) {
        const int c = lane + it * 64;
        const int px = c / 20, ch = c - px * 20;
        const uint4 v = *reinterpret_cast<const uint4*>(st + px * HF_SR + ch * 16);
        *reinterpret_cast<uint4*>(dst + c * 16) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    // lanes of one kq group (16 pixels) -> one partial per channel; waves -> LDS; one atomic per (sample, channel) per tile
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = group_sum<16>(s1[nf][r]), q = group_sum<16>(s2[nf][r]);
        if (p16 == 0) {
          red[wave][nf * 16 + kq * 4 + r] = a;
          red[wave][32 + nf * 16 + kq * 4 + r] = q;
        }
      }
    __syncthreads();  // (also: every wave is done reading the halo tile)
    if (tid < 64) {
      const float v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
      if (det_ws) {  // det_reduce: one row of 64 partials per tile, added up in tile order by vsx_det_group_sum
        det_ws[((size_t)b * tiles_img + (size_t)tyi * tiles_x + txi) * 64 + tid] = v;
      } else {
        atomicAdd((tid < 32 ? ssum : ssq) + (size_t)b * HC_CMID + (tid & 31), v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// persistent workgroups walk 8x16-pixel tiles; contraction index of the MFMA = 32 pixels (2 tile rows), both operands
// come out of row-major LDS tiles through the transposing read.  Output 32 x 217 (216 weight columns + a ones column
// = bias gradient) = 2 x 14 fragments, column fragments dealt to the 4 waves.
constexpr int HW_DS = HC_ZO * HC_CMID * 2;  // 320 B per dU pixel: 4 rows x 4 column groups of a transposing read = 16 distinct bank pairs
__global__ __launch_bounds__(256) void head_conv_wgrad_kernel(const bf16_t* __restrict__ hin, const bf16_t* __restrict__ dU,
                                                              float* __restrict__ dW, float* __restrict__ db, int B, int H2,
                                                              int W2) {
  __shared__ __attribute__((aligned(16))) char du_t[128 * HW_DS];      // 40 KB
  __shared__ __attribute__((aligned(16))) char hin_t[10 * 18 * HF_PS]; // 20 KB
  __shared__ __attribute__((aligned(16))) unsigned short consts[8];    // [1, 0, 0, 0 | 0, 0, 0, 0] (bf16)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane & 15, kq = lane >> 4;
  if (tid < 8) consts[tid] = tid == 0 ? 0x3f80 : 0;
  const int tiles_x = W2 / 16, tiles_y = H2 / 8;
  const int ntiles = B * tiles_y * tiles_x;
  // this wave's column fragments jf = wave, wave + 4, wave + 8, wave + 12 (< 14)
  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // per-lane B-operand address pieces: column group cg = q & 3 -> 4 columns c = cg*4.. of the 16-column fragment:
  // c < 8 -> tap 2*jf, channels c..c+3; c >= 8 -> tap 2*jf + 1, channels c-8..
  const int slot = kq * 4 + (q >> 2);   // contraction slot (pixel within the first 16 of a 32-pixel step); +16 = next tile row
  const int cg = q & 3;
  int boff[4];  // halo-tile offset of (tap displacement, dz plane, channel) ; -1 -> ones column, -2 -> zero
#pragma unroll
  for (int jl = 0; jl < 4; ++jl) {
    const int jf = wave + jl * 4;
    const int tap = 2 * jf + (cg >> 1);
    if (jf >= 14 || tap > 27) boff[jl] = -2;
    else if (tap == 27) boff[jl] = (cg & 1) ? -2 : -1;
    else {
      const int dyx = tap / 3, dz = tap - dyx * 3;
      const int dy = dyx / 3, dx = dyx - dy * 3;
      boff[jl] = (dy * 18 + dx) * HF_PS + dz * 16 + (cg & 1) * 8;
    }
  }
  const char* cptr1 = reinterpret_cast<const char*>(consts);
  const char* cptr0 = cptr1 + 8;
#pragma unroll 1
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int b = t / (tiles_y * tiles_x), rem = t - b * tiles_y * tiles_x;
    const int ty0 = (rem / tiles_x) * 8, tx0 = (rem % tiles_x) * 16;
    const size_t img = (size_t)b * H2 * W2;
    __syncthreads();  // previous tile fully consumed
    {  // all loads of the tile in flight together, then the LDS stores (see head_conv_fwd_kernel)
      constexpr int NDU = 128 * 20 / 256, NHI = (180 * HC_D7 + 255) / 256;
      uint4 dv[NDU], hv[NHI];
#pragma unroll
      for (int it = 0; it < NDU; ++it) {  // dU tile: 128 pixels x 20 chunks
        const int c = tid + it * 256;
        const int pix = c / 20, ch = c - pix * 20;
        const int y = ty0 + (pix >> 4), x = tx0 + (pix & 15);
        dv[it] = *reinterpret_cast<const uint4*>(dU + (img + (size_t)y * W2 + x) * (HC_ZO * HC_CMID) + ch * 8);
      }
#pragma unroll
      for (int it = 0; it < NHI; ++it) {
        const int c = tid + it * 256;
        const int pix = c / HC_D7, zc = c - pix * HC_D7;
        const int py = pix / 18, px = pix - py * 18;
        const int y = ty0 + py - 1, x = tx0 + px - 1;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (c < 180 * HC_D7 && y >= 0 && y < H2 && x >= 0 && x < W2)
          v = *reinterpret_cast<const uint4*>(hin + (img + (size_t)y * W2 + x) * (HC_D7 * HC_C3) + zc * HC_C3);
        hv[it] = v;
      }
#pragma unroll
      for (int it = 0; it < NDU; ++it) {
        const int c = tid + it * 256;
        const int pix = c / 20, ch = c - pix * 20;
        const uint4 v = dv[it];
        *reinterpret_cast<uint4*>(du_t + pix * HW_DS + ch * 16) = v;
      }
#pragma unroll
      for (int it = 0; it < NHI; ++it) {
        const int c = tid + it * 256;
        if (c < 180 * HC_D7) {
          const int pix = c / HC_D7, zc = c - pix * HC_D7;
          const uint4 v = hv[it];
          *reinterpret_cast<uint4*>(hin_t + pix * HF_PS + zc * 16) = v;
        }
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int chunk = 0; chunk < 4; ++chunk) {  // 32 pixels = tile rows 2*chunk, 2*chunk + 1
      const char* arow = du_t + (chunk * 32 + slot) * HW_DS + cg * 8;
      const char* brow = hin_t + ((chunk * 2) * 18 + slot) * HF_PS;
#pragma unroll
      for (int z = 0; z < HC_ZO; ++z) {
        bf16x8 af[2], bfr[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = hc_tr(arow + (z * 32 + i * 16) * 2, arow + 16 * HW_DS + (z * 32 + i * 16) * 2);
#pragma unroll
        for (int jl = 0; jl < 4; ++jl) {
          // ONE transposing read for the whole wave (it exchanges data between lanes: never under divergent control flow);
          // lanes of the ones / zero columns just point at the constants
          const char* b0 = boff[jl] >= 0 ? brow + boff[jl] + z * 16 : (boff[jl] == -1 ? cptr1 : cptr0);
          const char* b1 = boff[jl] >= 0 ? b0 + 18 * HF_PS : b0;
          bfr[jl] = hc_tr(b0, b1);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jl = 0; jl < 4; ++jl) acc[i][jl] = hc_mfma(af[i], bfr[jl], acc[i][jl]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jl = 0; jl < 4; ++jl) {
      const int jf = wave + jl * 4;
      if (jf >= 14) continue;
      const int col = jf * 16 + q;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = i * 16 + kq * 4 + r;
        if (col < HC_K) atomicAdd(dW + (size_t)n * HC_K + col, acc[i][jl][r]);
        else if (col == HC_K && db != nullptr) atomicAdd(db + n, acc[i][jl][r]);
      }
    }
}

// ------------------------------------------------------------------------------------------------ data gradient
// out channels i = (z', c) (56, padded to 4 fragments of 16 = 2 planes each); for an input plane z only the planes
// z' in {z, z+1, z+2} get a contribution = exactly 2 of the 4 fragments (fa = z/2, fb = fa + 1 for even z;
// (z-1)/2, (z+1)/2 for odd z).  Weight fragments are pre-packed per (tap, z) step by head_conv_dgrad_prep_kernel.
constexpr int HD_PS = HC_ZO * HC_CMID * 2 + 16;  // 336 B per dU halo pixel (pad: 16 pixels -> 16 distinct bank groups)
__global__ void head_conv_dgrad_prep_kernel(const bf16_t* __restrict__ Wc, bf16_t* __restrict__ Wp) {
  // Wp[step = (ey*3 + ex)*5 + z][s = 0/1][lane 64][8]
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= 45 * 2 * 64) return;
  const int lane = g & 63, s = (g >> 6) & 1, step = g >> 7;
  const int p16 = lane & 15, kq = lane >> 4;
  const int z = step % 5, tap = step / 5;
  const int ey = tap / 3, ex = tap - ey * 3;
  const int fa = (z & 1) ? (z - 1) / 2 : z / 2;
  const int frag = fa + s;
  const int i = frag * 16 + p16, zp = i >> 3, c = i & 7;
  const int dz = zp - z;
  bf16_t* dst = Wp + (size_t)g * 8;
  const bool ok = zp < HC_D7 && dz >= 0 && dz <= 2;
  const int t = ((2 - ey) * 3 + (2 - ex)) * 3 + dz;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int n = kq * 8 + e;
    unsigned short v = 0;
    if (ok) v = *reinterpret_cast<const unsigned short*>(Wc + (size_t)n * HC_K + t * 8 + c);
    *reinterpret_cast<unsigned short*>(dst + e) = v;
  }
}

// Round 5: the packed weights travel L2 -> LDS ONCE per workgroup, one tap (5 steps x 2 fragments = 10 KiB) ahead of the MFMAs, by
// LDS-DMA into two alternating buffers; every wave reads its fragments from there (lane-linear, conflict-free ds_read_b128).  The
// first version had every wave fetch the fragments of every step from global memory — 360 KB of L1 traffic per workgroup for a
// 58 KB halo tile: the pass ran at 2.5 TB/s of its bytes with the texture path busy on weights.
template <int Z>
__device__ __forceinline__ void hd_step(const char* halo, int pb, const char* wl, f32x4 (&acc)[2][4]) {
  constexpr int FA = (Z & 1) ? (Z - 1) / 2 : Z / 2;
  const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(wl + (Z * 2 + 0) * 1024);
  const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(wl + (Z * 2 + 1) * 1024);
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) {
    const bf16x8 pf = *reinterpret_cast<const bf16x8*>(halo + pb + mf * 18 * HD_PS);
    acc[mf][FA] = hc_mfma(a0, pf, acc[mf][FA]);
    acc[mf][FA + 1] = hc_mfma(a1, pf, acc[mf][FA + 1]);
  }
}

__global__ __launch_bounds__(256) void head_conv_dgrad_kernel(const bf16_t* __restrict__ dU, const bf16_t* __restrict__ Wp,
                                                              bf16_t* __restrict__ dhin, int H2, int W2) {
  // separate LDS objects: the compiler's alias scopes then let the halo / fragment reads proceed while the DMA into the OTHER
  // weight buffer is in flight (csrc/mlp.hip)
  __shared__ __attribute__((aligned(16))) char halo[10 * 18 * HD_PS];  // 60 KB
  __shared__ __attribute__((aligned(1024))) char wb0[10 * 1024];
  __shared__ __attribute__((aligned(1024))) char wb1[10 * 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p16 = lane & 15, kq = lane >> 4;
  const int b = blockIdx.z, ty0 = blockIdx.y * 8, tx0 = blockIdx.x * 16;
  const size_t img = (size_t)b * H2 * W2;
  // the 10 KiB of tap `tap` -> buffer: pieces wave, wave + 4, wave + 8 (1 KiB = one wave instruction)
  const char* wsrc = reinterpret_cast<const char*>(Wp) + lane * 16;
  auto wdma = [&](int tap, char* dst) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int pc = wave + 4 * i;
      // (inline assembly on purpose: behind the builtin hipcc drains vmcnt(0) in front of every LDS read whose object it cannot
      // tell from the DMA's — here in every second tap — which serialises the prefetch; this kernel counts its own waits)
      if (pc < 10) {
        const char* src = wsrc + (size_t)(tap * 10 + pc) * 1024;
        const uint32_t d = (uint32_t)(size_t)(const __attribute__((address_space(3))) char*)(dst + pc * 1024);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(d) : "memory", "m0");
      }
    }
  };
  wdma(0, wb0);
  {  // all loads of the halo tile in flight together, then the LDS stores (see head_conv_fwd_kernel)
    constexpr int NST = (180 * 20 + 255) / 256;
    uint4 sv[NST];
#pragma unroll
    for (int it = 0; it < NST; ++it) {
      const int c = tid + it * 256;
      const int pix = c / 20, ch = c - pix * 20;
      const int py = pix / 18, px = pix - py * 18;
      const int y = ty0 + py - 1, x = tx0 + px - 1;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (c < 180 * 20 && y >= 0 && y < H2 && x >= 0 && x < W2)
        v = *reinterpret_cast<const uint4*>(dU + (img + (size_t)y * W2 + x) * (HC_ZO * HC_CMID) + ch * 8);
      sv[it] = v;
    }
#pragma unroll
    for (int it = 0; it < NST; ++it) {
      const int c = tid + it * 256;
      if (c < 180 * 20) {
        const int pix = c / 20, ch = c - pix * 20;
        const uint4 v = sv[it];
        *reinterpret_cast<uint4*>(halo + pix * HD_PS + ch * 16) = v;
      }
    }
  }
  f32x4 acc[2][4];
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[mf][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // wave w: output pixel rows 2w, 2w + 1; lane: pixel x = p16, channel quarter kq (8 of the 32 n per plane)
  const int pb0 = ((wave * 2) * 18 + p16) * HD_PS + kq * 16;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    // tap's weights (this wave's pieces: everything it has outstanding) and, at tap 0, the halo tile have landed for every wave;
    // every wave is done with the buffer that is refilled next
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tap + 1 < 9) wdma(tap + 1, (tap & 1) ? wb0 : wb1);
    const char* wl = ((tap & 1) ? wb1 : wb0) + lane * 16;
    const int ey = tap / 3, ex = tap - ey * 3;
    const int pb = pb0 + (ey * 18 + ex) * HD_PS;
    hd_step<0>(halo, pb + 0 * 64, wl, acc);
    hd_step<1>(halo, pb + 1 * 64, wl, acc);
    hd_step<2>(halo, pb + 2 * 64, wl, acc);
    hd_step<3>(halo, pb + 3 * 64, wl, acc);
    hd_step<4>(halo, pb + 4 * 64, wl, acc);
  }
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) {
    const size_t pix = img + (size_t)(ty0 + wave * 2 + mf) * W2 + tx0 + p16;
    bf16_t* dst = dhin + pix * (HC_D7 * HC_C3) + kq * 4;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      if (f * 16 + kq * 4 < HC_D7 * HC_C3) {
        uint2 o;
        o.x = f32x2_to_bf16x2_bits(acc[mf][f][0], acc[mf][f][1]);
        o.y = f32x2_to_bf16x2_bits(acc[mf][f][2], acc[mf][f][3]);
        *reinterpret_cast<uint2*>(dst + f * 16) = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ C-ABI
static int hc_check(const char* who, int B, int H2, int W2, int c3, int cmid, int zo, int dtype) {
  VSX_CHECK(dtype == VSX_BF16, "%s: the direct head convolution is built for bf16 only (use vsx_gemm_nt / VSX_A_CONV3 for fp32)", who);
  VSX_CHECK(c3 == HC_C3 && cmid == HC_CMID && zo == HC_ZO, "%s: built for %d -> %d channels, %d output planes (got %d -> %d, %d)", who,
            HC_C3, HC_CMID, HC_ZO, c3, cmid, zo);
  VSX_CHECK(B > 0 && H2 > 0 && W2 > 0 && H2 % 16 == 0 && W2 % 16 == 0, "%s: H2=%d, W2=%d must be positive multiples of 16", who, H2, W2);
  return 0;
}
extern "C" int32_t vsx_head_conv_supported(int32_t H2, int32_t W2, int32_t c3, int32_t cmid, int32_t zo, int32_t dtype) {
  return dtype == VSX_BF16 && c3 == HC_C3 && cmid == HC_CMID && zo == HC_ZO && H2 > 0 && W2 > 0 && H2 % 16 == 0 && W2 % 16 == 0;
}
extern "C" int32_t vsx_head_conv_fwd(const void* hin, const void* Wc, const float* bias, void* U, float* ssum, float* ssq,
                                     int32_t B, int32_t H2, int32_t W2, int32_t c3, int32_t cmid, int32_t zo, int32_t dtype,
                                     vsx_stream_t stream) {
  if (int e = hc_check("vsx_head_conv_fwd", B, H2, W2, c3, cmid, zo, dtype)) return e;
  VSX_CHECK(hin && Wc && U && ssum && ssq, "vsx_head_conv_fwd: null pointer");
  float* det_ws = nullptr;
  const int tiles = (W2 / 16) * (H2 / HF_TY);
  if (g_vsx_det_reduce) {  // fixed-order InstanceNorm sums: 64 partials per workgroup, then one ordered pass per array
    const long need = (long)B * tiles * 64;
    VSX_CHECK(g_vsx_det_ws != nullptr && g_vsx_det_ws_floats >= need, "vsx_head_conv_fwd: det_reduce needs vsx_det_workspace(>= %ld floats)", need);
    det_ws = g_vsx_det_ws;
  }
  const long ntiles = (long)B * tiles;
  if ((g_vsx_head_rows & 32) && ntiles <= 0x7fffffffL) {  // persistent (round 6): three workgroups per CU, a contiguous tile range each
    const int tpw = vsx_cdiv(ntiles, 256L * HCF_WPE);
    hipLaunchKernelGGL(head_conv_fwd_persist_kernel, dim3(vsx_cdiv(ntiles, (long)tpw)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)hin, (const bf16_t*)Wc, bias, (bf16_t*)U, ssum, ssq, H2, W2, det_ws, (int)ntiles, tpw);
  } else
  hipLaunchKernelGGL(head_conv_fwd_kernel, dim3(W2 / 16, H2 / HF_TY, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)hin,
                     (const bf16_t*)Wc, bias, (bf16_t*)U, ssum, ssq, H2, W2, det_ws);
  VSX_LAUNCH_CHECK();
  if (det_ws) {
    if (int e = vsx_det_group_sum(det_ws, 64, 0, ssum, B, tiles, HC_CMID, (hipStream_t)stream)) return e;
    return vsx_det_group_sum(det_ws, 64, 32, ssq, B, tiles, HC_CMID, (hipStream_t)stream);
  }
  return 0;
}
extern "C" int32_t vsx_head_conv_wgrad(const void* hin, const void* dU, float* dW, float* db, int32_t B, int32_t H2, int32_t W2,
                                       int32_t c3, int32_t cmid, int32_t zo, int32_t dtype, vsx_stream_t stream) {
  if (int e = hc_check("vsx_head_conv_wgrad", B, H2, W2, c3, cmid, zo, dtype)) return e;
  VSX_CHECK(hin && dU && dW, "vsx_head_conv_wgrad: null pointer");
  int tiles = B * (H2 / 8) * (W2 / 16);
  int grid = tiles < 512 ? tiles : 512;  // 2 workgroups per CU (60 KB LDS each); every workgroup ends with 32 x 217 atomics
  hipLaunchKernelGGL(head_conv_wgrad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)hin, (const bf16_t*)dU,
                     dW, db, B, H2, W2);
  VSX_LAUNCH_CHECK();
  return 0;
}
extern "C" int32_t vsx_head_conv_dgrad_prep(const void* Wc, void* Wp, int32_t dtype, vsx_stream_t stream) {
  VSX_CHECK(dtype == VSX_BF16 && Wc && Wp, "vsx_head_conv_dgrad_prep: bf16 pointers required");
  hipLaunchKernelGGL(head_conv_dgrad_prep_kernel, dim3(vsx_cdiv(45 * 2 * 64, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)Wc, (bf16_t*)Wp);
  VSX_LAUNCH_CHECK();
  return 0;
}
extern "C" int32_t vsx_head_conv_dgrad(const void* dU, const void* Wp, void* dhin, int32_t B, int32_t H2, int32_t W2, int32_t c3,
                                       int32_t cmid, int32_t zo, int32_t dtype, vsx_stream_t stream) {
  if (int e = hc_check("vsx_head_conv_dgrad", B, H2, W2, c3, cmid, zo, dtype)) return e;
  VSX_CHECK(dU && Wp && dhin, "vsx_head_conv_dgrad: null pointer");
  hipLaunchKernelGGL(head_conv_dgrad_kernel, dim3(W2 / 16, H2 / 8, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dU,
                     (const bf16_t*)Wp, (bf16_t*)dhin, H2, W2);
  VSX_LAUNCH_CHECK();
  return 0;
}
