// Depthwise 7x7 convolution on the matrix cores (bf16 production path) — forward and data gradient.
//
// The VALU stencil in dwconv.hip is compute-bound: 49 FMAs + bf16 unpacks per output keep it at 1.3–1.6 TB/s where a
// streaming kernel reaches 3.5.  A depthwise convolution has no channel contraction, but for ONE channel a tap row is a
// banded (Toeplitz) matrix product along x:
//
//     out_c[x, y] = sum_ky  sum_k  T_c,ky[x, k] * in_c[k, y + ky],      T_c,ky[x, k] = w_c[ky][k - x]  (0 <= k - x < 7)
//
// i.e. 7 MFMAs v_mfma_f32_16x16x32_bf16 per (channel, 16 x 16 output tile): A = T (16 output columns x 32 input columns,
// of which 22 matter), B = the input columns of 16 image rows shifted by ky, D accumulates in fp32.  22 % of the MFMA's
// MACs are useful — still > 5x the VALU rate, which makes the kernel memory-bound.  What it costs is layout: the tensor is
// channels-last in HBM and the MFMA wants x contiguous per channel, so tiles are transposed through LDS
// (16-byte global accesses, 2-byte LDS accesses on both sides) into per-channel planes.
//
// Structure (one persistent workgroup of 16 waves per CU):
//   * a workgroup owns one 32-channel slab and a contiguous range of 16 x (16*NXT)-pixel tiles; wave w owns channels
//     2w, 2w+1 of the slab and keeps their 2 x 7 Toeplitz fragments in registers for the whole launch (built once from the
//     fp32 tap-major weights, rounded to bf16 — the operand precision of the reference's autocast convolution);
//   * tiles are double-buffered: the global loads of tile i+1 are issued before the MFMAs of tile i and land in the other
//     LDS buffer afterwards; a channel's outputs overwrite its own input plane (only its wave reads it), then all threads
//     gather 16-byte channel vectors back out of the planes for the global store (+ the optional residual `add`).
//
// LDS plane of channel ch: [22 rows][PITCH] bf16 at ch * PLANE + (ch / 8) * 16 elements.  PITCH = 48 (NXT = 2) / 40
// (NXT = 1) keeps the ds_read_b128 of a B fragment 16-byte aligned and conflict-free; the (ch / 8) * 16 skew spreads the
// four 8-channel vectors of a pixel over distinct banks for the 2-byte transposing accesses.  Columns past the staged
// 16*NXT + 6 are read by the K = 32 window of the last x tile: they are zeroed once per launch (A is zero there, but
// 0 * garbage must not be NaN).
#include "vsx_common.h"
#include "../../include/vsx.h"

typedef __attribute__((ext_vector_type(4))) float dwm_f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 dwm_bf16x8;

constexpr int DWM_THREADS = 1024;
constexpr int DWM_CB = 32;      // channels per slab
constexpr int DWM_ROWS = 22;    // 16 output rows + 6 halo rows

template <int NXT>
struct DwmGeom {
  static constexpr int TW = 16 * NXT;
  static constexpr int IW = TW + 6;                       // staged columns
  static constexpr int PITCH = NXT == 1 ? 40 : 16 * NXT + 16;  // elements per plane row (>= 16*(NXT-1) + 32, rows 16-B aligned)
  static constexpr int PLANE = DWM_ROWS * PITCH;          // elements per channel plane (multiple of 8)
  static constexpr int BUF = DWM_CB * PLANE + (DWM_CB / 8) * 16;  // elements per buffer
};

static __device__ __forceinline__ int dwm_plane_base(int ch, int plane) { return ch * plane + (ch >> 3) * 16; }

template <int NXT, bool FLIP>
__global__ __launch_bounds__(DWM_THREADS) void dwconv7_mfma_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                                   const float* __restrict__ bias,
                                                                   const bf16_t* __restrict__ add, bf16_t* __restrict__ y,
                                                                   int B, int H, int W, int C, int nslab, int tiles_total,
                                                                   int tiles_per_wg) {
  typedef DwmGeom<NXT> G;
  constexpr int THREADS = DWM_THREADS, CPW = DWM_CB / (DWM_THREADS / 64);  // channels per wave (2)
  __shared__ __attribute__((aligned(16))) unsigned short lds[2][G::BUF];

  // XCD-aware order (workgroups are dealt round-robin to the 8 XCDs): one XCD gets a contiguous range of logical ids, so
  // the slabs of one tile range — which split every 128-byte line of a pixel between them — meet in one L2
  int bid = blockIdx.x;
  if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
  const int slab = bid % nslab;
  const int chunk = bid / nslab;
  const int t_begin = chunk * tiles_per_wg;
  const int t_end = min(tiles_total, t_begin + tiles_per_wg);
  if (t_begin >= t_end) return;
  const int c_base = slab * DWM_CB;
  const int tiles_x = (W + G::TW - 1) / G::TW, tiles_y = (H + 15) / 16;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int p16 = lane & 15, kq = lane >> 4;

  // ---- Toeplitz fragments of this wave's two channels: A[i = p16][k = kq*8 + e] = w[ky][k - i]
  // The slab's 49 x 32 taps go through LDS first (one coalesced load per thread): built straight from global memory the 112
  // predicated scalar loads of a lane compiled to 112 branch + load + s_waitcnt vmcnt(0) round trips in a row (950 lines of
  // ISA) with the results parked in scratch.
  __shared__ float wsm[49 * DWM_CB];
  for (int i = tid; i < 49 * DWM_CB; i += THREADS) {
    const int tap = i / DWM_CB, cl = i - tap * DWM_CB;
    wsm[i] = c_base + cl < C ? w[(size_t)tap * C + c_base + cl] : 0.f;
  }
  __syncthreads();
  dwm_bf16x8 afrag[CPW][7];
  float bias_v[CPW];
#pragma unroll
  for (int cc = 0; cc < CPW; ++cc) {
    const int cl = wave * CPW + cc;
    bias_v[cc] = (bias && c_base + cl < C) ? bias[c_base + cl] : 0.f;
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      uint32_t pk[4];
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        float v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int t = kq * 8 + e2 * 2 + h - p16;
          const bool in = t >= 0 && t < 7;
          const int tt = in ? t : 0;
          const int tap = FLIP ? 48 - (ky * 7 + tt) : ky * 7 + tt;
          const float wv = wsm[tap * DWM_CB + cl];
          v[h] = in ? wv : 0.f;
        }
        pk[e2] = f32x2_to_bf16x2_bits(v[0], v[1]);
      }
      uint4 q = make_uint4(pk[0], pk[1], pk[2], pk[3]);
      afrag[cc][ky] = __builtin_bit_cast(dwm_bf16x8, q);
    }
  }

  // ---- zero the pad columns of both buffers once (never written afterwards)
  {
    constexpr int PADW = G::PITCH - G::IW;
    for (int i = tid; i < 2 * DWM_CB * DWM_ROWS * PADW; i += THREADS) {
      const int col = G::IW + i % PADW;
      int r = i / PADW;
      const int row = r % DWM_ROWS; r /= DWM_ROWS;
      const int ch = r % DWM_CB;
      const int buf = r / DWM_CB;
      lds[buf][dwm_plane_base(ch, G::PLANE) + row * G::PITCH + col] = 0;
    }
  }

  // staging items: (pixel of the 22 x IW halo tile, 8-channel vector): consecutive lanes = the 4 vectors of a pixel
  constexpr int ITEMS = DWM_ROWS * G::IW * 4;
  constexpr int NIT = (ITEMS + THREADS - 1) / THREADS;
  uint4 stage[NIT];

  // tile coordinates advance incrementally (x fastest): two runtime integer divisions per tile and use were a measurable part
  // of the 0.9 us an empty pass through the tile loop cost
  struct TileAt { int b, y0, x0; };
  auto tile_first = [&](int t) {
    TileAt a;
    const int tx = t % tiles_x;
    t /= tiles_x;
    a.b = t / tiles_y;
    a.y0 = (t % tiles_y) * 16;
    a.x0 = tx * G::TW;
    return a;
  };
  auto tile_next = [&](TileAt a) {
    a.x0 += G::TW;
    if (a.x0 >= tiles_x * G::TW) {
      a.x0 = 0;
      a.y0 += 16;
      if (a.y0 >= tiles_y * 16) { a.y0 = 0; a.b += 1; }
    }
    return a;
  };
  auto load_tile = [&](const TileAt& at, int tv) {
    const int b = at.b, y0 = at.y0, x0 = at.x0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tv + it * THREADS;
      const int cv = i & 3, p = i >> 2;
      const int col = p % G::IW, row = p / G::IW;
      const int gy = y0 + row - 3, gx = x0 + col - 3;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (i < ITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W && c_base + cv * 8 < C)
        v = *reinterpret_cast<const uint4*>(x + (((size_t)b * H + gy) * W + gx) * C + c_base + cv * 8);
      stage[it] = v;
    }
  };
  auto store_tile_lds = [&](int buf, int tv) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tv + it * THREADS;
      if (i < ITEMS) {
        const int cv = i & 3, p = i >> 2;
        const int col = p % G::IW, row = p / G::IW;
        unsigned short* dst = &lds[buf][dwm_plane_base(cv * 8, G::PLANE) + row * G::PITCH + col];
        const uint32_t d[4] = {stage[it].x, stage[it].y, stage[it].z, stage[it].w};
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          dst[(e2 * 2) * G::PLANE] = (unsigned short)(d[e2] & 0xffffu);
          dst[(e2 * 2 + 1) * G::PLANE] = (unsigned short)(d[e2] >> 16);
        }
      }
    }
  };

  TileAt at_cur = tile_first(t_begin), at_nxt = tile_next(at_cur);
  load_tile(at_cur, tid);
  __syncthreads();  // pad zeroing done
  store_tile_lds(0, tid);
  __syncthreads();

  for (int t = t_begin; t < t_end; ++t) {
    const int cur = (t - t_begin) & 1;
    const bool more = t + 1 < t_end;
    // the per-thread index arithmetic of the staging / gather loops is cheap; kept loop-invariant (hipcc hoists it out of the tile
    // loop) it occupies ~40 registers for the whole launch — an opaque copy of the thread index per tile makes it local again
    int tv = tid;
    asm volatile("" : "+v"(tv));
    if (more) load_tile(at_nxt, tv);  // in flight under the MFMAs

    // ---- compute: this wave's two channels, NXT x tiles of 16 columns, 16 rows
#pragma unroll
    for (int cc = 0; cc < CPW; ++cc) {
      const int chl = wave * CPW + cc;
      const unsigned short* plane = &lds[cur][dwm_plane_base(chl, G::PLANE)];
      dwm_f32x4 acc[NXT];
#pragma unroll
      for (int xt = 0; xt < NXT; ++xt) acc[xt] = dwm_f32x4{bias_v[cc], bias_v[cc], bias_v[cc], bias_v[cc]};
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
        for (int xt = 0; xt < NXT; ++xt) {
          const uint4 q = *reinterpret_cast<const uint4*>(plane + (p16 + ky) * G::PITCH + xt * 16 + kq * 8);
          acc[xt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[cc][ky], __builtin_bit_cast(dwm_bf16x8, q), acc[xt], 0, 0, 0);
        }
      }
      // D: lane (row j = p16, columns kq*4 .. +3) -> the channel's own plane, row j, tile-local column
#pragma unroll
      for (int xt = 0; xt < NXT; ++xt) {
        uint2 o;
        o.x = f32x2_to_bf16x2_bits(acc[xt][0], acc[xt][1]);
        o.y = f32x2_to_bf16x2_bits(acc[xt][2], acc[xt][3]);
        *reinterpret_cast<uint2*>(const_cast<unsigned short*>(plane) + p16 * G::PITCH + xt * 16 + kq * 4) = o;
      }
    }
    if (more) store_tile_lds(cur ^ 1, tv);
    __syncthreads();

    // ---- gather channel vectors back and store
    {
      const int b = at_cur.b, y0 = at_cur.y0, x0 = at_cur.x0;
      constexpr int OITEMS = 16 * G::TW * 4;
#pragma unroll
      for (int it = 0; it < (OITEMS + THREADS - 1) / THREADS; ++it) {
        const int i = tv + it * THREADS;
        const int cv = i & 3, p = i >> 2;
        const int col = p % G::TW, row = p / G::TW;
        const int gy = y0 + row, gx = x0 + col;
        if (i < OITEMS && gy < H && gx < W && c_base + cv * 8 < C) {
          const unsigned short* src = &lds[cur][dwm_plane_base(cv * 8, G::PLANE) + row * G::PITCH + col];
          uint32_t d[4];
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2)
            d[e2] = (uint32_t)src[(e2 * 2) * G::PLANE] | ((uint32_t)src[(e2 * 2 + 1) * G::PLANE] << 16);
          const size_t off = (((size_t)b * H + gy) * W + gx) * C + c_base + cv * 8;
          uint4 o = make_uint4(d[0], d[1], d[2], d[3]);
          if (add) {
            float a[8], f[8];
            unpack<bf16_t>(*reinterpret_cast<const uint4*>(add + off), a);
            unpack<bf16_t>(o, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] += a[j];
            o = pack<bf16_t>(f);
          }
          *reinterpret_cast<uint4*>(y + off) = o;
        }
      }
    }
    __syncthreads();  // the planes of `cur` are overwritten by the staging of tile t + 2
    at_cur = at_nxt;
    at_nxt = tile_next(at_nxt);
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// The same forward / data gradient with its tiles fetched by LDS-DMA (round 4).  The register-staged kernel above issues the
// loads of tile t + 1 at the top of iteration t and needs them 0.2 us later (14 - 28 MFMAs per wave): at the 128-register cap
// of a 16-wave workgroup there is no room for a second register stage, so every iteration exposes most of a memory round trip
// (SQ counters: 58 % of the wave cycles in s_waitcnt vmcnt).  `global_load_lds_dwordx4` needs no registers: a lane hands over a
// 16-byte global address and the data lands at (wave-uniform LDS base + lane * 16).  So:
//   * the halo tile (22 x 22 pixels x 64 B of the slab) arrives pixel-major in a RAW buffer, two tiles ahead of its use
//     (raw[2] alternate; the shortcut operand `add` of the data gradient the same way, one tile ahead — the old kernel fetched it
//     with a dependent load inside the store loop);
//   * out-of-image halo pixels read a 64-byte block of zeros in global memory (a DMA cannot write a constant);
//   * an LDS -> LDS pass (16-byte reads, the same 2-byte transposing writes as above) fills the per-channel planes; MFMAs, the
//     write-back into the planes and the gather to 16-byte channel vectors are unchanged;
//   * waits are counted: loads return in order, so "at most n operations outstanding" with n = the DMA pieces issued in THIS
//     iteration proves every older DMA piece has landed whatever the (unordered) stores in between are doing; each wave waits
//     for its own pieces, the workgroup barrier that follows makes them visible to everyone.
// 16 x 16 tiles only (raw 2 x 31 KB + add 2 x 16 KB + planes 55 KB + taps 6 KB = 155 KB of the CU's 160).  The buffers are
// separate __shared__ objects and the tile loop is unrolled by two so that every access names its buffer at compile time
// (through one array every ds_read behind a DMA costs an s_waitcnt vmcnt(0): csrc/mlp.hip).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(64))) unsigned int g_dwm_zero_block[16];  // zero-initialised: the source of halo pixels outside the image

constexpr int DWD_IW = 22, DWD_PITCH = 40, DWD_PLANE = DWM_ROWS * DWD_PITCH;
// CB channels per workgroup (a wave owns two): 32 = the 64-byte slab of the register-staged kernel, one 16-wave workgroup per CU.
// (CB = 16 — half slabs, 8 waves and 79 KB of LDS per workgroup, so that TWO workgroups share a CU and one's transposition / store
// phase runs under the other's MFMA phase — compiles from the same source and was measured: 32-byte row pieces cost more than the
// overlap buys, forward 266 -> 299 us, data gradient 300 -> 435 us at 64 x 64 x 96, B = 512; only CB = 32 is instantiated.)
template <int CB>
struct DwdGeom {
  static constexpr int THREADS = CB * 32, NW = CB / 2, PP = CB / 8;   // PP: 16-byte pieces per pixel
  static constexpr int XITEMS = DWM_ROWS * DWD_IW * PP;               // pieces of a halo tile (1936 / 968)
  static constexpr int XINSTR = (XITEMS + 63) / 64;                   // wave-wide DMA instructions per halo tile (31 / 16)
  static constexpr int XIT = (XINSTR + NW - 1) / NW;                  // ... per wave, at most
  static constexpr int RAW = XINSTR * 1024, ADD = 256 * PP * 16;
  static_assert(256 * PP == THREADS, "one output piece per thread");
};

// One wave-wide DMA piece: lane l's 16 bytes at `gsrc` land at LDS byte address lds_base + l * 16 (lds_base wave-uniform).  Issued
// as inline assembly ON PURPOSE: hipcc puts `s_waitcnt vmcnt(0)` in front of LDS reads behind a `__builtin_amdgcn_global_load_lds`
// whenever it cannot prove which LDS object the DMA writes (it could not here), which serialises the prefetch; this kernel counts
// its own waits (dwd_wait_older_than) and orders LDS traffic with raw barriers.
static __device__ __forceinline__ void dwd_dma16(const void* gsrc, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_base) : "memory", "m0");
}
static __device__ __forceinline__ uint32_t dwd_lds_addr(const void* p) {
  return (uint32_t)(size_t)(const __attribute__((address_space(3))) char*)p;
}
// LDS traffic of this wave done, then the workgroup barrier — WITHOUT the vmcnt(0) a __syncthreads() carries
static __device__ __forceinline__ void dwd_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// PAIR (round 6): the LDS -> LDS transposition and the gather back out of the planes move TWO horizontally adjacent pixels per
// access — 32-bit words [col, col + 1] of a channel plane instead of 2-byte elements: 8 ds_write_b32 per 2 x 16-byte piece where
// the first version issued 16 ds_write_b16, 8 ds_read_b32 by half of the threads where all of them issued 8 ds_read_u16.  A tile
// of this kernel is ~43 LDS instructions per wave against 14 MFMAs (profiles/r05_sq_counters.txt: 4.14e8 / 1.42e8), 24 of them these
// 2-byte accesses.  Planes of PAIR builds are skewed by 32 elements per 8 channels (the four 8-channel vectors of 16 pixel pairs
// land on 4 x 16 distinct banks).  Same arithmetic, bit-identical outputs.
template <bool PAIR>
static __device__ __forceinline__ int dwd_plane_base(int ch) { return ch * DWD_PLANE + (ch >> 3) * (PAIR ? 32 : 16); }

template <bool FLIP, int CB, bool PAIR = false>
__global__ __launch_bounds__(CB * 32, CB == 16 ? 4 : 1) void dwconv7_mfma_dma_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                                       const float* __restrict__ bias,
                                                                       const bf16_t* __restrict__ add, bf16_t* __restrict__ y,
                                                                       int B, int H, int W, int C, int nslab, int tiles_total,
                                                                       int tiles_per_wg) {
  typedef DwdGeom<CB> G;
  constexpr int THREADS = G::THREADS, CPW = 2, PP = G::PP;
  __shared__ __attribute__((aligned(1024))) char raw0[G::RAW];
  __shared__ __attribute__((aligned(1024))) char raw1[G::RAW];
  __shared__ __attribute__((aligned(1024))) char radd0[G::ADD];
  __shared__ __attribute__((aligned(1024))) char radd1[G::ADD];
  __shared__ __attribute__((aligned(16))) unsigned short planes[CB * DWD_PLANE + (CB / 8) * 32];
  __shared__ float wsm[49 * CB];

  int bid = blockIdx.x;
  if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
  const int slab = bid % nslab;
  const int chunk = bid / nslab;
  const int t_begin = chunk * tiles_per_wg;
  const int t_end = min(tiles_total, t_begin + tiles_per_wg);
  if (t_begin >= t_end) return;
  const int c_base = slab * CB;
  const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p16 = lane & 15, kq = lane >> 4;

  for (int i = tid; i < 49 * CB; i += THREADS) {
    const int tap = i / CB, cl = i - tap * CB;
    wsm[i] = w[(size_t)tap * C + c_base + cl];
  }
  __syncthreads();
  dwm_bf16x8 afrag[CPW][7];
  float bias_v[CPW];
#pragma unroll
  for (int cc = 0; cc < CPW; ++cc) {
    const int cl = wave * CPW + cc;
    bias_v[cc] = bias ? bias[c_base + cl] : 0.f;
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      uint32_t pk[4];
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        float v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int t = kq * 8 + e2 * 2 + h - p16;
          const bool in = t >= 0 && t < 7;
          const int tt = in ? t : 0;
          const int tap = FLIP ? 48 - (ky * 7 + tt) : ky * 7 + tt;
          const float wv = wsm[tap * CB + cl];
          v[h] = in ? wv : 0.f;
        }
        pk[e2] = f32x2_to_bf16x2_bits(v[0], v[1]);
      }
      uint4 q = make_uint4(pk[0], pk[1], pk[2], pk[3]);
      afrag[cc][ky] = __builtin_bit_cast(dwm_bf16x8, q);
    }
  }
  {  // pad columns of the planes: read by the K = 32 window, multiplied by zeros of A, must be finite; never written afterwards
    constexpr int PADW = DWD_PITCH - DWD_IW;
    for (int i = tid; i < CB * DWM_ROWS * PADW; i += THREADS) {
      const int col = DWD_IW + i % PADW;
      int r = i / PADW;
      const int row = r % DWM_ROWS;
      const int ch = r / DWM_ROWS;
      planes[dwd_plane_base<PAIR>(ch) + row * DWD_PITCH + col] = 0;
    }
  }

  struct TileAt { int b, y0, x0; };
  auto tile_first = [&](int t) {
    TileAt a;
    const int tx = t % tiles_x;
    t /= tiles_x;
    a.b = t / tiles_y;
    a.y0 = (t % tiles_y) * 16;
    a.x0 = tx * 16;
    return a;
  };
  auto tile_next = [&](TileAt a) {
    a.x0 += 16;
    if (a.x0 >= tiles_x * 16) {
      a.x0 = 0;
      a.y0 += 16;
      if (a.y0 >= tiles_y * 16) { a.y0 = 0; a.b += 1; }
    }
    return a;
  };
  const char* zsrc = reinterpret_cast<const char*>(g_dwm_zero_block);
  // this wave's DMA pieces of a halo tile are the items tid + it * THREADS (the items the same thread transposes later)
  int nx = 0;                                              // ... how many of them (wave-uniform)
#pragma unroll
  for (int it = 0; it < G::XIT; ++it) nx += wave + G::NW * it < G::XINSTR ? 1 : 0;
  // Round 6: addresses = one 64-bit SCALAR tile origin + a 32-bit byte offset that a thread computes ONCE.  Written per piece
  // as (((b * H + gy) * W + gx) * C + ...) the index chain compiled to seven quarter-rate 32 / 64-bit multiplies per address, four
  // addresses per tile: ~450 issue cycles per wave and tile, four waves per SIMD (the kernel is bound by its instruction issue:
  // profiles/r05_sq_counters.txt, waves active 19 % each).  (dispatch: 32 rows of the image stay below 2^31 bytes)
  // (few live registers on purpose: the kernel sits at its 128-register cap, a spilled thread constant is reloaded from scratch
  // in front of the DMA issue and costs more than the multiplies did — measured: +30 % per launch with ten constants held)
  static_assert(G::XIT <= 2 && THREADS % PP == 0, "row / column of the thread's two halo pieces packed into one register");
  int xbyte[G::XIT];
  uint32_t xrc = 0;  // (row, col) of halo piece `it` in bits [16 it, 16 it + 16)
#pragma unroll
  for (int it = 0; it < G::XIT; ++it) {
    const int i = tid + it * THREADS;
    const int cv = i % PP, p = i / PP;
    const int col = p % DWD_IW, row = p / DWD_IW;
    xrc |= (uint32_t)((row << 8) | col) << (16 * it);
    xbyte[it] = (((row - 3) * W + (col - 3)) * C + cv * 8) * 2;
  }
  const int ocv = tid % PP;
#define ocol ((tid / PP) & 15)
#define orow ((tid / PP) >> 4)
  const int obyte = ((orow * W + ocol) * C + ocv * 8) * 2;
  auto tile_origin = [&](const TileAt& at) -> long long {   // element index of (b, y0, x0, c_base): scalar
    return (((long long)at.b * H + at.y0) * W + at.x0) * (long long)C + c_base;
  };
  auto issue_x = [&](const TileAt& at, uint32_t dst) {
    const char* xt = reinterpret_cast<const char*>(x + tile_origin(at));
#pragma unroll
    for (int it = 0; it < G::XIT; ++it) {
      if (wave + G::NW * it >= G::XINSTR) break;
      const int i = tid + it * THREADS;
      if (i < G::XITEMS) {
        const int gy = at.y0 + (int)((xrc >> (16 * it + 8)) & 0xff) - 3, gx = at.x0 + (int)((xrc >> (16 * it)) & 0xff) - 3;
        const bool in = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const char* src = in ? xt + xbyte[it] : zsrc + ocv * 16;
        dwd_dma16(src, dst + (wave + G::NW * it) * 1024);
      }
    }
  };
  auto issue_add = [&](const TileAt& at, uint32_t dst) {
    const bool in = at.y0 + orow < H && at.x0 + ocol < W;
    const char* src = in ? reinterpret_cast<const char*>(add + tile_origin(at)) + obyte : zsrc + ocv * 16;
    dwd_dma16(src, dst + wave * 1024);
  };
  auto transpose_in = [&](const char* rawb) {
    if constexpr (PAIR) {
      constexpr int XPAIRS = DWM_ROWS * (DWD_IW / 2) * PP;  // (pixel pair, 8-channel vector) items of a halo tile: 968 at CB = 32
      static_assert(!PAIR || (XPAIRS <= THREADS && DWD_IW % 2 == 0), "one pixel pair per thread");
      if (tid < XPAIRS) {
        const int cv = tid % PP, pp = tid / PP;
        const int colp = pp % (DWD_IW / 2), row = pp / (DWD_IW / 2);
        const int ia = (row * DWD_IW + 2 * colp) * PP + cv;     // raw piece of the left pixel; the right one is PP pieces on
        const uint4 va = *reinterpret_cast<const uint4*>(rawb + (size_t)ia * 16);
        const uint4 vb = *reinterpret_cast<const uint4*>(rawb + (size_t)(ia + PP) * 16);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&planes[dwd_plane_base<PAIR>(cv * 8) + row * DWD_PITCH + 2 * colp]);
        const uint32_t da[4] = {va.x, va.y, va.z, va.w}, db[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          dst[(e2 * 2) * (DWD_PLANE / 2)] = __builtin_amdgcn_perm(db[e2], da[e2], 0x05040100u);      // {a.lo, b.lo}
          dst[(e2 * 2 + 1) * (DWD_PLANE / 2)] = __builtin_amdgcn_perm(db[e2], da[e2], 0x07060302u);  // {a.hi, b.hi}
        }
      }
      return;
    }
#pragma unroll
    for (int it = 0; it < G::XIT; ++it) {
      const int i = tid + it * THREADS;
      if (i < G::XITEMS) {
        const int cv = i % PP, p = i / PP;
        const int col = p % DWD_IW, row = p / DWD_IW;
        const uint4 v = *reinterpret_cast<const uint4*>(rawb + (size_t)i * 16);
        unsigned short* dst = &planes[dwd_plane_base<PAIR>(cv * 8) + row * DWD_PITCH + col];
        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          dst[(e2 * 2) * DWD_PLANE] = (unsigned short)(d[e2] & 0xffffu);
          dst[(e2 * 2 + 1) * DWD_PLANE] = (unsigned short)(d[e2] >> 16);
        }
      }
    }
  };
  auto wait_older_than = [&](int n) {  // every vector-memory operation but the n youngest has completed (n is wave-uniform)
    static_assert(G::XIT <= 2, "the wait below knows 0 .. 3 pieces in flight");
    if (n >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  // one tile: `rawc` / `addc` hold tile t (complete for this wave since the previous iteration's wait), `rawn` / `addn` receive
  // tile t + 1's shortcut operand now and tile t + 2's halo tile once rawc has been transposed
  auto tile_step = [&](int t, const TileAt& at, const TileAt& at1, const TileAt& at2, char* rawc, char* addc, char* addn) {
    dwd_barrier();                                    // A: gather(t - 1) done by everyone; everyone's pieces of tile t landed
    const bool more1 = t + 1 < t_end, more2 = t + 2 < t_end;
    if (add && more1) issue_add(at1, dwd_lds_addr(addn));
    transpose_in(rawc);
    dwd_barrier();                                    // D: planes complete; rawc free
    if (more2) issue_x(at2, dwd_lds_addr(rawc));
#pragma unroll
    for (int cc = 0; cc < CPW; ++cc) {
      const int chl = wave * CPW + cc;
      unsigned short* plane = &planes[dwd_plane_base<PAIR>(chl)];
      dwm_f32x4 acc = dwm_f32x4{bias_v[cc], bias_v[cc], bias_v[cc], bias_v[cc]};
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) {
        const uint4 q = *reinterpret_cast<const uint4*>(plane + (p16 + ky) * DWD_PITCH + kq * 8);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[cc][ky], __builtin_bit_cast(dwm_bf16x8, q), acc, 0, 0, 0);
      }
      uint2 o;
      o.x = f32x2_to_bf16x2_bits(acc[0], acc[1]);
      o.y = f32x2_to_bf16x2_bits(acc[2], acc[3]);
      *reinterpret_cast<uint2*>(plane + p16 * DWD_PITCH + kq * 4) = o;
    }
    // tile t + 1 (halo pieces issued one iteration ago, shortcut piece at the top of this one) must have landed before the
    // next barrier A; what may stay in flight are the pieces issued in THIS iteration behind it: the halo pieces of t + 2 —
    // and, issued BEFORE the halo pieces of t + 1?  No: order of issue is add(t+1) [this iteration], x(t+2) [this iteration];
    // x(t+1) and add(t) are older.  add(t+1) is needed by the NEXT gather only, so it may stay in flight as well.
    wait_older_than((add && more1 ? 1 : 0) + (more2 ? nx : 0));
    dwd_barrier();                                    // H: every plane holds its channel's outputs
    if constexpr (PAIR) {
      if (tid < 128 * PP) {  // (pixel pair, 8-channel vector) items of the 16 x 16 outputs
        const int cv = tid % PP, pp = tid / PP;
        const int colp = pp & 7, row = pp >> 3;
        const int gy = at.y0 + row, gx = at.x0 + 2 * colp;
        if (gy < H && gx < W) {
          const uint32_t* src = reinterpret_cast<const uint32_t*>(&planes[dwd_plane_base<PAIR>(cv * 8) + row * DWD_PITCH + 2 * colp]);
          uint32_t w[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e] = src[e * (DWD_PLANE / 2)];
          uint4 oa, ob;
          oa.x = __builtin_amdgcn_perm(w[1], w[0], 0x05040100u); ob.x = __builtin_amdgcn_perm(w[1], w[0], 0x07060302u);
          oa.y = __builtin_amdgcn_perm(w[3], w[2], 0x05040100u); ob.y = __builtin_amdgcn_perm(w[3], w[2], 0x07060302u);
          oa.z = __builtin_amdgcn_perm(w[5], w[4], 0x05040100u); ob.z = __builtin_amdgcn_perm(w[5], w[4], 0x07060302u);
          oa.w = __builtin_amdgcn_perm(w[7], w[6], 0x05040100u); ob.w = __builtin_amdgcn_perm(w[7], w[6], 0x07060302u);
          const int ip = (row * 16 + 2 * colp) * PP + cv;  // shortcut piece of the left pixel in `addc`
          const size_t off = (size_t)tile_origin(at) + (size_t)((row * W + 2 * colp) * C + cv * 8);
          if (add) {
            float a[8], f[8];
            unpack<bf16_t>(*reinterpret_cast<const uint4*>(addc + (size_t)ip * 16), a);
            unpack<bf16_t>(oa, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] += a[j];
            oa = pack<bf16_t>(f);
          }
          *reinterpret_cast<uint4*>(y + off) = oa;
          if (gx + 1 < W) {
            if (add) {
              float a[8], f[8];
              unpack<bf16_t>(*reinterpret_cast<const uint4*>(addc + (size_t)(ip + PP) * 16), a);
              unpack<bf16_t>(ob, f);
#pragma unroll
              for (int j = 0; j < 8; ++j) f[j] += a[j];
              ob = pack<bf16_t>(f);
            }
            *reinterpret_cast<uint4*>(y + off + C) = ob;
          }
        }
      }
    } else {
      const int cv = ocv, col = ocol, row = orow;
      const int gy = at.y0 + row, gx = at.x0 + col;
      if (gy < H && gx < W) {
        const unsigned short* src = &planes[dwd_plane_base<PAIR>(cv * 8) + row * DWD_PITCH + col];
        uint32_t d[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2)
          d[e2] = (uint32_t)src[(e2 * 2) * DWD_PLANE] | ((uint32_t)src[(e2 * 2 + 1) * DWD_PLANE] << 16);
        uint4 o = make_uint4(d[0], d[1], d[2], d[3]);
        if (add) {
          float a[8], f[8];
          unpack<bf16_t>(*reinterpret_cast<const uint4*>(addc + (size_t)tid * 16), a);
          unpack<bf16_t>(o, f);
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] += a[j];
          o = pack<bf16_t>(f);
        }
        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(y + tile_origin(at)) + obyte) = o;
      }
    }
  };

  TileAt a0 = tile_first(t_begin), a1 = tile_next(a0), a2 = tile_next(a1);
  __syncthreads();                                    // pad zeroing done (no DMA in flight yet)
  issue_x(a0, dwd_lds_addr(raw0));
  if (add) issue_add(a0, dwd_lds_addr(radd0));
  if (t_begin + 1 < t_end) issue_x(a1, dwd_lds_addr(raw1));
  wait_older_than(t_begin + 1 < t_end ? nx : 0);      // tile t_begin complete for this wave
  for (int t = t_begin; t < t_end; t += 2) {
    tile_step(t, a0, a1, a2, raw0, radd0, radd1);
    a0 = a1; a1 = a2; a2 = tile_next(a2);
    if (t + 1 < t_end) {
      tile_step(t + 1, a0, a1, a2, raw1, radd1, radd0);
      a0 = a1; a1 = a2; a2 = tile_next(a2);
    }
  }
}


#undef ocol
#undef orow
// ---------------------------------------------------------------------------------------------------------------------
// weight gradient on the matrix cores.  dw_c[ky][kx] = sum_{y,x} dy_c[y][x] * in_c[y + ky - 3][x + kx - 3]: for one
// channel and one ky, contract over 16 image ROWS with v_mfma_f32_16x16x16_bf16:
//
//     G_ky[u][v] += sum_{row} in_c[row + ky][vb + u] * dy_c[row][vb + v]          (plane coordinates, block start vb)
//
// and dw_c[ky][kx] is the kx-th diagonal of G_ky (u - v = kx).  Blocks start every 8 columns and all accumulate into the
// SAME G_ky (the diagonals of columns v < 8 of each block are complete, u = v + kx <= 14; columns v >= 8 belong to the next
// block and are ignored at the end), so a channel needs 7 accumulators (+ 1 with an all-ones A for db = sum dy), kept in
// registers for the whole launch; the diagonals are extracted once, at the end, through LDS atomics into one workspace
// row per tile range — no global atomics, a fixed summation order.
// Both operands are row-major [row][x] planes whose fragments run along the contraction (row) axis: they are fetched
// with the gfx950 transpose read ds_read_b64_tr_b16 (lane (r, cseg) of a 16-lane group supplies the address of 4
// contiguous columns of row r; lane l receives column l & 15 of the group's 4 rows) — the shift by ky is a row offset of
// the address, never a misaligned access.  Same persistent 16-wave workgroups and 2-byte LDS transposition as the forward;
// one LDS buffer (in + dy planes of 32 channels = 117 KB), the next tile's global loads are in flight under the MFMAs.
// ---------------------------------------------------------------------------------------------------------------------
typedef short dwm_s16x4 __attribute__((ext_vector_type(4)));
constexpr int DWM_WPITCH = 48;  // elements per plane row: tr reads of 4 rows x 16 columns (and of the paired group) hit distinct banks

template <int NXT>
__global__ __launch_bounds__(DWM_THREADS) void dwconv7_wgrad_mfma_kernel(const bf16_t* __restrict__ dy,
                                                                         const bf16_t* __restrict__ x, float* __restrict__ ws,
                                                                         int B, int H, int W, int C, int nslab, int tiles_total,
                                                                         int tiles_per_wg) {
  constexpr int TW = 16 * NXT, IW = TW + 6, P = DWM_WPITCH;
  constexpr int XPLANE = DWM_ROWS * P, DPLANE = 16 * P;
  constexpr int XBUF = DWM_CB * XPLANE + (DWM_CB / 8) * 16, DBUF = DWM_CB * DPLANE + (DWM_CB / 8) * 16;
  __shared__ __attribute__((aligned(16))) unsigned short xl[XBUF];
  __shared__ __attribute__((aligned(16))) unsigned short dl[DBUF];
  __shared__ float red[DWM_CB * 50];

  int bid = blockIdx.x;
  if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
  const int slab = bid % nslab;
  const int chunk = bid / nslab;
  const int t_begin = chunk * tiles_per_wg;
  const int t_end = min(tiles_total, t_begin + tiles_per_wg);
  if (t_begin >= t_end) return;
  const int c_base = slab * DWM_CB;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + 15) / 16;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int p16 = lane & 15, kq = lane >> 4;
  const int tr_r = (lane & 15) >> 2, tr_c = (lane & 3) * 4;  // this lane's address role in a transpose read

  for (int i = tid; i < XBUF; i += DWM_THREADS) xl[i] = 0;   // pads (and everything else) finite before the first MFMA
  for (int i = tid; i < DBUF; i += DWM_THREADS) dl[i] = 0;
  for (int i = tid; i < DWM_CB * 50; i += DWM_THREADS) red[i] = 0.f;

  constexpr int XITEMS = DWM_ROWS * IW * 4, DITEMS = 16 * TW * 4, ITEMS = XITEMS + DITEMS;
  constexpr int NIT = (ITEMS + DWM_THREADS - 1) / DWM_THREADS;
  uint4 stage[NIT];

  auto tile_origin = [&](int t, int& b, int& y0, int& x0) {
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    b = t / tiles_y;
    y0 = ty * 16;
    x0 = tx * TW;
  };
  auto load_tile = [&](int t) {
    int b, y0, x0;
    tile_origin(t, b, y0, x0);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * DWM_THREADS;
      const bool is_x = i < XITEMS;
      const int k = is_x ? i : i - XITEMS;
      const int cv = k & 3, p = k >> 2;
      const int wdt = is_x ? IW : TW;
      const int col = p % wdt, row = p / wdt;
      const int gy = y0 + row - (is_x ? 3 : 0), gx = x0 + col - (is_x ? 3 : 0);
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (i < ITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W && c_base + cv * 8 < C)
        v = *reinterpret_cast<const uint4*>((is_x ? x : dy) + (((size_t)b * H + gy) * W + gx) * C + c_base + cv * 8);
      stage[it] = v;
    }
  };
  auto store_tile_lds = [&]() {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * DWM_THREADS;
      if (i < ITEMS) {
        const bool is_x = i < XITEMS;
        const int k = is_x ? i : i - XITEMS;
        const int cv = k & 3, p = k >> 2;
        const int wdt = is_x ? IW : TW;
        const int col = p % wdt, row = p / wdt;
        const int plane = is_x ? XPLANE : DPLANE;
        unsigned short* dst = (is_x ? xl : dl) + dwm_plane_base(cv * 8, plane) + row * P + col;
        const uint32_t d[4] = {stage[it].x, stage[it].y, stage[it].z, stage[it].w};
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          dst[(e2 * 2) * plane] = (unsigned short)(d[e2] & 0xffffu);
          dst[(e2 * 2 + 1) * plane] = (unsigned short)(d[e2] >> 16);
        }
      }
    }
  };

  dwm_f32x4 acc[2][8];
#pragma unroll
  for (int cc = 0; cc < 2; ++cc)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[cc][k] = dwm_f32x4{0.f, 0.f, 0.f, 0.f};
  const dwm_s16x4 ones = {(short)0x3F80, (short)0x3F80, (short)0x3F80, (short)0x3F80};

  load_tile(t_begin);
  __syncthreads();
  for (int t = t_begin; t < t_end; ++t) {
    store_tile_lds();
    __syncthreads();
    if (t + 1 < t_end) load_tile(t + 1);  // in flight under the MFMAs
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int chl = wave * 2 + cc;
      const unsigned short* xp = xl + dwm_plane_base(chl, XPLANE) + (kq * 4 + tr_r) * P + tr_c;
      const unsigned short* dp = dl + dwm_plane_base(chl, DPLANE) + (kq * 4 + tr_r) * P + tr_c;
#pragma unroll
      for (int vb = 0; vb < TW; vb += 8) {
        const dwm_s16x4 bfr = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) dwm_s16x4*)(dp + vb));
        acc[cc][7] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ones, bfr, acc[cc][7], 0, 0, 0);
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
          const dwm_s16x4 afr = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) dwm_s16x4*)(xp + ky * P + vb));
          acc[cc][ky] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(afr, bfr, acc[cc][ky], 0, 0, 0);
        }
      }
    }
    __syncthreads();  // planes are rewritten by the next tile
  }

  // ---- diagonals: lane holds G[u = kq*4 + r][v = p16]; column v < 8, kx = u - v in [0, 7)
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
    const int chl = wave * 2 + cc;
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kx = kq * 4 + r - p16;
        if (p16 < 8 && kx >= 0 && kx < 7) atomicAdd(&red[chl * 50 + ky * 7 + kx], acc[cc][ky][r]);
      }
    if (p16 < 8 && kq == 0) atomicAdd(&red[chl * 50 + 49], acc[cc][7][0]);  // every row u of the ones product = column sum
  }
  __syncthreads();
  float* row = ws + (size_t)chunk * 50 * C;
  for (int i = tid; i < DWM_CB * 50; i += DWM_THREADS) {
    const int chl = i % DWM_CB, n = i / DWM_CB;
    if (c_base + chl < C) row[(size_t)n * C + c_base + chl] = red[chl * 50 + n];
  }
}

extern int g_vsx_dw_mfma;

// *taken = 1 when the launch went to the MFMA path, 0 when the caller should use the VALU stencil; returns 0 or an error code
int vsx_dwconv7_mfma_try(const void* x, const float* w, const float* bias, const void* add, void* y, int B, int H, int W, int C,
                         bool flip, hipStream_t s, int* taken) {
  *taken = 0;
  if (!(g_vsx_dw_mfma & 1) || H < 16 || W < 16 || (C & 7)) return 0;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  // bit 3: tiles by LDS-DMA, two ahead (16 x 16 tiles; whole slabs only) — where that kernel wins (tools/perf_dw.py, B = 512, us,
  // register-staged / DMA): the data gradient with its shortcut operand at every size (366 / 314 at 64 x 64 x 96, 176 / 163 at
  // 32 x 32 x 192, 107 / 83 at 16 x 16 x 384, 721 / 652 at 64 x 64 x 224: the operand rides the DMA instead of a dependent load in
  // the store loop) and maps of one tile (forward 75 / 69 at 16 x 16 x 384); the forward of larger maps keeps the 16 x 32 tiles of
  // the register-staged kernel (256 / 282, 125 / 133, 521 / 547: a smaller halo and half the barriers per pixel).  Bit 4: the
  // DMA kernel wherever it can run (A/B knob)
  const bool dma = C % DWM_CB == 0 && (long)32 * W * C * 2 < 0x7fffffffL &&
                   (((g_vsx_dw_mfma & 8) && (add != nullptr || (H <= 16 && W <= 16))) || (g_vsx_dw_mfma & 16));
  const int nxt = (W >= 24 && !dma) ? 2 : 1;
  const int tw = 16 * nxt;
  const int nslab = vsx_cdiv(C, DWM_CB);
  const long tiles_l = (long)B * vsx_cdiv(H, 16) * vsx_cdiv(W, tw);
  if (tiles_l > 0x7fffffffL) return 0;
  const int tiles = (int)tiles_l;
  int chunks = n_cu / nslab;
  if (chunks < 1) chunks = 1;
  if (chunks > tiles) chunks = tiles;
  const int per = vsx_cdiv(tiles, chunks);
  chunks = vsx_cdiv(tiles, per);
  int grid = chunks * nslab;
  // keep the XCD remap valid: pad the grid to a multiple of 8 (surplus workgroups exit at once)
  grid = (grid + 7) & ~7;
#define DWM_LAUNCH(NXT, FLIP)                                                                                          \
  hipLaunchKernelGGL((dwconv7_mfma_kernel<NXT, FLIP>), dim3(grid), dim3(DWM_THREADS), 0, s, (const bf16_t*)x, w, bias,  \
                     (const bf16_t*)add, (bf16_t*)y, B, H, W, C, nslab, tiles, per)
#define DWD_LAUNCH(FLIP, CBV, PAIRV)                                                                                            \
  hipLaunchKernelGGL((dwconv7_mfma_dma_kernel<FLIP, CBV, PAIRV>), dim3(grid), dim3(CBV * 32), 0, s, (const bf16_t*)x, w, bias,   \
                     (const bf16_t*)add, (bf16_t*)y, B, H, W, C, nslab, tiles, per)
  if (dma && (g_vsx_dw_mfma & 32)) {  // bit 5 (round 6): two pixels per LDS access in the transposition / gather phases
    if (flip) DWD_LAUNCH(true, 32, true); else DWD_LAUNCH(false, 32, true);
  } else if (dma) {
    if (flip) DWD_LAUNCH(true, 32, false); else DWD_LAUNCH(false, 32, false);
  } else if (nxt == 2) {
    if (flip) DWM_LAUNCH(2, true); else DWM_LAUNCH(2, false);
  } else {
    if (flip) DWM_LAUNCH(1, true); else DWM_LAUNCH(1, false);
  }
#undef DWM_LAUNCH
  VSX_LAUNCH_CHECK();
  *taken = 1;
  return 0;
}

// *taken = number of workspace rows written (to be folded by the caller), 0 when the VALU kernel should run instead
int vsx_dwconv7_wgrad_mfma_try(const void* dy, const void* x, float* ws, int ws_rows, int B, int H, int W, int C,
                               hipStream_t s, int* taken) {
  *taken = 0;
  if (!(g_vsx_dw_mfma & 2) || H < 16 || W < 16 || (C & 7)) return 0;
  int dev = 0;
  static int n_cu = 0;
  if (n_cu == 0) {
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int nxt = (W >= 24 && !(g_vsx_dw_mfma & 4)) ? 2 : 1;  // bit 2: 16-column tiles at every width (A/B knob)
  const int nslab = vsx_cdiv(C, DWM_CB);
  const long tiles_l = (long)B * vsx_cdiv(H, 16) * vsx_cdiv(W, 16 * nxt);
  if (tiles_l > 0x7fffffffL) return 0;
  const int tiles = (int)tiles_l;
  int chunks = n_cu / nslab;
  if (chunks < 1) chunks = 1;
  if (chunks > tiles) chunks = tiles;
  if (chunks > ws_rows) chunks = ws_rows;
  const int per = vsx_cdiv(tiles, chunks);
  chunks = vsx_cdiv(tiles, per);
  const int grid = (chunks * nslab + 7) & ~7;
  if (nxt == 2)
    hipLaunchKernelGGL((dwconv7_wgrad_mfma_kernel<2>), dim3(grid), dim3(DWM_THREADS), 0, s, (const bf16_t*)dy, (const bf16_t*)x,
                       ws, B, H, W, C, nslab, tiles, per);
  else
    hipLaunchKernelGGL((dwconv7_wgrad_mfma_kernel<1>), dim3(grid), dim3(DWM_THREADS), 0, s, (const bf16_t*)dy, (const bf16_t*)x,
                       ws, B, H, W, C, nslab, tiles, per);
  VSX_LAUNCH_CHECK();
  *taken = chunks;
  return 0;
}
