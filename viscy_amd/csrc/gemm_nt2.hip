// gemm_nt, second generation, for the launches that carry the step (bf16, plain row operands):
//
//     C[M, N] = A[M, K] . B[N, K]^T  (+ the fused epilogues of vsx_gemm_nt)
//
// What bounds the first-generation kernel (gemm.hip: 128 x 128 tile, global -> VGPR -> LDS staging) was measured in round 3
// with probe builds of this kernel (MFMA off / DMA off / stores off, tools/perf_nt_gen2.py): the K loop runs at the speed of
// its OPERAND STREAM, L2 -> LDS, which saturates near 10 TB/s chip-wide however it is issued (register staging, LDS-DMA with
// 64- or 128-byte row pieces, 1 or 2 workgroups per CU, rotated K order: all within 10 %), and the epilogue traffic adds to it
// instead of hiding under it (they share the memory pipeline).  With MFMA switched off entirely a K = 1536 launch takes 96 %
// of its full time.  A 128 x 128 tile moves (1/128 + 1/128) operand bytes per output element and K step; the only lever
// left is fewer bytes per flop, i.e. larger tiles in BOTH directions:
//
//   * 256 x BN tile, BN = 128 / 256 / 384 chosen so that N is covered with as few column tiles as possible (N = 384 and
//     N = 1536 / 3072: 384; N = 192 / 224: 256 — the 4C-wide or C-wide A panel is then streamed from HBM exactly once);
//     8 wave64 as 4 (M) x 2 (N), 64 x BN/2 per wave = 4 x {4, 8, 12} fragments of v_mfma_f32_16x16x32_bf16 (up to 192
//     accumulator registers);
//   * operands travel HBM / L2 -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging VGPRs (the accumulators need them),
//     no ds_write pass; 3 stages of 32-deep slabs, ONE raw s_barrier per slab and a counted s_waitcnt vmcnt: the next
//     slab's DMA stays in flight across the barrier (a __syncthreads() would drain it);
//   * the LDS image of a DMA is lane-linear (1 KiB = 16 rows x 64 B per wave instruction), so the bank-conflict-free
//     layout is an XOR swizzle applied to the SOURCE address: 16-byte chunk c of row r sits at position
//     c ^ ((r >> 2) & 2) — the 4 x 16 lane groups of ds_read_b128 then touch 16 distinct bank quads (brute-forced over all
//     four hardware lane groups);
//   * the epilogue is WAVE-PRIVATE: each wave parks 16 rows x 64 columns of its accumulators in its own LDS patch, reads
//     them back row-contiguous and streams 16-byte vectors (128-byte row segments) — no block barrier between the K loop
//     and the last store;
//   * column reductions (GRN sum of squares, dz statistics) are butterflied across the 8 row-lanes of a column group and
//     combined over the 4 M-waves through LDS: one atomic per column and tile.
//
// K order and epilogue arithmetic are those of the first-generation kernels: results are bit-identical (tests/test_gpu_ops.py).
// Dispatch (gemm.hip: vsx_gemm_nt): bf16, VSX_A_ROWS on both sides, no operand prologue, K % 32 == 0, M % 256 == 0,
// hw % 64 == 0 where an epilogue is per sample; everything else stays on the first-generation kernels.
#include "vsx_common.h"
#include "../../include/vsx.h"

typedef __attribute__((ext_vector_type(4))) float nt2_f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 nt2_bf16x8;

extern int g_vsx_nt2;

namespace {

constexpr int BM = 256, BK = 32, NST = 3;
constexpr int ROWB = BK * 2;                       // 64 bytes per operand row and slab
constexpr int CW_LD = 68;                          // floats per row of a wave's staging patch (16 rows x 64 columns)
constexpr int CW_BYTES = 16 * CW_LD * 4;
constexpr int RED_OFF = 8 * CW_BYTES;              // column partials of the 8 waves behind the patches

template <int BN>
struct Nt2Geom {
  static constexpr int FN = BN / 32;                                   // 16-column fragments per wave (wave tile 64 x BN/2)
  static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
  static constexpr int NPA = (BM / 16) / 8, NPB = (BN / 16) / 8;       // 1 KiB DMA pieces (16 rows) per wave and slab
  static constexpr int PER_SLAB = NPA + NPB;
  static constexpr int LDS_BYTES = NST * STAGE;
  static constexpr int WAVES_PER_SIMD = BN == 128 ? 4 : 2;             // 72 KB -> two workgroups per CU; 96 / 120 KB -> one
  static_assert(RED_OFF + 8 * 2 * (BN / 2) * 4 <= LDS_BYTES, "epilogue staging overlays the operand stages");
};

// PRO: the GRN prologue a = bf16(g * s[b, k] + beta[k]) of the fc2 forward (VSX_PRO_GRN), applied to the A FRAGMENTS in registers
// between their ds_read and the MFMAs (80 VALU operations per slab and lane next to 48 MFMAs; the first-generation kernel
// does it in its ds_write staging pass, which the DMA path no longer has); s[b, :] and beta live in LDS behind the stages
// (2 K floats).  One sample per tile (dispatch: hw % 256 == 0).  Same arithmetic, same rounding: bit-identical results.
template <int EPI, int BN, bool PRO = false>
__global__ __launch_bounds__(512, Nt2Geom<BN>::WAVES_PER_SIMD) void gemm_nt2_kernel(const VsxGemm p) {
  typedef Nt2Geom<BN> G;
  constexpr int FN = G::FN, A_BYTES = G::A_BYTES, STAGE = G::STAGE, WN = BN / 2;
  constexpr int PRO_FLOATS = PRO ? 2 * 3072 : 4;   // s[b, :K] then beta[:K], K <= 3072
  __shared__ __attribute__((aligned(1024))) char smem[G::LDS_BYTES];
  __shared__ __attribute__((aligned(16))) float gsb[PRO_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int p16 = lane & 15, kq = lane >> 4;

  const int tiles_n = (p.N + BN - 1) / BN;
  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);  // consecutive tiles (same A panel) on one XCD
  const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int b_tile = p.hw > 0 ? m0 / p.hw : 0;

  // ---- DMA source offsets: lane l of a 1 KiB piece fills LDS slot l = (row l >> 2, position l & 3); position pos of row r
  // holds chunk pos ^ swz(r), swz(r) = (r >> 2) & 2, so the lane FETCHES chunk (l & 3) ^ swz(its row)
  const int dr = lane >> 2;
  const int dc = (lane & 3) ^ ((lane >> 4) & 2);
  const char* Ab = reinterpret_cast<const char*>(p.A) + ((size_t)p.a_coff[0] + (size_t)m0 * (size_t)p.lda) * 2;
  const char* Bb = reinterpret_cast<const char*>(p.B) + ((size_t)p.b_off[0] + (size_t)b_tile * (size_t)p.b_bstride) * 2;
  // wave w owns pieces w, w + 8, ... of each operand: rows (w + 8 i) * 16 + dr
  uint32_t offA[G::NPA], offB[G::NPB];
#pragma unroll
  for (int i = 0; i < G::NPA; ++i) offA[i] = (uint32_t)((wave + 8 * i) * 16 + dr) * (uint32_t)(p.lda * 2) + dc * 16;
#pragma unroll
  for (int i = 0; i < G::NPB; ++i) {
    int nb = n0 + (wave + 8 * i) * 16 + dr;
    nb = nb < p.N ? nb : p.N - 1;  // rows past N: clamped (their columns are never stored)
    offB[i] = (uint32_t)nb * (uint32_t)(p.ldb * 2) + dc * 16;
  }

  auto issue = [&](int kt, int st) {
    char* S = smem + st * STAGE;
    const char* Ak = Ab + (size_t)kt * ROWB;
    const char* Bk = Bb + (size_t)kt * ROWB;
#pragma unroll
    for (int i = 0; i < G::NPA; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Ak + offA[i]),
                                       (__attribute__((address_space(3))) void*)(S + (wave + 8 * i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < G::NPB; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Bk + offB[i]),
                                       (__attribute__((address_space(3))) void*)(S + A_BYTES + (wave + 8 * i) * 1024), 16, 0, 0);
  };

  if constexpr (PRO) {
    // s[b, :] and beta travel by LDS-DMA too (64 floats per wave instruction), issued AHEAD of the first operand slab: they are the
    // oldest pieces outstanding, so the first slab's counted wait covers them (round 5).  As ordinary loads + ds_writes they cost
    // every workgroup a cold global round trip — each tile is another sample — before its first operand DMA was even issued.
    // (Round 5 also built the prologue IN LDS one slab ahead of the matrix cores — the A panel a slab ahead of the B panel in a
    // four-slot ring, two 16-byte chunks per thread rescaled in place in 2-VALU micro-steps behind individual MFMAs; an ablation of
    // that kernel priced the whole rescaling at 15 us of the 50 us a prologue launch loses against the plain launch at C = 384,
    // B = 512 — the rest stayed with BOTH forms when the arithmetic was compiled out — and the hand-fenced loop ran no faster than
    // this fragment form: 232 vs 236 us.  Not kept)
    const float* gs = p.grn_s + (size_t)b_tile * p.K;
    for (int c = wave; c * 64 < p.K; c += 8) {
      int i = c * 64 + lane;
      i = i < p.K ? i : p.K - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gs + i),
                                       (__attribute__((address_space(3))) void*)(gsb + c * 64), 4, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.grn_b + i),
                                       (__attribute__((address_space(3))) void*)(gsb + 3072 + c * 64), 4, 0, 0);
    }
  }

  nt2_f32x4 acc[4][FN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (nt2_f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment addresses: lane (p16, kq) reads chunk kq of row (.. + p16) at position kq ^ swz(p16)
  const int fpos = (kq ^ ((p16 >> 2) & 2)) * 16;
  const int fragA = (wm * 64 + p16) * ROWB + fpos;
  const int fragB = A_BYTES + (wn * WN + p16) * ROWB + fpos;
  auto compute = [&](int st, int kt) {
    const char* S = smem + st * STAGE;
    nt2_bf16x8 af[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const nt2_bf16x8*>(S + fragA + i * 16 * ROWB);
    if constexpr (PRO) {
      // lane (p16, kq) holds k = kt * 32 + kq * 8 .. + 7 of its rows
      const float* sp = gsb + kt * BK + kq * 8;
      const float4 s0 = *reinterpret_cast<const float4*>(sp), s1 = *reinterpret_cast<const float4*>(sp + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(sp + 3072), b1 = *reinterpret_cast<const float4*>(sp + 3076);
      const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float f[8];
        unpack<bf16_t>(__builtin_bit_cast(uint4, af[i]), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], sv[e], bv[e]);
        af[i] = __builtin_bit_cast(nt2_bf16x8, pack<bf16_t>(f));
      }
    }
#pragma unroll
    for (int jg = 0; jg < FN; jg += 4) {  // B fragments four at a time (register budget of the 12-fragment geometry)
      nt2_bf16x8 bf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const nt2_bf16x8*>(S + fragB + (jg + j) * 16 * ROWB);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][jg + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][jg + j], 0, 0, 0);
    }
  };

  const int nk = p.K / BK;
  // Round 5 (VSX_EPI_BIAS_RES): the shortcut rows the epilogue adds are requested RES_D passes (16 rows x 64 columns each) ahead of
  // their use — the first RES_D while the last slabs are still being multiplied (BN <= 256; at BN = 384 the accumulators leave no
  // registers for that and the prefetch starts with the epilogue) — instead of inside each pass, where every one of the FN passes
  // of a wave began with a cold global round trip (the fc1 data gradient's LayerNorm epilogue lost 19 % to the same pattern).
  constexpr bool RES = EPI == VSX_EPI_BIAS_RES;
  constexpr int RES_D = RES ? (BN == 384 ? 2 : (BN == 256 ? 4 : 1)) : 1;   // (BN = 128 runs at a 128-register cap: one pass ahead only)
  constexpr bool RES_INLOOP = RES && BN == 256;
  uint4 rq[RES_D][2];
  float rsv[RES ? 4 : 1][2];
  const int e_r = lane >> 3, e_c = (lane & 7) * 8;
  auto res_fetch = [&](int q, int slot) {   // pass q = cg * 4 + i
    if constexpr (RES) {
      const int cgq = q >> 2, iq = q & 3;
      const int nq = n0 + wn * WN + cgq * 64 + e_c;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int m = m0 + wm * 64 + iq * 16 + e_r + 8 * h;
        rq[slot][h] = (nq < p.N && m < p.M) ? ldvec<bf16_t>(reinterpret_cast<const bf16_t*>(p.res) + (size_t)m * p.ldr + nq) : make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  auto res_prologue = [&]() {
    if constexpr (RES) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int m = m0 + wm * 64 + i * 16 + e_r + 8 * h;
          rsv[i][h] = (p.rscale && m < p.M) ? p.rscale[p.hw > 0 ? m / p.hw : 0] : 1.f;
        }
#pragma unroll
      for (int q = 0; q < RES_D && q < FN; ++q) res_fetch(q, q);
    }
  };
  issue(0, 0);
  if (nk > 1) issue(1, 1);
  if (RES_INLOOP && nk < 3) res_prologue();
  int st = 0, stn = NST - 1;
  for (int kt = 0; kt < nk; ++kt) {
    // slab kt has landed for this wave (its pieces are the oldest outstanding); slab kt + 1 may still be in flight
    if (kt + 1 < nk) {
      if constexpr (G::PER_SLAB == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if constexpr (G::PER_SLAB == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // ... for every wave; and every wave is done reading the stage that is refilled next
    if (kt + 2 < nk) issue(kt + 2, stn);
    if (RES_INLOOP && kt == nk - 3) res_prologue();
    compute(st, kt);
    st = st == NST - 1 ? 0 : st + 1;
    stn = stn == NST - 1 ? 0 : stn + 1;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // the operand stages are dead: the staging patches overlay them
  if (RES && !RES_INLOOP) res_prologue();

  // ---- epilogue, wave-private: (BN / 128) column groups x 4 row passes of 16 rows x 64 columns
  float* Cw = reinterpret_cast<float*>(smem + wave * CW_BYTES);
  const int er = lane >> 3, ec = (lane & 7) * 8;   // this lane's rows er, er + 8 and columns ec .. ec + 7 of a patch
  constexpr bool REDUCE = (EPI == VSX_EPI_BIAS_GELU_SQ || EPI == VSX_EPI_DZ);
  float* red = reinterpret_cast<float*>(smem + RED_OFF);  // [8 waves][2][WN]

#pragma unroll
  for (int cg = 0; cg < FN / 4; ++cg) {
    const int n = n0 + wn * WN + cg * 64 + ec;
    const bool ncol_ok = n < p.N;
    const size_t ccol = (size_t)p.c_coff[0] + n;
    float bias[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bias[j] = 0.f;
    if constexpr (EPI != VSX_EPI_NONE && EPI != VSX_EPI_DZ) {
      if (ncol_ok && p.bias != nullptr) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
        bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w;
        bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
      }
    }
    float r0[8], r1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { r0[j] = 0.f; r1[j] = 0.f; }

#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cw[(kq * 4 + r) * CW_LD + j * 16 + p16] = acc[i][cg * 4 + j][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float v[2][8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float4 t0 = *reinterpret_cast<const float4*>(Cw + (er + 8 * h) * CW_LD + ec);
        const float4 t1 = *reinterpret_cast<const float4*>(Cw + (er + 8 * h) * CW_LD + ec + 4);
        v[h][0] = t0.x + bias[0]; v[h][1] = t0.y + bias[1]; v[h][2] = t0.z + bias[2]; v[h][3] = t0.w + bias[3];
        v[h][4] = t1.x + bias[4]; v[h][5] = t1.y + bias[5]; v[h][6] = t1.z + bias[6]; v[h][7] = t1.w + bias[7];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (ncol_ok) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int m = m0 + wm * 64 + i * 16 + er + 8 * h;
          if (m < p.M) {
            if constexpr (EPI == VSX_EPI_BIAS_RES) {
              float rf[8];
              unpack<bf16_t>(rq[(cg * 4 + i) % RES_D][h], rf);
              const float rs = rsv[i][h];
#pragma unroll
              for (int j = 0; j < 8; ++j) v[h][j] = fmaf(v[h][j], rs, rf[j]);
            } else if constexpr (EPI == VSX_EPI_BIAS_GELU_SQ) {
              float gv[8];
#pragma unroll
              for (int j = 0; j < 8; j += 2) {
                const vsx_v2f x = {round_bf16(v[h][j]), round_bf16(v[h][j + 1])};
                vsx_v2f cdf, pdf;
                gelu_parts2(x, cdf, pdf);
                const vsx_v2f gg = x * cdf;
                gv[j] = round_bf16(gg.x);
                gv[j + 1] = round_bf16(gg.y);
                r0[j] += gv[j] * gv[j];
                r0[j + 1] += gv[j + 1] * gv[j + 1];
              }
              stvec<bf16_t>(reinterpret_cast<bf16_t*>(p.C2) + (size_t)m * p.ldc + ccol, pack<bf16_t>(gv));
            } else if constexpr (EPI == VSX_EPI_DZ) {
              float gf[8];
              unpack<bf16_t>(ldvec<bf16_t>(reinterpret_cast<const bf16_t*>(p.aux) + (size_t)m * p.ldx + n), gf);
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const float dz = round_bf16(v[h][j]);
                r0[j] += dz * gf[j];
                r1[j] += dz;
              }
            }
            if (EPI != VSX_EPI_BIAS_GELU_SQ || p.C != nullptr)
              stvec<bf16_t>(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + ccol, pack<bf16_t>(v[h]));
          }
        }
      }
      if constexpr (RES) {  // this pass's slot is free: request the rows of pass q + RES_D
        if (cg * 4 + i + RES_D < FN) res_fetch(cg * 4 + i + RES_D, (cg * 4 + i) % RES_D);
      }
    }

    if constexpr (REDUCE) {
      // sum over the 8 row-lanes of a column group (lane bits 3..5)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        r0[j] += __shfl_xor(r0[j], 8, 64);
        r0[j] += __shfl_xor(r0[j], 16, 64);
        r0[j] += __shfl_xor(r0[j], 32, 64);
        if constexpr (EPI == VSX_EPI_DZ) {
          r1[j] += __shfl_xor(r1[j], 8, 64);
          r1[j] += __shfl_xor(r1[j], 16, 64);
          r1[j] += __shfl_xor(r1[j], 32, 64);
        }
      }
      if (p.hw % BM == 0) {
        // the whole tile lies in one sample: park, the 4 M-waves are combined below
        if (lane < 8) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            red[(wave * 2 + 0) * WN + cg * 64 + lane * 8 + j] = r0[j];
            if constexpr (EPI == VSX_EPI_DZ) red[(wave * 2 + 1) * WN + cg * 64 + lane * 8 + j] = r1[j];
          }
        }
      } else if (lane < 8 && ncol_ok) {
        // hw = 64 / 128: this wave's 64 rows lie in one sample (dispatch: hw % 64 == 0)
        const size_t bs = (size_t)((m0 + wm * 64) / p.hw);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          atomicAdd(p.red0 + bs * p.N + n + j, r0[j]);
          if constexpr (EPI == VSX_EPI_DZ) atomicAdd(p.red1 + bs * p.N + n + j, r1[j]);
        }
      }
    }
  }

  if constexpr (REDUCE) {
    if (p.hw % BM == 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (tid < BN && n0 + tid < p.N) {
        const int w0 = tid / WN, c = tid % WN;  // N-half, column inside it
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          a0 += red[((w * 2 + w0) * 2 + 0) * WN + c];
          if constexpr (EPI == VSX_EPI_DZ) a1 += red[((w * 2 + w0) * 2 + 1) * WN + c];
        }
        if (EPI == VSX_EPI_BIAS_GELU_SQ && p.aux != nullptr) {
          // det_reduce (the dispatcher put the workspace into the otherwise unused `aux`): one row of column sums per row tile,
          // added up per sample in tile order by vsx_det_group_sum
          reinterpret_cast<float*>(const_cast<void*>(p.aux))[(size_t)tile_m * p.N + n0 + tid] = a0;
        } else {
          atomicAdd(p.red0 + (size_t)b_tile * p.N + n0 + tid, a0);
          if constexpr (EPI == VSX_EPI_DZ) atomicAdd(p.red1 + (size_t)b_tile * p.N + n0 + tid, a1);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// fc1 data gradient with the block LayerNorm's backward in its epilogue (VSX_EPI_LN_BWD):
//     dx^ = dh . W1'            (the GEMM: M x C, K = 4C)
//     dy  = rstd * (dx^ - mean_c(dx^) - x^ * mean_c(dx^ * x^))          (LayerNorm without affine: it is folded into fc1)
// One column tile spans the whole row (C <= 256), and the waves are laid out 8 (M) x 1 (N) — a wave owns 32 complete rows, so
// the two row means are sums over the lanes of ONE wave.  dx^ is rounded to bf16 where the unfused pair stored it; it is never
// written: the launch that wrote it, the LayerNorm-backward launch that read it back with x^, and 2 C-wide passes per block go.
// ------------------------------------------------------------------------------------------------
// NT2_TS (probe builds only: -DNT2_TS=1): s_memtime stamps of ONE workgroup's waves along a tile of the fused data-gradient /
// LayerNorm kernel, read back with vsx_debug_nt2_ts (tools/nt2_timeline.py): entry, first slab landed, K loop done (+ per-slab
// stamps), first 16-row pass done, exit.
#ifdef NT2_TS
__device__ unsigned long long g_nt2_ts[8 * 80];
#define NT2_STAMP(k) do { if (blockIdx.x == gridDim.x / 2 + 3 && lane == 0) g_nt2_ts[wave * 80 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define NT2_STAMP(k) do { } while (0)
#endif

template <int BN>
__global__ __launch_bounds__(512, 2) void gemm_nt2_lnbwd_kernel(const VsxGemm p) {
  typedef Nt2Geom<BN> G;
  constexpr int FN = BN / 16, A_BYTES = G::A_BYTES, STAGE = G::STAGE, NCG = BN / 64;
  __shared__ __attribute__((aligned(1024))) char smem[G::LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p16 = lane & 15, kq = lane >> 4;
  int bid = blockIdx.x;
  const int m0 = bid * BM;
  NT2_STAMP(0);

  const int dr = lane >> 2;
  const int dc = (lane & 3) ^ ((lane >> 4) & 2);
  const char* Ab = reinterpret_cast<const char*>(p.A) + (size_t)m0 * (size_t)p.lda * 2;
  const char* Bb = reinterpret_cast<const char*>(p.B);
  uint32_t offA[G::NPA], offB[G::NPB];
#pragma unroll
  for (int i = 0; i < G::NPA; ++i) offA[i] = (uint32_t)((wave + 8 * i) * 16 + dr) * (uint32_t)(p.lda * 2) + dc * 16;
#pragma unroll
  for (int i = 0; i < G::NPB; ++i) {
    int nb = (wave + 8 * i) * 16 + dr;
    nb = nb < p.N ? nb : p.N - 1;
    offB[i] = (uint32_t)nb * (uint32_t)(p.ldb * 2) + dc * 16;
  }
  auto issue = [&](int kt, int st) {
    char* S = smem + st * STAGE;
    const char* Ak = Ab + (size_t)kt * ROWB;
    const char* Bk = Bb + (size_t)kt * ROWB;
#pragma unroll
    for (int i = 0; i < G::NPA; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Ak + offA[i]),
                                       (__attribute__((address_space(3))) void*)(S + (wave + 8 * i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < G::NPB; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Bk + offB[i]),
                                       (__attribute__((address_space(3))) void*)(S + A_BYTES + (wave + 8 * i) * 1024), 16, 0, 0);
  };

  nt2_f32x4 acc[2][FN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (nt2_f32x4){0.f, 0.f, 0.f, 0.f};
  const int fpos = (kq ^ ((p16 >> 2) & 2)) * 16;
  const int fragA = (wave * 32 + p16) * ROWB + fpos;
  const int fragB = A_BYTES + p16 * ROWB + fpos;
  auto compute = [&](int st) {
    const char* S = smem + st * STAGE;
    nt2_bf16x8 af[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const nt2_bf16x8*>(S + fragA + i * 16 * ROWB);
#pragma unroll
    for (int jg = 0; jg < FN; jg += 4) {
      nt2_bf16x8 bf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const nt2_bf16x8*>(S + fragB + (jg + j) * 16 * ROWB);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][jg + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][jg + j], 0, 0, 0);
    }
  };

  const int nk = p.K / BK;
  // Round 5: the epilogue's row operands — this lane's 16-byte pieces of y / x^ (and the row means / rstd) — are requested while
  // the last slabs are still being multiplied (behind the LAST operand DMA, at slab nk - 3), not inside the epilogue: there every
  // 16-row pass began with a cold global round trip that nothing covered (one workgroup per CU).  The counted waits of the last
  // two slabs then see these loads as the youngest outstanding operations: they make slab nk - 1 land one step early, no more.
  constexpr int er_ = 0;
  const int erow = lane >> 3, ecol = (lane & 7) * 8;
  const bf16_t* XHp = reinterpret_cast<const bf16_t*>(p.aux);
  // (the first 16-row pass's operands from inside the K loop, the second pass's at the top of the first: both sets live across the
  // K loop would spill 39 registers at BN = 256)
  uint4 xpre[2][NCG][2];
  float mupre[2][2], rspre[2][2];
  auto prefetch_pass = [&](int i) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int m = m0 + wave * 32 + i * 16 + erow + 8 * h;
      mupre[i][h] = p.grn_b ? p.grn_b[m] : 0.f;
      rspre[i][h] = p.grn_s[m];
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg) {
        const int n = cg * 64 + ecol;
        xpre[i][cg][h] = n < p.N ? ldvec<bf16_t>(XHp + (size_t)m * p.ldx + n) : make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  auto prefetch_rows = [&]() { prefetch_pass(0); };
  (void)er_;
  issue(0, 0);
  if (nk > 1) issue(1, 1);
  if (nk < 3) prefetch_rows();
  int st = 0, stn = NST - 1;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      if constexpr (G::PER_SLAB == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (kt < 64) NT2_STAMP(8 + kt);
    if (kt + 2 < nk) issue(kt + 2, stn);
    if (kt == nk - 3) prefetch_rows();
    compute(st);
    st = st == NST - 1 ? 0 : st + 1;
    stn = stn == NST - 1 ? 0 : stn + 1;
  }
  NT2_STAMP(1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  NT2_STAMP(2);

  // ---- epilogue: per 16-row fragment, all column groups are read back into registers (dx^ as bf16 values, x^), the row
  // means are reduced over the 8 lanes that share a row, then dy is formed and stored
  float* Cw = reinterpret_cast<float*>(smem + wave * CW_BYTES);
  const int er = lane >> 3, ec = (lane & 7) * 8;
  const float invC = 1.f / (float)p.N;
  const bf16_t* XH = reinterpret_cast<const bf16_t*>(p.aux);
  bf16_t* DY = reinterpret_cast<bf16_t*>(p.C);
  // grn_b = the row means: `aux` then holds the UN-normalised rows y (x^ was never stored; it is re-formed here with the forward's
  // expression), and the A operand is dh * rstd, so the accumulator already is rstd * dx^ (csrc/mlp.hip MODE 7)
  const float* MEAN = p.grn_b;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float v[NCG][2][8];
    uint4 (&xq)[NCG][2] = xpre[i];
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) {
      const int n = cg * 64 + ec;
      const bool ok = n < p.N;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cw[(kq * 4 + r) * CW_LD + j * 16 + p16] = acc[i][cg * 4 + j][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float4 t0 = *reinterpret_cast<const float4*>(Cw + (er + 8 * h) * CW_LD + ec);
        const float4 t1 = *reinterpret_cast<const float4*>(Cw + (er + 8 * h) * CW_LD + ec + 4);
        const float t[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        float xf[8];
        unpack<bf16_t>(xq[cg][h], xf);
        if (MEAN && ok) {
          const float mu = mupre[i][h], rsd = rspre[i][h];
#pragma unroll
          for (int e = 0; e < 8; ++e) xf[e] = (xf[e] - mu) * rsd;
          xq[cg][h] = pack<bf16_t>(xf);
          unpack<bf16_t>(xq[cg][h], xf);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = ok ? round_bf16(t[e]) : 0.f;   // dx^ as the unfused pair stored it
          v[cg][h][e] = d;
          s1[h] += d;
          s2[h] = fmaf(d, xf[e], s2[h]);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    NT2_STAMP(3 + i);
    if (i == 0) prefetch_pass(1);  // (behind the accumulator hand-over of this pass: its staging registers are free)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // the 8 lanes er * 8 .. + 7 hold the row's column groups
      s1[h] += __shfl_xor(s1[h], 1, 64); s1[h] += __shfl_xor(s1[h], 2, 64); s1[h] += __shfl_xor(s1[h], 4, 64);
      s2[h] += __shfl_xor(s2[h], 1, 64); s2[h] += __shfl_xor(s2[h], 2, 64); s2[h] += __shfl_xor(s2[h], 4, 64);
      const int m = m0 + wave * 32 + i * 16 + er + 8 * h;
      const float rs = MEAN ? 1.f : rspre[i][h];
      const float m1 = s1[h] * invC, m2 = s2[h] * invC;
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg) {
        const int n = cg * 64 + ec;
        if (n < p.N) {
          float xf[8], o[8];
          unpack<bf16_t>(xq[cg][h], xf);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = rs * (v[cg][h][e] - m1 - xf[e] * m2);
          stvec<bf16_t>(DY + (size_t)m * p.ldc + n, pack<bf16_t>(o));
        }
      }
    }
  }
  NT2_STAMP(5);
}

template <int EPI, int BN>
int launch_bn(const VsxGemm* p, hipStream_t s) {
  const int tiles = (p->M / BM) * vsx_cdiv(p->N, BN);
  if constexpr (EPI == VSX_EPI_BIAS_RES || EPI == VSX_EPI_NONE) {
    if (p->pro == VSX_PRO_GRN) {
      hipLaunchKernelGGL((gemm_nt2_kernel<EPI, BN, true>), dim3(tiles), dim3(512), 0, s, *p);
      VSX_LAUNCH_CHECK();
      return 0;
    }
  }
  hipLaunchKernelGGL((gemm_nt2_kernel<EPI, BN>), dim3(tiles), dim3(512), 0, s, *p);
  VSX_LAUNCH_CHECK();
  return 0;
}

// column-tile width: the one that moves the fewest operand bytes per 256 rows and K step — tiles x (256 + BN) — among
// those that idle at most a third of their MFMA work on columns past N
int pick_bn(int N) {
  if (g_vsx_nt2 & 4) return 128;                       // A/B knob: the 256 x 128 geometry everywhere
  int best = 128, best_cost = 1 << 30;
  for (int bn = 128; bn <= ((g_vsx_nt2 & 8) ? 256 : 384); bn += 128) {  // (bit 3: A/B knob, no 384-wide tiles)
    const int tiles = (N + bn - 1) / bn;
    if (3 * (tiles * bn - N) > N && bn != 128) continue;
    const int cost = tiles * (BM + bn);
    if (cost < best_cost) { best_cost = cost; best = bn; }
  }
  return best;
}

template <int EPI>
int launch(const VsxGemm* p, hipStream_t s) {
  switch (pick_bn(p->N)) {
    case 384: return launch_bn<EPI, 384>(p, s);
    case 256: return launch_bn<EPI, 256>(p, s);
    default: return launch_bn<EPI, 128>(p, s);
  }
}

}  // namespace

// true if this kernel family takes the launch (bf16 only; the caller has validated the common fields)
// the fused fc1-data-gradient + LayerNorm-backward launch exists on this kernel family only
bool vsx_gemm_nt2_lnbwd_ok(const VsxGemm* p) {
  return p->a_mode == VSX_A_ROWS && p->c_mode == VSX_A_ROWS && p->pro == VSX_PRO_NONE && p->nz <= 1 && p->K % BK == 0 &&
         p->M % BM == 0 && p->N >= 64 && p->N <= 256 && p->N % 8 == 0 && p->b_bstride == 0 &&
         (unsigned long long)BM * p->lda * 2 < (1ull << 32) && (unsigned long long)p->N * p->ldb * 2 < (1ull << 32);
}

bool vsx_gemm_nt2_ok(const VsxGemm* p) {
  if (p->epi == VSX_EPI_LN_BWD) return vsx_gemm_nt2_lnbwd_ok(p);
  if (!(g_vsx_nt2 & 1)) return false;
  if (p->a_mode != VSX_A_ROWS || p->c_mode != VSX_A_ROWS || p->nz > 1) return false;
  if (p->pro != VSX_PRO_NONE) {  // GRN prologue: one sample per tile, epilogues of the fc2 forward only, s / beta fit the LDS copy
    if (p->pro != VSX_PRO_GRN || p->hw <= 0 || p->hw % BM != 0 || p->K > 3072) return false;
    if (p->epi != VSX_EPI_BIAS_RES && p->epi != VSX_EPI_NONE) return false;
  }
  if (p->K % BK != 0 || p->M % BM != 0 || p->N < 64 || p->N % 8 != 0) return false;
  if (p->epi < VSX_EPI_NONE || p->epi > VSX_EPI_DZ) return false;
  const bool per_sample = p->epi == VSX_EPI_BIAS_GELU_SQ || p->epi == VSX_EPI_DZ || p->rscale != nullptr;
  if (per_sample && (p->hw <= 0 || p->hw % 64 != 0)) return false;
  if (p->b_bstride != 0 && (p->hw <= 0 || p->hw % BM != 0)) return false;
  if (p->hw > 0 && p->hw % BM != 0 && p->hw != 64 && p->hw != 128) return false;
  if ((unsigned long long)BM * p->lda * 2 >= (1ull << 32) || (unsigned long long)p->N * p->ldb * 2 >= (1ull << 32)) return false;
  if (g_vsx_nt2 & 2) return true;  // (tests / A-B runs: every supported launch)
  // det_reduce: this kernel is the one whose GELU / sum-of-squares epilogue has the fixed-order path (the first-generation kernels
  // reduce with atomics) — it takes every such launch it supports, also the small ones the heuristics below leave to them
  if (g_vsx_det_reduce && p->epi == VSX_EPI_BIAS_GELU_SQ) return true;
  // Where the wide tiles pay (tools/perf_nt_gen2.py, B = 512, against the first-generation kernel): the K-heavy launches
  // whose output is C-wide — fc2 / fc1 data gradient of the C = 384 / 768 stages -10 .. -25 %, of the 224-channel decoder
  // stage -4 .. -10 % — and the C = 384 fc1 (-14 %).  Not the dz epilogue (its second operand stream eats the gain), not
  // K <= 384 with a narrow N (three slabs per tile at one workgroup per CU: +5 .. +25 %).
  const int bn = pick_bn(p->N);
  if ((long)(p->M / BM) * vsx_cdiv(p->N, bn) < 256) return false;  // few tiles: the BK = 128 generic path
  if (p->epi == VSX_EPI_DZ) return false;
  if (p->pro == VSX_PRO_GRN) return p->K >= 768 && p->N >= 192;  // the fc2 forward of the 16 x 16 maps (C = 384)
  if (p->epi == VSX_EPI_BIAS_GELU_SQ) return bn == 384 && p->N % 384 == 0 && p->K >= 384 && p->K <= 768;
  return p->K >= 768 && (p->N > 192 || ((g_vsx_nt2 & 16) && p->N == 192));  // bit 4 (round 6, A/B): the C = 192 fc2 forward too
}

int vsx_gemm_nt2(const VsxGemm* p0, hipStream_t s) {
  g_vsx_last_kernel = "gemm_nt2";
  VsxGemm det = *p0;
  const VsxGemm* p = p0;
  const bool det_sums = g_vsx_det_reduce && p0->epi == VSX_EPI_BIAS_GELU_SQ && p0->hw > 0 && p0->hw % BM == 0 && p0->hw > BM;
  if (det_sums) {  // (hw <= 256: at most two adds per address — already independent of their order)
    const long need = (long)(p0->M / BM) * p0->N;
    VSX_CHECK(g_vsx_det_ws != nullptr && g_vsx_det_ws_floats >= need, "vsx_gemm_nt: det_reduce needs vsx_det_workspace(>= %ld floats)", need);
    det.aux = g_vsx_det_ws;
    p = &det;
  }
  if (det_sums) {
    int e = launch<VSX_EPI_BIAS_GELU_SQ>(p, s);
    if (e) return e;
    return vsx_det_group_sum(g_vsx_det_ws, p->N, 0, p->red0, p->M / p->hw, p->hw / BM, p->N, s);
  }
  if (p->epi == VSX_EPI_LN_BWD) {
    const int tiles = p->M / BM;
    if (p->N <= 128) hipLaunchKernelGGL((gemm_nt2_lnbwd_kernel<128>), dim3(tiles), dim3(512), 0, s, *p);
    else hipLaunchKernelGGL((gemm_nt2_lnbwd_kernel<256>), dim3(tiles), dim3(512), 0, s, *p);
    VSX_LAUNCH_CHECK();
    return 0;
  }
  switch (p->epi) {
    case VSX_EPI_NONE: return launch<VSX_EPI_NONE>(p, s);
    case VSX_EPI_BIAS: return launch<VSX_EPI_BIAS>(p, s);
    case VSX_EPI_BIAS_GELU_SQ: return launch<VSX_EPI_BIAS_GELU_SQ>(p, s);
    case VSX_EPI_BIAS_RES: return launch<VSX_EPI_BIAS_RES>(p, s);
    default: return launch<VSX_EPI_DZ>(p, s);
  }
}

#ifdef NT2_TS
extern "C" int32_t vsx_debug_nt2_ts(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_nt2_ts), sizeof(g_nt2_ts)) == hipSuccess ? 0 : 1;
}
#endif
