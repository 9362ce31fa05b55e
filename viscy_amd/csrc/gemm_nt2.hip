// gemm_nt, second generation, for the launches that carry the step (bf16, plain row operands):
//
//     C[M, N] = A[M, K] . B[N, K]^T  (+ the fused epilogues of vsx_gemm_nt)
//
// What the first lean kernel (gemm.hip: 128 x 128 tile, global -> VGPR -> LDS staging, block-wide staged epilogue) left on
// the table, measured at B = 512: the wide-output launches (fc1 of the C = 384 / 768 blocks: two 4C-wide outputs, and
// their dz) run at 2.4 - 2.9 TB/s and 0.45 - 0.49 PFLOP/s — neither pipe is busy; every tile walks
// load -> wait -> ds_write -> barrier -> MFMA -> barrier and then a store phase in which all four waves of the workgroup
// move in lock step through two block-wide LDS passes.
//
// gfx950 mapping of this kernel
//   * 256 x 128 tile, 8 wave64 as 4 (M) x 2 (N), 64 x 64 per wave (4 x 4 fragments of v_mfma_f32_16x16x32_bf16);
//   * operands travel HBM / L2 -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass (a
//     ds_write_b128 costs 13 LDS-issue cycles per wave on this part), 3 stages of 32-deep slabs (72 KB -> two workgroups
//     per CU), ONE raw s_barrier per slab and a counted s_waitcnt vmcnt(3): the next slab's DMA stays in flight across
//     the barrier (a __syncthreads() would drain it);
//   * the LDS image of a DMA is lane-linear (1 KiB = 16 rows x 64 B per wave instruction), so the bank-conflict-free
//     layout is an XOR swizzle applied to the SOURCE address: 16-byte chunk c of row r sits at position
//     c ^ ((r >> 2) & 2) — the 4 x 16 lane groups of ds_read_b128 then touch 16 distinct bank quads (brute-forced over all
//     four hardware lane groups);
//   * the epilogue is WAVE-PRIVATE: each wave parks 16 rows x 64 columns of its accumulators in its own LDS patch, reads
//     them back row-contiguous and streams 16-byte vectors (128-byte row segments) — no block barrier between the K loop
//     and the last store, so the waves of a workgroup drift apart, the store stream of one tile runs under the MFMA work of
//     the other workgroup on the CU, and a wave retires as soon as its own stores are issued;
//   * column reductions (GRN sum of squares, dz statistics) are butterflied across the 8 row-lanes of a column group and
//     combined over the 4 M-waves through LDS: 128 atomics per tile.
//
// Dispatch (gemm.hip: vsx_gemm_nt): bf16, VSX_A_ROWS on both sides, no operand prologue, K % 32 == 0, M % 256 == 0,
// hw % 64 == 0 where an epilogue is per sample; everything else stays on the first-generation kernels.
#include "vsx_common.h"
#include "../../include/vsx.h"

typedef __attribute__((ext_vector_type(4))) float nt2_f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 nt2_bf16x8;

extern int g_vsx_nt2;

namespace {

constexpr int BM = 256, BN = 128;
constexpr int CW_LD = 68;                          // floats per row of a wave's staging patch (16 rows x 64 columns)
constexpr int CW_BYTES = 16 * CW_LD * 4;
constexpr int RED_OFF = 8 * CW_BYTES;              // [8 waves][2][64] floats of column partials behind the patches

// BK = 32: 64-byte row pieces, 3 stages of 24 KB (two workgroups per CU); BK = 64: 128-byte row pieces (whole cache lines),
// 2 stages of 48 KB (one workgroup per CU)
template <int BK, int NST>
struct Nt2Geom {
  static constexpr int ROWB = BK * 2;                        // bytes per operand row and slab
  static constexpr int RPP = 1024 / ROWB;                    // rows per 1 KiB DMA piece
  static constexpr int CPR = ROWB / 16;                      // 16-byte chunks per row
  static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
  static constexpr int NPA = (BM / RPP) / 8, NPB = (BN / RPP) / 8;  // DMA pieces per wave and slab
  static constexpr int LDS_BYTES = NST * STAGE;
  static_assert(RED_OFF + 8 * 2 * 64 * 4 <= LDS_BYTES, "epilogue staging overlays the operand stages");
};

template <int EPI, int BK, int NST>
__global__ __launch_bounds__(512, BK == 32 ? 4 : 2) void gemm_nt2_kernel(const VsxGemm p) {
  typedef Nt2Geom<BK, NST> G;
  constexpr int ROWB = G::ROWB, A_BYTES = G::A_BYTES, STAGE = G::STAGE, LDS_BYTES = G::LDS_BYTES;
  __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];
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

  // ---- DMA source offsets: lane l of a 1 KiB piece fills LDS slot l = (row l / CPR, position l % CPR); position pos of row r
  // holds chunk pos ^ swz(r), so the lane FETCHES chunk (l % CPR) ^ swz(its row).  swz: BK = 32 -> (r >> 2) & 2,
  // BK = 64 -> (r >> 1) & 7 (both brute-forced conflict-free over the four 16-lane groups of ds_read_b128)
  const int dr = lane / G::CPR;
  const int dc = BK == 32 ? ((lane & 3) ^ ((lane >> 4) & 2)) : ((lane & 7) ^ ((((wave & 1) * 8 + dr) >> 1) & 7));
  const char* Ab = reinterpret_cast<const char*>(p.A) + ((size_t)p.a_coff[0] + (size_t)m0 * (size_t)p.lda) * 2;
  const char* Bb = reinterpret_cast<const char*>(p.B) + ((size_t)p.b_off[0] + (size_t)b_tile * (size_t)p.b_bstride) * 2;
  // wave w owns pieces w, w + 8, ... of each operand: rows (w + 8 i) * RPP + dr
  uint32_t offA[G::NPA], offB[G::NPB];
#pragma unroll
  for (int i = 0; i < G::NPA; ++i) offA[i] = (uint32_t)((wave + 8 * i) * G::RPP + dr) * (uint32_t)(p.lda * 2) + dc * 16;
#pragma unroll
  for (int i = 0; i < G::NPB; ++i) {
    int nb = n0 + (wave + 8 * i) * G::RPP + dr;
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

  nt2_f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (nt2_f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment addresses: lane (p16, kq) reads chunk (kk * 4 + kq) of row (.. + p16) at its swizzled position
  const int fswz = BK == 32 ? ((p16 >> 2) & 2) : ((p16 >> 1) & 7);
  const int fragA = (wm * 64 + p16) * ROWB;
  const int fragB = A_BYTES + (wn * 64 + p16) * ROWB;
  auto compute = [&](int st) {
    const char* S = smem + st * STAGE;
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      const int fpos = ((kk * 4 + kq) ^ fswz) * 16;
      nt2_bf16x8 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const nt2_bf16x8*>(S + fragA + fpos + i * 16 * ROWB);
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const nt2_bf16x8*>(S + fragB + fpos + j * 16 * ROWB);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  };

  // K order: slab (kt + rot) % nk.  rot = 0 keeps the accumulation order of the first-generation kernels (bit-identical
  // results); probe bit 32 rotates the start per M tile so that the tiles of one N column do not all ask the L2 for the same
  // weight slab at the same time
  const int nk = p.K / BK;
  const int rot = (p.pro & 8192) ? (tile_m % nk) : 0;
  auto slab = [&](int kt) { const int k = kt + rot; return k >= nk ? k - nk : k; };
  constexpr int PER_SLAB = G::NPA + G::NPB;  // DMA instructions per wave and slab
#pragma unroll
  for (int i = 0; i < NST - 1; ++i)
    if (i < nk) issue(slab(i), i);
  int st = 0, stn = NST - 1;
  for (int kt = 0; kt < nk; ++kt) {
    // slab kt has landed for this wave (its pieces are the oldest outstanding); up to NST - 2 later slabs stay in flight
    const int ahead = nk - 1 - kt < NST - 2 ? nk - 1 - kt : NST - 2;
    if ((p.pro & 2048) || ahead == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (NST == 3) {
      if constexpr (PER_SLAB == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // ... for every wave; and every wave is done reading the stage that is refilled next
    if (kt + NST - 1 < nk && !(p.pro & 2048)) issue(slab(kt + NST - 1), stn);
    if (!(p.pro & 1024)) compute(st);
    st = st == NST - 1 ? 0 : st + 1;
    stn = stn == NST - 1 ? 0 : stn + 1;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // the operand stages are dead: the staging patches overlay them

  // ---- epilogue, wave-private: 4 passes of 16 rows x 64 columns
  float* Cw = reinterpret_cast<float*>(smem + wave * CW_BYTES);
  const int er = lane >> 3, ec = (lane & 7) * 8;   // this lane's rows er, er + 8 and columns ec .. ec + 7 of a patch
  const int n = n0 + wn * 64 + ec;
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
  constexpr bool REDUCE = (EPI == VSX_EPI_BIAS_GELU_SQ || EPI == VSX_EPI_DZ);
  float r0[8], r1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { r0[j] = 0.f; r1[j] = 0.f; }

#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) Cw[(kq * 4 + r) * CW_LD + j * 16 + p16] = acc[i][j][r];
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
    if (ncol_ok && !(p.pro & 4096)) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int m = m0 + wm * 64 + i * 16 + er + 8 * h;
        if (m < p.M) {
          if constexpr (EPI == VSX_EPI_BIAS_RES) {
            float rf[8];
            unpack<bf16_t>(ldvec<bf16_t>(reinterpret_cast<const bf16_t*>(p.res) + (size_t)m * p.ldr + n), rf);
            const float rs = p.rscale ? p.rscale[p.hw > 0 ? m / p.hw : 0] : 1.f;
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
  }

  if constexpr (REDUCE) {
    // sum over the 8 row-lanes of a column group (lane bits 3..5), then over the M-waves that share a sample
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
      // the whole tile lies in one sample: combine the 4 M-waves through LDS, one atomic per column
      float* red = reinterpret_cast<float*>(smem + RED_OFF);
      if (lane < 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          red[(wave * 2 + 0) * 64 + lane * 8 + j] = r0[j];
          if constexpr (EPI == VSX_EPI_DZ) red[(wave * 2 + 1) * 64 + lane * 8 + j] = r1[j];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (tid < BN && n0 + tid < p.N) {
        const int w0 = tid >> 6, c = tid & 63;  // N-half, column inside it
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          a0 += red[((w * 2 + w0) * 2 + 0) * 64 + c];
          if constexpr (EPI == VSX_EPI_DZ) a1 += red[((w * 2 + w0) * 2 + 1) * 64 + c];
        }
        atomicAdd(p.red0 + (size_t)b_tile * p.N + n0 + tid, a0);
        if constexpr (EPI == VSX_EPI_DZ) atomicAdd(p.red1 + (size_t)b_tile * p.N + n0 + tid, a1);
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

template <int EPI>
int launch(const VsxGemm* p, hipStream_t s) {
  const int tiles = (p->M / BM) * vsx_cdiv(p->N, BN);
  VsxGemm q = *p;
  // probe bits (tools/perf_nt_gen2.py): 4 = no MFMA, 8 = no DMA after the prologue, 16 = no epilogue stores, 32 = rotated K order
  q.pro |= (g_vsx_nt2 & 0x3C) << 8;
  if ((g_vsx_nt2 & 64) && p->K % 64 == 0)
    hipLaunchKernelGGL((gemm_nt2_kernel<EPI, 64, 2>), dim3(tiles), dim3(512), 0, s, q);
  else
    hipLaunchKernelGGL((gemm_nt2_kernel<EPI, 32, 3>), dim3(tiles), dim3(512), 0, s, q);
  VSX_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// 1 if this kernel family takes the launch (bf16 only; the caller has validated the common fields)
bool vsx_gemm_nt2_ok(const VsxGemm* p) {
  if (!g_vsx_nt2) return false;
  if (p->a_mode != VSX_A_ROWS || p->c_mode != VSX_A_ROWS || p->pro != VSX_PRO_NONE || p->nz > 1) return false;
  if (p->K % 32 != 0 || p->M % BM != 0 || p->N < 64 || p->N % 8 != 0) return false;
  if (p->epi < VSX_EPI_NONE || p->epi > VSX_EPI_DZ) return false;
  const bool per_sample = p->epi == VSX_EPI_BIAS_GELU_SQ || p->epi == VSX_EPI_DZ || p->rscale != nullptr;
  if (per_sample && (p->hw <= 0 || p->hw % 64 != 0)) return false;
  if (p->b_bstride != 0 && (p->hw <= 0 || p->hw % BM != 0)) return false;
  if (p->hw > 0 && p->hw % BM != 0 && p->hw != 64 && p->hw != 128) return false;
  if ((unsigned long long)BM * p->lda * 2 >= (1ull << 32) || (unsigned long long)p->N * p->ldb * 2 >= (1ull << 32)) return false;
  if (!(g_vsx_nt2 & 2) && (long)(p->M / BM) * vsx_cdiv(p->N, BN) < 512) return false;  // few tiles: the BK = 128 generic path
  return true;
}

int vsx_gemm_nt2(const VsxGemm* p, hipStream_t s) {
  switch (p->epi) {
    case VSX_EPI_NONE: return launch<VSX_EPI_NONE>(p, s);
    case VSX_EPI_BIAS: return launch<VSX_EPI_BIAS>(p, s);
    case VSX_EPI_BIAS_GELU_SQ: return launch<VSX_EPI_BIAS_GELU_SQ>(p, s);
    case VSX_EPI_BIAS_RES: return launch<VSX_EPI_BIAS_RES>(p, s);
    default: return launch<VSX_EPI_DZ>(p, s);
  }
}
