// MFMA GEMMs for the pointwise / patch / 3x3 convolutions of UNeXt2.
//
//   gemm_nt : C[M,N]  = pro(A)[M,K] · B[N,K]^T      forward and data-gradient GEMMs
//   gemm_tn : W[N,K] += X[M,N]^T · pro(A)[M,K]      weight-gradient GEMMs (contraction over pixels)
//
// gfx950 mapping: 256-thread workgroups = 4 wave64; 16x16 MFMA fragments
// (v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x4_f32 — same C layout, so one code path serves the
// bf16 production mode and the exact-fp32 parity mode); operands staged global → VGPR → LDS in
// 16-byte chunks (double-buffered LDS, one barrier per K step); LDS row strides chosen so the
// ds_read_b128 fragment reads are bank-conflict free; accumulators go back through LDS so the
// fused epilogues (bias, GELU + GRN sum-of-squares, residual, IN statistics …) run on row-contiguous
// 16-byte vectors and HBM stores are fully coalesced.
#include "vsx_common.h"
#include "../../include/vsx.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

extern int g_vsx_tn_tr;
extern int g_vsx_tn_rect;
extern int g_vsx_nt_wide;
extern int g_vsx_nt_fast;
extern int g_vsx_tn_wide;
extern int g_vsx_nt_stream;
extern int g_vsx_tn_want;
extern int g_vsx_tn_p2_rounds;
extern int g_vsx_tn_want2;
extern int g_vsx_tn_want3;
extern int g_vsx_tn_fill;
extern int g_vsx_tn_contig;
extern int g_vsx_tn_stream;
bool vsx_gemm_nt2_ok(const VsxGemm* p);           // gemm_nt2.hip
bool vsx_gemm_nt2_lnbwd_ok(const VsxGemm* p);
int vsx_gemm_nt2(const VsxGemm* p, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// operand gather: returns the 16-byte chunk holding A(m, k .. k+VN-1) after the prologue
// ------------------------------------------------------------------------------------------------
struct RowCoord {
  int b, y, x;
  bool valid;
};

__device__ __forceinline__ RowCoord decode_row(const VsxGemm& p, int m) {
  RowCoord r;
  r.valid = m < p.M;
  int mm = r.valid ? m : 0;
  if (p.a_mode == VSX_A_ROWS && p.c_mode == VSX_A_ROWS) {
    r.b = p.hw > 0 ? mm / p.hw : 0;
    r.y = 0;
    r.x = mm;  // plain row index
  } else {
    int hw = p.gh * p.gw;
    r.b = mm / hw;
    int rem = mm - r.b * hw;
    r.y = rem / p.gw;
    r.x = rem - r.y * p.gw;
  }
  return r;
}

template <typename T>
__device__ __forceinline__ typename VT<T>::vec load_a_chunk(const VsxGemm& p, const RowCoord& rc, int m, int k,
                                                            int coff) {
  constexpr int VN = VT<T>::N;
  typedef typename VT<T>::vec vec;
  if (!rc.valid || k >= p.K) return vzero<T>();
  const T* A = reinterpret_cast<const T*>(p.A);
  const T* ptr;
  if (p.a_mode == VSX_A_ROWS) {
    ptr = A + (size_t)m * p.lda + coff + k;
  } else {
    int tap = k / p.cs;
    int c = k - tap * p.cs;
    size_t pix;
    if (p.a_mode == VSX_A_PATCH2) {
      int ky = tap >> 1, kx = tap & 1;
      pix = ((size_t)rc.b * (2 * p.gh) + 2 * rc.y + ky) * (2 * p.gw) + 2 * rc.x + kx;
    } else {
      int ky = tap / 3, kx = tap - 3 * ky;
      int yy = rc.y + ky - 1, xx = rc.x + kx - 1;
      if (yy < 0 || yy >= p.gh || xx < 0 || xx >= p.gw) return vzero<T>();
      pix = ((size_t)rc.b * p.gh + yy) * p.gw + xx;
    }
    ptr = A + pix * p.lda + coff + c;
  }
  return ldvec<T>(ptr);
}

// GRN applied on the fly to an already-activated operand: a = g * s[b, k] + beta[k].  Runs when the staged
// registers are written to LDS (AFTER the MFMA work of the current tile), so the global loads of the raw
// chunk stay in flight across the compute phase.
template <typename T>
__device__ __forceinline__ typename VT<T>::vec apply_prologue(const VsxGemm& p, typename VT<T>::vec v, int b, int k) {
  constexpr int VN = VT<T>::N;
  if (p.pro != VSX_PRO_GRN || k >= p.K) return v;
  float f[VN];
  unpack<T>(v, f);
  const float* s = p.grn_s + (size_t)b * p.K + k;
  const float* bt = p.grn_b + k;
#pragma unroll
  for (int j = 0; j < VN; j += 4) {
    const float4 sv = *reinterpret_cast<const float4*>(s + j);
    const float4 bv = *reinterpret_cast<const float4*>(bt + j);
    f[j] = fmaf(f[j], sv.x, bv.x); f[j + 1] = fmaf(f[j + 1], sv.y, bv.y);
    f[j + 2] = fmaf(f[j + 2], sv.z, bv.z); f[j + 3] = fmaf(f[j + 3], sv.w, bv.w);
  }
  return pack<T>(f);
}

// ------------------------------------------------------------------------------------------------
// MFMA fragments
// ------------------------------------------------------------------------------------------------
template <typename T>
struct Frag;
template <>
struct Frag<bf16_t> {
  typedef bf16x8 type;
  static constexpr int MK = 32;
};
template <>
struct Frag<float> {
  typedef float type;
  static constexpr int MK = 4;
};

__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(const float& a, const float& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// fragment of a [rows][k] (k contiguous) LDS tile, row stride RS bytes; kk = MFMA sub-step inside BK
template <typename T>
__device__ __forceinline__ typename Frag<T>::type lds_frag_rk(const char* tile, int row, int RS, int kk, int kq);
template <>
__device__ __forceinline__ bf16x8 lds_frag_rk<bf16_t>(const char* tile, int row, int RS, int kk, int kq) {
  return *reinterpret_cast<const bf16x8*>(tile + row * RS + kk * 64 + kq * 16);
}
template <>
__device__ __forceinline__ float lds_frag_rk<float>(const char* tile, int row, int RS, int kk, int kq) {
  return *reinterpret_cast<const float*>(tile + row * RS + (kk * 4 + kq) * 4);
}

// ------------------------------------------------------------------------------------------------
// gemm_nt
// ------------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int WM_, int WN_, int BK, int NBUF = 2>
__global__ __launch_bounds__(256, NBUF == 1 ? 3 : 1) void gemm_nt_kernel(const VsxGemm p) {
  constexpr int ES = sizeof(T);
  constexpr int RS = BK * ES + (ES == 2 ? 32 : 16);
  constexpr int CPR = BK * ES / 16;
  constexpr int VN = VT<T>::N;
  constexpr int FM = BM / WM_ / 16, FN = BN / WN_ / 16;
  constexpr int NA = (BM * CPR + 255) / 256, NB = (BN * CPR + 255) / 256;
  constexpr int MAIN_BYTES = NBUF * (BM + BN) * RS;
  constexpr int CS_LD = BN + 4;
  constexpr int MAXBT = 8;  // batch samples one tile may span on the LDS reduction path
  constexpr int EPI_BYTES = (BM / (BN >= 64 ? 2 : 1)) * CS_LD * 4 + 2 * MAXBT * BN * 4;
  constexpr int LDS_BYTES = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
  constexpr int MK = Frag<T>::MK;
  typedef typename VT<T>::vec vec;
  typedef typename Frag<T>::type frag_t;
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN_, wn = wave % WN_;
  const int p16 = lane & 15, kq = lane >> 4;
  const int z = blockIdx.z;

  // XCD-aware tile order: consecutive tile ids (sharing an A row-panel) stay on one XCD's L2
  const int tiles_n = (p.N + BN - 1) / BN;
  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
  const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int a_coff = p.a_coff[z];
  const T* Bw = reinterpret_cast<const T*>(p.B) + p.b_off[z];

  // per-thread staging descriptors
  RowCoord arc[NA];
  int arow[NA], ach[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    int cid = tid + i * 256;
    arow[i] = cid / CPR;
    ach[i] = cid % CPR;
    arc[i] = decode_row(p, m0 + arow[i]);
    if (cid >= BM * CPR) arc[i].valid = false;
  }
  vec areg[1][NA], breg[1][NB];

  auto load_tiles = [&](int kt, vec* ar, vec* br) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int k = kt * BK + ach[i] * VN;
      ar[i] = load_a_chunk<T>(p, arc[i], m0 + arow[i], k, a_coff);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int cid = tid + i * 256;
      int row = cid / CPR, ch = cid % CPR;
      int n = n0 + row, k = kt * BK + ch * VN;
      if (cid < BN * CPR && n < p.N && k < p.K)
        br[i] = ldvec<T>(Bw + (size_t)n * p.ldb + k);
      else
        br[i] = vzero<T>();
    }
  };
  auto store_tiles = [&](int kt, int buf, const vec* ar, const vec* br) {
    char* As = smem + buf * (BM + BN) * RS;
    char* Bs = As + BM * RS;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int cid = tid + i * 256;
      if (cid < BM * CPR) {
        vec v = ar[i];
        if (p.pro != VSX_PRO_NONE && arc[i].valid) v = apply_prologue<T>(p, v, arc[i].b, kt * BK + ach[i] * VN);
        *reinterpret_cast<vec*>(As + arow[i] * RS + ach[i] * 16) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int cid = tid + i * 256;
      if (cid < BN * CPR) *reinterpret_cast<vec*>(Bs + (cid / CPR) * RS + (cid % CPR) * 16) = br[i];
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const char* As = smem + buf * (BM + BN) * RS;
    const char* Bs = As + BM * RS;
#pragma unroll
    for (int kk = 0; kk < BK / MK; ++kk) {
      frag_t af[FM], bf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = lds_frag_rk<T>(As, (wm * FM + i) * 16 + p16, RS, kk, kq);
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[j] = lds_frag_rk<T>(Bs, (wn * FN + j) * 16 + p16, RS, kk, kq);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
    }
  };

  const int nk = (p.K + BK - 1) / BK;
  load_tiles(0, areg[0], breg[0]);
  if constexpr (NBUF == 2) {
    store_tiles(0, 0, areg[0], breg[0]);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) load_tiles(kt + 1, areg[0], breg[0]);   // in flight across the MFMA phase
      compute(kt & 1);
      if (kt + 1 < nk) store_tiles(kt + 1, (kt + 1) & 1, areg[0], breg[0]);  // prologue math happens here
      __syncthreads();
    }
  } else {
    // one LDS buffer, twice the K per slab: the registers carry a whole 2x-wide slab in flight across the MFMA
    // phase (bytes in flight per workgroup double at the same LDS footprint -> same workgroups per CU)
    for (int kt = 0; kt < nk; ++kt) {
      store_tiles(kt, 0, areg[0], breg[0]);
      __syncthreads();
      if (kt + 1 < nk) load_tiles(kt + 1, areg[0], breg[0]);
      compute(0);
      __syncthreads();
    }
  }

  // ---- epilogue: accumulators → LDS (fp32) → row-contiguous vectors, in two passes of BM/2 rows so that the
  // staging buffer (not the K-loop buffers) never decides how many workgroups fit on a CU
  constexpr int NPASS = BN >= 64 ? 2 : 1;
  constexpr int HR = BM / NPASS;  // rows per pass
  float* Cs = reinterpret_cast<float*>(smem);
  float* red = Cs + HR * CS_LD;  // [2][MAXBT][BN]
  constexpr int NCH = BN / VN;        // column chunks per row
  constexpr int RSTEP = 256 / NCH;    // rows handled per sweep
  constexpr bool PARK = 2 * RSTEP <= HR;  // partial-sum parking needs two Cs rows per thread
  const int cc = tid % NCH, rr = tid / NCH;
  const int n = n0 + cc * VN;
  const bool ncol_ok = n < p.N;
  const int epi = p.epi;
  const bool reduce = (epi == VSX_EPI_BIAS_GELU_SQ || epi == VSX_EPI_DZ || epi == VSX_EPI_BIAS_STATS);
  const int mlast = (m0 + BM < p.M ? m0 + BM : p.M) - 1;
  const int hwb = p.hw > 0 ? p.hw : p.M;
  const int b_first = m0 / hwb;
  const int nbt = mlast / hwb - b_first + 1;       // samples covered by this tile
  const bool uniform = reduce && nbt <= MAXBT;     // LDS reduction path (else: direct global atomics)
  const bool park = uniform && nbt == 1 && PARK;
  float r0[VN], r1[VN];
#pragma unroll
  for (int j = 0; j < VN; ++j) { r0[j] = 0.f; r1[j] = 0.f; }
  float bias[VN];
#pragma unroll
  for (int j = 0; j < VN; ++j) bias[j] = 0.f;
  if (ncol_ok && p.bias != nullptr && epi != VSX_EPI_NONE && epi != VSX_EPI_DZ) {
#pragma unroll
    for (int j = 0; j < VN; ++j) bias[j] = p.bias[n + j];
  }
  T* Cg = reinterpret_cast<T*>(p.C);
  const int c_coff = p.c_coff[z];
  if (reduce) {
    for (int i = tid; i < 2 * MAXBT * BN; i += 256) red[i] = 0.f;
  }

#pragma unroll 1
  for (int half = 0; half < NPASS; ++half) {
    __syncthreads();  // K-loop buffers / previous pass no longer read
    if ((wm * FM * 16) / HR == half) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            Cs[((wm * FM + i) * 16 - half * HR + kq * 4 + r) * CS_LD + (wn * FN + j) * 16 + p16] = acc[i][j][r];
    }
    __syncthreads();
    if (ncol_ok) {
      for (int row = rr; row < HR; row += RSTEP) {
        const int m = m0 + half * HR + row;
        if (m >= p.M) break;
        float v[VN];
#pragma unroll
        for (int j = 0; j < VN; j += 4) {
          float4 t = *reinterpret_cast<const float4*>(Cs + row * CS_LD + cc * VN + j);
          v[j] = t.x; v[j + 1] = t.y; v[j + 2] = t.z; v[j + 3] = t.w;
        }
#pragma unroll
        for (int j = 0; j < VN; ++j) v[j] += bias[j];
        const int b = m / hwb;
        if (epi == VSX_EPI_BIAS_RES) {
          float rf[VN];
          unpack<T>(ldvec<T>(reinterpret_cast<const T*>(p.res) + (size_t)m * p.ldr + n), rf);
          const float rs = p.rscale ? p.rscale[b] : 1.f;  // stochastic depth: the branch of sample b is dropped / rescaled
#pragma unroll
          for (int j = 0; j < VN; ++j) v[j] = fmaf(v[j], rs, rf[j]);
        } else if (epi == VSX_EPI_BIAS_GELU_SQ) {
          // second output: the activation g = gelu(h) (storage-rounded); the GRN statistics use the stored value
          float gv[VN];
#pragma unroll
          for (int j = 0; j < VN; j += 2) {  // packed fp32 pairs (gelu_parts2)
            const vsx_v2f x = {round_to<T>(v[j]), round_to<T>(v[j + 1])};
            vsx_v2f cdf, pdf;
            gelu_parts2(x, cdf, pdf);
            const vsx_v2f gg = x * cdf;
            gv[j] = round_to<T>(gg.x);
            gv[j + 1] = round_to<T>(gg.y);
            r0[j] += gv[j] * gv[j];
            r0[j + 1] += gv[j + 1] * gv[j + 1];
          }
          stvec<T>(reinterpret_cast<T*>(p.C2) + (size_t)m * p.ldc + c_coff + n, pack<T>(gv));
        } else if (epi == VSX_EPI_DZ) {
          float gf[VN];  // aux = stored activation g
          unpack<T>(ldvec<T>(reinterpret_cast<const T*>(p.aux) + (size_t)m * p.ldx + n), gf);
#pragma unroll
          for (int j = 0; j < VN; ++j) {
            float dz = round_to<T>(v[j]);
            r0[j] += dz * gf[j];
            r1[j] += dz;
          }
        } else if (epi == VSX_EPI_BIAS_STATS) {
#pragma unroll
          for (int j = 0; j < VN; ++j) {
            float c = round_to<T>(v[j]);
            r0[j] += c;
            r1[j] += c * c;
          }
        }
        T* dst;
        if (p.c_mode == VSX_A_PATCH2) {
          int hw = p.gh * p.gw;
          int bb = m / hw, rem = m - bb * hw;
          int y = rem / p.gw, x = rem - y * p.gw;
          int tap = n / p.c_cs, c = n - tap * p.c_cs;
          size_t pix = ((size_t)bb * (2 * p.gh) + 2 * y + (tap >> 1)) * (2 * p.gw) + 2 * x + (tap & 1);
          dst = Cg + pix * p.ldc + c_coff + c;
        } else {
          dst = Cg + (size_t)m * p.ldc + c_coff + n;
        }
        if (epi != VSX_EPI_BIAS_GELU_SQ || Cg != nullptr) stvec<T>(dst, pack<T>(v));
        if (reduce && !uniform) {
          // tile spans more than MAXBT batch samples (very small feature maps): direct atomics
#pragma unroll
          for (int j = 0; j < VN; ++j) {
            atomicAdd(p.red0 + (size_t)b * p.N + n + j, r0[j]);
            if (epi == VSX_EPI_BIAS_STATS || epi == VSX_EPI_DZ) atomicAdd(p.red1 + (size_t)b * p.N + n + j, r1[j]);
            r0[j] = 0.f;
            r1[j] = 0.f;
          }
        } else if (reduce && !park) {
          // rows of one sample are contiguous: keep accumulating in registers while the sample is
          // unchanged, flush to the sample's LDS slot when the next row belongs to another sample
          const int rnext = row + RSTEP;
          const bool flush = rnext >= HR || m0 + half * HR + rnext >= p.M || (m0 + half * HR + rnext) / hwb != b;
          if (flush) {
            float* rs = red + (size_t)(b - b_first) * BN + cc * VN;
#pragma unroll
            for (int j = 0; j < VN; ++j) {
              atomicAdd(rs + j, r0[j]);
              if (epi != VSX_EPI_BIAS_GELU_SQ) atomicAdd(rs + MAXBT * BN + j, r1[j]);
              r0[j] = 0.f;
              r1[j] = 0.f;
            }
          }
        }
      }
    }
  }
  if (park) {
    // single-sample tile (the common case): no atomics in LDS.  A thread's Cs rows are dead once it has
    // read them, so it parks its partial column sums (both passes) in its own first two rows; then BN
    // threads add the RSTEP partial rows of each column.
    __syncthreads();
    if (ncol_ok) {
#pragma unroll
      for (int j = 0; j < VN; ++j) {
        Cs[rr * CS_LD + cc * VN + j] = r0[j];
        Cs[(rr + RSTEP) * CS_LD + cc * VN + j] = r1[j];
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.N) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll 4
      for (int r = 0; r < RSTEP; ++r) {
        a0 += Cs[r * CS_LD + tid];
        a1 += Cs[(r + RSTEP) * CS_LD + tid];
      }
      atomicAdd(p.red0 + (size_t)b_first * p.N + n0 + tid, a0);
      if (epi == VSX_EPI_BIAS_STATS || epi == VSX_EPI_DZ) atomicAdd(p.red1 + (size_t)b_first * p.N + n0 + tid, a1);
    }
  } else if (uniform) {
    __syncthreads();
    for (int i = tid; i < nbt * BN; i += 256) {
      const int bs = i / BN, c = i - bs * BN;
      if (n0 + c < p.N) {
        atomicAdd(p.red0 + (size_t)(b_first + bs) * p.N + n0 + c, red[i]);
        if (epi == VSX_EPI_BIAS_STATS || epi == VSX_EPI_DZ)
          atomicAdd(p.red1 + (size_t)(b_first + bs) * p.N + n0 + c, red[MAXBT * BN + i]);
      }
    }
  }
}

// a = g * s[k..] + beta[k..] on one 16-byte chunk (GRN prologue of the lean kernels)
template <typename T>
__device__ __forceinline__ typename VT<T>::vec grn_apply(typename VT<T>::vec v, const float* grn_s, const float* grn_b, int k) {
  constexpr int VN = VT<T>::N;
  float f[VN];
  unpack<T>(v, f);
#pragma unroll
  for (int j = 0; j < VN; j += 4) {
    const float4 sv = *reinterpret_cast<const float4*>(grn_s + k + j);
    const float4 bv = *reinterpret_cast<const float4*>(grn_b + k + j);
    f[j] = fmaf(f[j], sv.x, bv.x); f[j + 1] = fmaf(f[j + 1], sv.y, bv.y);
    f[j + 2] = fmaf(f[j + 2], sv.z, bv.z); f[j + 3] = fmaf(f[j + 3], sv.w, bv.w);
  }
  return pack<T>(f);
}

// ------------------------------------------------------------------------------------------------
// gemm_nt, lean instantiation for the shapes that carry the step: plain row operands, 128x128 tile, K a multiple
// of 32, tile inside one batch sample.  Same math and LDS layout as gemm_nt_kernel; what changes is the
// instruction count (PMC on the generic kernel: 1443 VALU + 1021 SALU instructions per wave for 16-112 MFMAs —
// issue-bound, not memory-bound): the epilogue kind and the prologue are template parameters, operand addresses
// are one scalar base per K-slab plus a per-lane 32-bit offset computed once, out-of-range rows / columns are
// clamped at load time (their results are never stored) instead of predicated per chunk per slab.
// ------------------------------------------------------------------------------------------------
template <typename T, int EPI, bool PRO, int BK = 32, int NBUF = 2, int BM = 128>
__global__ __launch_bounds__(256, BM == 256 ? 2 : ((BK == 64 && NBUF == 1) ? 3 : 1)) void gemm_nt_fast_kernel(const VsxGemm p) {
  // (BM = 256 — 128x64 wave tiles — measured -5..-9 % on isolated wide-output launches and nothing on the whole step; its
  // dispatch (`nt_tall`) was removed in round 4: the launches it served run on the 256 x BN kernel of gemm_nt2.hip)
  constexpr int BN = 128, WN_ = 2, FM = BM / 32, FN = 4;
  constexpr int ES = sizeof(T);
  constexpr int RS = BK * ES + (ES == 2 ? 32 : 16);
  constexpr int CPR = BK * ES / 16;
  constexpr int VN = VT<T>::N;
  constexpr int NA = BM * CPR / 256, NB = BN * CPR / 256;
  constexpr int STAGE = (BM + BN) * RS;
  constexpr int CS_LD = BN + 4;
  constexpr int HR = 64;               // rows per epilogue pass
  constexpr int NPASS = BM / HR;
  constexpr int EPI_BYTES = HR * CS_LD * 4;
  constexpr int LDS_BYTES = NBUF * STAGE > EPI_BYTES ? NBUF * STAGE : EPI_BYTES;
  constexpr int MK = Frag<T>::MK;
  constexpr bool REDUCE = (EPI == VSX_EPI_BIAS_GELU_SQ || EPI == VSX_EPI_DZ);
  typedef typename VT<T>::vec vec;
  typedef typename Frag<T>::type frag_t;
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN_, wn = wave % WN_;
  const int p16 = lane & 15, kq = lane >> 4;
  const int z = blockIdx.z;

  const int tiles_n = (p.N + BN - 1) / BN;
  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);  // XCD-aware (see gemm_nt_kernel)
  const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int b_tile = p.hw > 0 ? m0 / p.hw : 0;  // the tile lies inside one sample (dispatch guarantees it)
  // the A panel is addressed from the tile's own first row (64-bit scalar base per workgroup), so the per-lane 32-bit
  // offsets stay below BM * lda bytes whatever M is (M * lda * es passes 4 GB at B = 640 patches / B = 16 at 2048^2)
  // round 6: the 2 x 2 stride-2 patch gather (VSX_A_PATCH2: the downsampling projections) on the lean kernel.  Row m = output
  // pixel (b, y, x) of the gh x gw grid reads input pixel (2y + ky, 2x + kx) of the [B, 2gh, 2gw, cs] tensor for k = (ky, kx, c):
  // the K slab picks (ky, kx, c0) — one scalar offset per slab (cs % BK == 0, dispatch) — and the lane offset is the distance of
  // the row's top-left input pixel from the tile's first one (monotonic in m, < 2^32 bytes: dispatch)
  const bool patch = p.a_mode == VSX_A_PATCH2;
  auto pix0 = [&](int m) -> size_t {
    const int hwg = p.gh * p.gw;
    const int bb = m / hwg, rem = m - bb * hwg;
    const int yy = rem / p.gw, xx = rem - yy * p.gw;
    return ((size_t)bb * (2 * p.gh) + 2 * yy) * (size_t)(2 * p.gw) + 2 * xx;
  };
  const size_t pixm0 = patch ? pix0(m0) : 0;
  const char* Abase = reinterpret_cast<const char*>(p.A) +
                      ((size_t)p.a_coff[z] + (patch ? pixm0 : (size_t)m0) * (size_t)p.lda) * ES;
  const char* Bbase = reinterpret_cast<const char*>(p.B) + ((size_t)p.b_off[z] + (size_t)b_tile * (size_t)p.b_bstride) * ES;
  uint32_t offA[NA], offB[NB];
  int ldsA[NA], ldsB[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int cid = tid + i * 256, row = cid / CPR, ch = cid % CPR;
    int m = m0 + row;
    m = m < p.M ? m : p.M - 1;
    offA[i] = (patch ? (uint32_t)(pix0(m) - pixm0) : (uint32_t)(m - m0)) * (uint32_t)(p.lda * ES) + ch * 16;
    ldsA[i] = row * RS + ch * 16;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int cid = tid + i * 256, row = cid / CPR, ch = cid % CPR;
    int n = n0 + row;
    n = n < p.N ? n : p.N - 1;
    offB[i] = (uint32_t)n * (uint32_t)(p.ldb * ES) + ch * 16;
    ldsB[i] = BM * RS + row * RS + ch * 16;
  }
  const float* grn_s = PRO ? p.grn_s + (size_t)b_tile * p.K : nullptr;
  // hw % 128 == 64 (8 x 8 feature maps): a 128-row tile holds two samples, one per 64-row epilogue pass; the GRN prologue
  // picks the scale row of the sample its A row belongs to, the column reductions are flushed after every pass
  const bool split_tile = BM == 128 && p.hw > 0 && (p.hw % 128) != 0;

  vec ar[NA], br[NB];
  const int nk = (p.K + BK - 1) / BK;
  const int ktail = p.K % BK;  // round 6: K need not be a multiple of BK (K = 144 of the 64 x 64 decoder projection): the chunks of
                               // the LAST slab at k >= K are loaded from the row's first chunk instead and zeroed (both operands)
  auto gload = [&](int kt, vec* ar, vec* br) {
    size_t aoff = (size_t)kt * (BK * ES);
    if (patch) {
      const int k0 = kt * BK, tap = k0 / p.cs, c0 = k0 - tap * p.cs;
      aoff = ((size_t)((tap >> 1) * (2 * p.gw) + (tap & 1)) * (size_t)p.lda + c0) * ES;
    }
    const char* Ak = Abase + aoff;
    const char* Bk = Bbase + (size_t)kt * (BK * ES);
    if (ktail != 0 && kt == nk - 1) {
      const int ch = tid % CPR;  // (256 % CPR == 0: every chunk of this thread sits at the same K position)
      const bool ok = ch * VN < ktail;
      const int back = ok ? 0 : ch * 16;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const vec v = *reinterpret_cast<const vec*>(Ak + offA[i] - back);
        ar[i] = ok ? v : vzero<T>();
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const vec v = *reinterpret_cast<const vec*>(Bk + offB[i] - back);
        br[i] = ok ? v : vzero<T>();
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) ar[i] = *reinterpret_cast<const vec*>(Ak + offA[i]);  // (non-temporal loads of a read-once A panel: +-0, round 4)
#pragma unroll
    for (int i = 0; i < NB; ++i) br[i] = *reinterpret_cast<const vec*>(Bk + offB[i]);
  };
  auto lstore = [&](int kt, int buf, const vec* ar, const vec* br) {
    char* S = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      vec v = ar[i];
      if constexpr (PRO) {  // a = g * s[b, k] + beta[k]
        const int k = kt * BK + ((tid + i * 256) % CPR) * VN;
        const float* grn_row = grn_s;
        if (split_tile) {
          const int m = m0 + (tid + i * 256) / CPR;
          grn_row = p.grn_s + (size_t)((m < p.M ? m : p.M - 1) / p.hw) * p.K;
        }
        float f[VN];
        unpack<T>(v, f);
#pragma unroll
        for (int j = 0; j < VN; j += 4) {
          const float4 sv = *reinterpret_cast<const float4*>(grn_row + k + j);
          const float4 bv = *reinterpret_cast<const float4*>(p.grn_b + k + j);
          f[j] = fmaf(f[j], sv.x, bv.x); f[j + 1] = fmaf(f[j + 1], sv.y, bv.y);
          f[j + 2] = fmaf(f[j + 2], sv.z, bv.z); f[j + 3] = fmaf(f[j + 3], sv.w, bv.w);
        }
        v = pack<T>(f);
      }
      *reinterpret_cast<vec*>(S + ldsA[i]) = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const vec vb = br[i];  // through a value: a direct array -> LDS aggregate copy keeps the array in scratch
      *reinterpret_cast<vec*>(S + ldsB[i]) = vb;
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fragA = (wm * FM * 16 + p16) * RS, fragB = BM * RS + (wn * FN * 16 + p16) * RS;
  auto compute = [&](int buf) {
    const char* S = smem + buf * STAGE;
#pragma unroll
    for (int kk = 0; kk < BK / MK; ++kk) {
      frag_t af[FM], bf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = lds_frag_rk<T>(S + fragA, i * 16, RS, kk, kq);
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[j] = lds_frag_rk<T>(S + fragB, j * 16, RS, kk, kq);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
    }
  };

  // one K-slab of register staging ahead of the MFMA work, LDS double buffer.  (Measured: staging two slabs ahead —
  // with unconditional loads so that the compiler really keeps `s_waitcnt vmcnt(4)` — changes nothing: the loop is
  // bound by LDS bandwidth (48 KB of LDS traffic per 16 MFMAs per wave tile of 64x64), not by load latency.)
  gload(0, ar, br);
  if constexpr (NBUF == 2) {
    lstore(0, 0, ar, br);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) gload(kt + 1, ar, br);  // in flight across the MFMA phase
      compute(kt & 1);
      if (kt + 1 < nk) lstore(kt + 1, (kt + 1) & 1, ar, br);
      __syncthreads();
    }
  } else {
    // one LDS buffer of twice the K depth (128-byte row pieces: full cache lines through L1 — measured +25 % L2-hit
    // throughput over 64-byte pieces, tools/micro/stream_tiles.hip) at the same LDS footprint
    for (int kt = 0; kt < nk; ++kt) {
      lstore(kt, 0, ar, br);
      __syncthreads();
      if (kt + 1 < nk) gload(kt + 1, ar, br);
      compute(0);
      __syncthreads();
    }
  }

  // ---- epilogue (two passes of 64 rows through a fp32 staging tile, row-contiguous 16-byte stores)
  float* Cs = reinterpret_cast<float*>(smem);
  constexpr int NCH = BN / VN;
  constexpr int RSTEP = 256 / NCH;
  const int cc = tid % NCH, rr = tid / NCH;
  const int n = n0 + cc * VN;
  const bool ncol_ok = n < p.N;
  float bias[VN];
#pragma unroll
  for (int j = 0; j < VN; ++j) bias[j] = 0.f;
  if constexpr (EPI != VSX_EPI_NONE && EPI != VSX_EPI_DZ) {
    if (ncol_ok && p.bias != nullptr) {
#pragma unroll
      for (int j = 0; j < VN; ++j) bias[j] = p.bias[n + j];
    }
  }
  float r0[VN], r1[VN];
#pragma unroll
  for (int j = 0; j < VN; ++j) { r0[j] = 0.f; r1[j] = 0.f; }
  const size_t ccol = (size_t)p.c_coff[z] + n;
  const bool ntst = (EPI == VSX_EPI_BIAS_GELU_SQ || EPI == VSX_EPI_DZ) && (p.pro & 256) != 0;  // set by launch_nt_fast
  const bool ntld = EPI == VSX_EPI_DZ && (p.pro & 512) != 0;  // the dZ epilogue is the last reader of the stored activation
#pragma unroll 1
  for (int half = 0; half < NPASS; ++half) {
    __syncthreads();
    // pass `half` = tile rows [64*half, 64*half + 64): owned by wave row wm = half / (NPASS/2), its fragments
    // (half % (NPASS/2)) * 4 .. + 4
    if (wm == half / (NPASS / 2)) {
#pragma unroll
      for (int ip = 0; ip < NPASS / 2; ++ip) {
        if (ip == half % (NPASS / 2)) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                Cs[(i * 16 + kq * 4 + r) * CS_LD + (wn * FN + j) * 16 + p16] = acc[ip * 4 + i][j][r];
        }
      }
    }
    __syncthreads();
    if (ncol_ok) {
#pragma unroll
      for (int it = 0; it < HR / RSTEP; ++it) {
        const int row = rr + it * RSTEP;
        const int m = m0 + half * HR + row;
        if (m < p.M) {
          float v[VN];
#pragma unroll
          for (int j = 0; j < VN; j += 4) {
            const float4 t = *reinterpret_cast<const float4*>(Cs + row * CS_LD + cc * VN + j);
            v[j] = t.x + bias[j]; v[j + 1] = t.y + bias[j + 1]; v[j + 2] = t.z + bias[j + 2]; v[j + 3] = t.w + bias[j + 3];
          }
          if constexpr (EPI == VSX_EPI_BIAS_RES) {
            float rf[VN];
            unpack<T>(ldvec<T>(reinterpret_cast<const T*>(p.res) + (size_t)m * p.ldr + n), rf);
            const float rs = p.rscale ? p.rscale[p.hw > 0 ? m / p.hw : 0] : 1.f;  // stochastic depth (see VsxGemm.rscale)
#pragma unroll
            for (int j = 0; j < VN; ++j) v[j] = fmaf(v[j], rs, rf[j]);
          } else if constexpr (EPI == VSX_EPI_BIAS_GELU_SQ) {
            float gv[VN];
#pragma unroll
            for (int j = 0; j < VN; j += 2) {  // packed fp32 pairs (gelu_parts2)
              const vsx_v2f x = {round_to<T>(v[j]), round_to<T>(v[j + 1])};
              vsx_v2f cdf, pdf;
              gelu_parts2(x, cdf, pdf);
              const vsx_v2f gg = x * cdf;
              gv[j] = round_to<T>(gg.x);
              gv[j + 1] = round_to<T>(gg.y);
              r0[j] += gv[j] * gv[j];
              r0[j + 1] += gv[j + 1] * gv[j + 1];
            }
            stvec_stream(reinterpret_cast<T*>(p.C2) + (size_t)m * p.ldc + ccol, pack<T>(gv), ntst);
          } else if constexpr (EPI == VSX_EPI_DZ) {
            float gf[VN];
            unpack<T>(ldvec_stream<T>(reinterpret_cast<const T*>(p.aux) + (size_t)m * p.ldx + n, ntld), gf);
#pragma unroll
            for (int j = 0; j < VN; ++j) {
              const float dz = round_to<T>(v[j]);
              r0[j] += dz * gf[j];
              r1[j] += dz;
            }
          }
          // inference passes C = nullptr with EPI_BIAS_GELU_SQ: only the activation (C2) is kept, not the pre-activation
          if constexpr (EPI == VSX_EPI_NONE || EPI == VSX_EPI_BIAS) {
            if (p.c_mode == VSX_A_PATCH2) {  // round 6: scatter the row back to its 2 x 2 input patch, n = (ky, kx, c) (dispatch: bit 1)
              const int tap = n / p.c_cs, c = n - tap * p.c_cs;
              const size_t pix = pix0(m) + (size_t)(tap >> 1) * (2 * p.gw) + (tap & 1);
              stvec<T>(reinterpret_cast<T*>(p.C) + pix * p.ldc + (size_t)p.c_coff[z] + c, pack<T>(v));
              continue;
            }
          }
          if (EPI != VSX_EPI_BIAS_GELU_SQ || p.C != nullptr)
            stvec_stream(reinterpret_cast<T*>(p.C) + (size_t)m * p.ldc + ccol, pack<T>(v), ntst);
        }
      }
    }
    if constexpr (REDUCE) {
      if (split_tile && half + 1 < NPASS && m0 + half * HR < p.M) {
        // this pass was one whole sample: its column sums go out now (same park-and-add as below)
        __syncthreads();
        if (ncol_ok) {
#pragma unroll
          for (int j = 0; j < VN; ++j) {
            Cs[rr * CS_LD + cc * VN + j] = r0[j];
            Cs[(rr + RSTEP) * CS_LD + cc * VN + j] = r1[j];
            r0[j] = 0.f;
            r1[j] = 0.f;
          }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.N) {
          float a0 = 0.f, a1 = 0.f;
#pragma unroll 4
          for (int r = 0; r < RSTEP; ++r) {
            a0 += Cs[r * CS_LD + tid];
            a1 += Cs[(r + RSTEP) * CS_LD + tid];
          }
          const size_t bs = (size_t)((m0 + half * HR) / p.hw);
          atomicAdd(p.red0 + bs * p.N + n0 + tid, a0);
          if constexpr (EPI == VSX_EPI_DZ) atomicAdd(p.red1 + bs * p.N + n0 + tid, a1);
        }
      }
    }
  }
  if constexpr (REDUCE) {
    // column sums of the tile: park the per-thread partials in the (dead) staging rows, then BN threads add them
    __syncthreads();
    if (ncol_ok) {
#pragma unroll
      for (int j = 0; j < VN; ++j) {
        Cs[rr * CS_LD + cc * VN + j] = r0[j];
        Cs[(rr + RSTEP) * CS_LD + cc * VN + j] = r1[j];
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.N) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll 4
      for (int r = 0; r < RSTEP; ++r) {
        a0 += Cs[r * CS_LD + tid];
        a1 += Cs[(r + RSTEP) * CS_LD + tid];
      }
      // (split_tile: what is left in r0 / r1 is the LAST pass = the tile's last sample)
      const size_t bs = split_tile ? (size_t)((m0 + (NPASS - 1) * HR < p.M ? m0 + (NPASS - 1) * HR : p.M - 1) / p.hw) : (size_t)b_tile;
      atomicAdd(p.red0 + bs * p.N + n0 + tid, a0);
      if constexpr (EPI == VSX_EPI_DZ) atomicAdd(p.red1 + bs * p.N + n0 + tid, a1);
    }
  }
}

template <typename T, int EPI, bool PRO>
static int launch_nt_fast(const VsxGemm* pin, hipStream_t s) {
  g_vsx_last_kernel = "gemm_nt_fast";
  VsxGemm pq = *pin;
  if (g_vsx_nt_stream & 1) pq.pro |= 256;
  if (g_vsx_nt_stream & 2) pq.pro |= 512;  // kernel-side flag bit (the prologue kind itself is a template parameter there)
  const VsxGemm* p = &pq;
  int tiles = vsx_cdiv(p->M, 128) * vsx_cdiv(p->N, 128);
  dim3 grid(tiles, 1, p->nz > 0 ? p->nz : 1);
  if constexpr (sizeof(T) == 2) {
    const bool k64 = p->K % 64 == 0 && p->K >= 256 && (p->a_mode == VSX_A_ROWS || p->cs % 64 == 0);
    if (g_vsx_nt_wide == 1 && k64) {
      hipLaunchKernelGGL((gemm_nt_fast_kernel<T, EPI, PRO, 64, 1>), grid, dim3(256), 0, s, *p);
      VSX_LAUNCH_CHECK();
      return 0;
    }
    if (g_vsx_nt_wide == 2 && k64) {
      hipLaunchKernelGGL((gemm_nt_fast_kernel<T, EPI, PRO, 64, 2>), grid, dim3(256), 0, s, *p);
      VSX_LAUNCH_CHECK();
      return 0;
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (g_vsx_nt_wide) {
      // short K (fc1, fc2 data gradient: epilogue-dominated): BK = 32 with ONE LDS buffer = 33.8 KB, 124 registers ->
      // 4 workgroups per CU instead of 3 (measured -5..-10 % on these launches)
      hipLaunchKernelGGL((gemm_nt_fast_kernel<T, EPI, PRO, 32, 1>), grid, dim3(256), 0, s, *p);
      VSX_LAUNCH_CHECK();
      return 0;
    }
  }
  hipLaunchKernelGGL((gemm_nt_fast_kernel<T, EPI, PRO>), grid, dim3(256), 0, s, *p);
  VSX_LAUNCH_CHECK();
  return 0;
}

static bool nt_fast_ok(const VsxGemm* p, int es) {
  // round 6 (nt_fast bit 1): K tails (K % 32 != 0: whole 16-byte chunks, zero-filled last slab) and the 2 x 2 patch gather
  const bool ext = (g_vsx_nt_fast & 2) != 0;
  if (!g_vsx_nt_fast || p->N <= 64) return false;
  if (p->c_mode != VSX_A_ROWS && !(ext && p->c_mode == VSX_A_PATCH2 && p->a_mode == VSX_A_ROWS && p->pro != VSX_PRO_GRN && p->b_bstride == 0 &&
                                   (p->epi == VSX_EPI_NONE || p->epi == VSX_EPI_BIAS)))
    return false;
  if (p->a_mode != VSX_A_ROWS && !(ext && p->a_mode == VSX_A_PATCH2 && p->cs % 32 == 0 && p->pro != VSX_PRO_GRN && p->b_bstride == 0 &&
                                   (p->epi == VSX_EPI_NONE || p->epi == VSX_EPI_BIAS)))
    return false;
  if (p->K % 32 != 0 && !(ext && p->a_mode == VSX_A_ROWS && p->K > 32 && p->pro != VSX_PRO_GRN)) return false;
  if (p->a_mode == VSX_A_PATCH2 && (unsigned long long)8 * p->gw * p->lda * es * (128 / p->gw + 2) >= (1ull << 32)) return false;
  if (p->epi == VSX_EPI_BIAS_STATS) return false;
  const bool reduce = p->epi == VSX_EPI_BIAS_GELU_SQ || p->epi == VSX_EPI_DZ;
  if (p->b_bstride != 0 && (p->hw <= 0 || p->hw % 128 != 0)) return false;
  // reductions / the GRN prologue need every 64-row epilogue pass inside one sample: hw a multiple of 128, or exactly 64
  // (8 x 8 feature maps: two samples per tile, handled per pass)
  if ((reduce || p->pro == VSX_PRO_GRN) && (p->hw <= 0 || (p->hw % 128 != 0 && p->hw != 64))) return false;
  if ((unsigned long long)256 * p->lda * es >= (1ull << 32) || (unsigned long long)p->N * p->ldb * es >= (1ull << 32)) return false;
  return true;
}

template <typename T>
static int dispatch_nt_fast(const VsxGemm* p, hipStream_t s) {
  const bool pro = p->pro == VSX_PRO_GRN;
  switch (p->epi) {
    case VSX_EPI_NONE: return pro ? launch_nt_fast<T, VSX_EPI_NONE, true>(p, s) : launch_nt_fast<T, VSX_EPI_NONE, false>(p, s);
    case VSX_EPI_BIAS: return pro ? launch_nt_fast<T, VSX_EPI_BIAS, true>(p, s) : launch_nt_fast<T, VSX_EPI_BIAS, false>(p, s);
    case VSX_EPI_BIAS_GELU_SQ: return pro ? launch_nt_fast<T, VSX_EPI_BIAS_GELU_SQ, true>(p, s) : launch_nt_fast<T, VSX_EPI_BIAS_GELU_SQ, false>(p, s);
    case VSX_EPI_BIAS_RES: return pro ? launch_nt_fast<T, VSX_EPI_BIAS_RES, true>(p, s) : launch_nt_fast<T, VSX_EPI_BIAS_RES, false>(p, s);
    default: return pro ? launch_nt_fast<T, VSX_EPI_DZ, true>(p, s) : launch_nt_fast<T, VSX_EPI_DZ, false>(p, s);
  }
}

template <typename T, int BM, int BN, int WM_, int WN_, int BK, int NBUF = 2>
static int launch_nt(const VsxGemm* p, hipStream_t s) {
  g_vsx_last_kernel = "gemm_nt_generic";
  int tiles = vsx_cdiv(p->M, BM) * vsx_cdiv(p->N, BN);
  dim3 grid(tiles, 1, p->nz > 0 ? p->nz : 1);
  hipLaunchKernelGGL((gemm_nt_kernel<T, BM, BN, WM_, WN_, BK, NBUF>), grid, dim3(256), 0, s, *p);
  VSX_LAUNCH_CHECK();
  return 0;
}

template <typename T>
static int dispatch_nt(const VsxGemm* p, hipStream_t s) {
  if (p->b_bstride != 0) {
    if (!nt_fast_ok(p, (int)sizeof(T))) {
      vsx_set_error("vsx_gemm_nt: per-sample weights (b_bstride) need plain row operands, N > 64, K %% 32 == 0, hw %% 128 == 0");
      return 1;
    }
    return dispatch_nt_fast<T>(p, s);
  }
  if (p->N > 64) {
    // few workgroups (< 2 per CU) and a long K loop: the loop is bound by global-load latency, not by MFMA
    // or bandwidth — stage 4x more K per barrier so 4x more bytes are in flight per workgroup
    long tiles = (long)vsx_cdiv(p->M, 128) * vsx_cdiv(p->N, 128) * (p->nz > 0 ? p->nz : 1);
    if constexpr (sizeof(T) == 2) {
      if (tiles < 512 && p->K >= 256) return launch_nt<T, 128, 128, 2, 2, 128>(p, s);
    }
    if (nt_fast_ok(p, (int)sizeof(T))) return dispatch_nt_fast<T>(p, s);
    return launch_nt<T, 128, 128, 2, 2, 32>(p, s);
  }
  if (p->N > 32) return launch_nt<T, 128, 64, 2, 2, 32>(p, s);
  if (p->N > 16) return launch_nt<T, 128, 32, 4, 1, 32>(p, s);
  return launch_nt<T, 128, 16, 4, 1, 32>(p, s);
}

static int check_common(const VsxGemm* p, int dtype, const char* who) {
  int vn = dtype == VSX_BF16 ? 8 : 4;
  VSX_CHECK(p != nullptr, "%s: null params", who);
  VSX_CHECK(dtype == VSX_F32 || dtype == VSX_BF16, "%s: bad dtype %d", who, dtype);
  VSX_CHECK(p->M > 0 && p->N > 0 && p->K > 0, "%s: empty GEMM %dx%dx%d", who, p->M, p->N, p->K);
  VSX_CHECK(p->K % vn == 0 && p->N % vn == 0, "%s: N=%d and K=%d must be multiples of %d", who, p->N, p->K, vn);
  VSX_CHECK(p->lda % vn == 0 && p->ldb % vn == 0, "%s: lda=%d ldb=%d must be multiples of %d", who, p->lda, p->ldb, vn);
  VSX_CHECK(p->nz >= 0 && p->nz <= 8, "%s: nz=%d out of range", who, p->nz);
  if (p->a_mode != VSX_A_ROWS) {
    VSX_CHECK(p->gh > 0 && p->gw > 0 && p->cs > 0 && p->cs % vn == 0, "%s: bad gather grid %dx%d cs=%d", who, p->gh,
              p->gw, p->cs);
    VSX_CHECK(p->K % p->cs == 0, "%s: K=%d not a multiple of cs=%d", who, p->K, p->cs);
    VSX_CHECK(p->M % (p->gh * p->gw) == 0, "%s: M=%d not a multiple of the %dx%d grid", who, p->M, p->gh, p->gw);
  }
  if (p->pro == VSX_PRO_GRN) VSX_CHECK(p->grn_s && p->grn_b && p->hw > 0, "%s: GRN prologue needs s, beta, hw", who);
  for (int z = 0; z < (p->nz > 0 ? p->nz : 1); ++z)
    VSX_CHECK(p->a_coff[z] % vn == 0 && p->b_off[z] % vn == 0 && p->c_coff[z] % vn == 0,
              "%s: z offsets must be multiples of %d", who, vn);
  return 0;
}

extern "C" int32_t vsx_gemm_nt(const VsxGemm* p, int32_t dtype, vsx_stream_t stream) {
  if (int e = check_common(p, dtype, "vsx_gemm_nt")) return e;
  int vn = dtype == VSX_BF16 ? 8 : 4;
  VSX_CHECK(p->ldc % vn == 0, "vsx_gemm_nt: ldc=%d must be a multiple of %d", p->ldc, vn);
  if (p->c_mode == VSX_A_PATCH2)
    VSX_CHECK(p->c_cs > 0 && p->c_cs % vn == 0 && p->N % p->c_cs == 0 && p->gh > 0 && p->gw > 0,
              "vsx_gemm_nt: bad patch scatter (c_cs=%d)", p->c_cs);
  if (p->epi == VSX_EPI_BIAS_GELU_SQ || p->epi == VSX_EPI_DZ || p->epi == VSX_EPI_BIAS_STATS)
    VSX_CHECK(p->red0 != nullptr && p->hw > 0, "vsx_gemm_nt: reduction epilogue needs red0 and hw");
  if (p->epi == VSX_EPI_DZ) VSX_CHECK(p->aux && p->red1, "vsx_gemm_nt: EPI_DZ needs aux and red1");
  if (p->epi == VSX_EPI_BIAS_GELU_SQ) VSX_CHECK(p->C2 != nullptr, "vsx_gemm_nt: EPI_BIAS_GELU_SQ needs the second output C2");
  if (p->epi == VSX_EPI_BIAS_STATS) VSX_CHECK(p->red1 != nullptr, "vsx_gemm_nt: EPI_BIAS_STATS needs red1");
  if (p->epi == VSX_EPI_BIAS_RES) VSX_CHECK(p->res != nullptr, "vsx_gemm_nt: EPI_BIAS_RES needs res");
  if (p->rscale) VSX_CHECK(p->epi == VSX_EPI_BIAS_RES && p->hw > 0, "vsx_gemm_nt: rscale needs EPI_BIAS_RES and hw");
  if (p->epi == VSX_EPI_LN_BWD)
    VSX_CHECK(dtype == VSX_BF16 && p->aux && p->grn_s && p->C && vsx_gemm_nt2_lnbwd_ok(p),
              "vsx_gemm_nt: EPI_LN_BWD needs bf16 row operands, aux = xh, grn_s = rstd, N <= 256, M %% 256 == 0, K %% 32 == 0 "
              "(query vsx_gemm_nt_ln_bwd_supported)");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == VSX_BF16 && vsx_gemm_nt2_ok(p)) return vsx_gemm_nt2(p, s);  // second-generation kernel (gemm_nt2.hip)
  return dtype == VSX_BF16 ? dispatch_nt<bf16_t>(p, s) : dispatch_nt<float>(p, s);
}

extern "C" int32_t vsx_gemm_nt_ln_bwd_supported(int64_t M, int32_t N, int32_t K, int32_t dtype) {
  if (dtype != VSX_BF16 || M <= 0 || M >= (1ll << 31)) return 0;
  VsxGemm q = {};
  q.M = (int32_t)M; q.N = N; q.K = K; q.lda = K; q.ldb = K; q.ldc = N; q.a_mode = VSX_A_ROWS; q.c_mode = VSX_A_ROWS; q.epi = VSX_EPI_LN_BWD;
  return vsx_gemm_nt2_lnbwd_ok(&q) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// gemm_tn : W[n][k] += sum_m X[m][n] * Y[m][k]
// LDS tiles keep the natural [m][n] / [m][k] layout; the MFMA operands need 8 consecutive
// contraction (m) values per lane, which gfx950's ds_read_b64_tr_b16 delivers straight from a
// row-major tile (TR = true).  TR = false is the scalar-read fallback, also used for fp32.
// k-slot ↔ m mapping inside a 32-row step (bf16): slot (kq, j) ↔ m = kq*4 + j (j < 4), 16 + kq*4 + j - 4.
// ------------------------------------------------------------------------------------------------
template <bool TR>
__device__ __forceinline__ bf16x8 lds_frag_mn_bf16(const char* tile, int LDB, int col0, int p16, int kq) {
  // tile: [32][LD] bf16, LDB = row stride bytes; returns the fragment for column col0 + p16
  if (TR) {
    const int q = p16;
    const char* a0 = tile + (kq * 4 + (q >> 2)) * LDB + (col0 + (q & 3) * 4) * 2;
    const char* a1 = a0 + 16 * LDB;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a1));
    union {
      struct { s16x4 lo, hi; } s;
      bf16x8 v;
    } u;
    u.s.lo = lo;
    u.s.hi = hi;
    return u.v;
  } else {
    union {
      unsigned short h[8];
      bf16x8 v;
    } u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int mloc = j < 4 ? kq * 4 + j : 16 + kq * 4 + (j - 4);
      u.h[j] = *reinterpret_cast<const unsigned short*>(tile + mloc * LDB + (col0 + p16) * 2);
    }
    return u.v;
  }
}

template <typename T, bool TR>
__device__ __forceinline__ typename Frag<T>::type lds_frag_mn(const char* tile, int LDB, int col0, int p16, int kq,
                                                              int kk);
template <>
__device__ __forceinline__ bf16x8 lds_frag_mn<bf16_t, true>(const char* t, int LDB, int c0, int p16, int kq, int) {
  return lds_frag_mn_bf16<true>(t, LDB, c0, p16, kq);
}
template <>
__device__ __forceinline__ bf16x8 lds_frag_mn<bf16_t, false>(const char* t, int LDB, int c0, int p16, int kq, int) {
  return lds_frag_mn_bf16<false>(t, LDB, c0, p16, kq);
}
template <>
__device__ __forceinline__ float lds_frag_mn<float, false>(const char* t, int LDB, int c0, int p16, int kq, int kk) {
  return *reinterpret_cast<const float*>(t + (kk * 4 + kq) * LDB + (c0 + p16) * 4);
}
template <>
__device__ __forceinline__ float lds_frag_mn<float, true>(const char* t, int LDB, int c0, int p16, int kq, int kk) {
  return *reinterpret_cast<const float*>(t + (kk * 4 + kq) * LDB + (c0 + p16) * 4);
}

// XCD-aware order of a (tiles x splits) TN grid.  Workgroups are dealt round-robin to the 8 XCDs in launch order (x fastest); all
// output tiles of one split stream the SAME row windows of X / Y, so they belong on one XCD (one L2) and adjacent in time —
// measured with FETCH_SIZE: without this every tile re-fetched its operands through the fabric.  Whole groups of 8 splits: XCD x
// takes split 8 g + x.  The splits past the last whole group (round 5: split counts are chosen to fill rounds of workgroups, not
// to be multiples of 8) are cut into 8 runs of consecutive (split, tile) pairs, one run per XCD: a run holds tiles of one or two
// splits (in launch order they would be scattered one tile per XCD: FETCH_SIZE of the rectangular-tile class 30.4 -> 35.9 GB / step).
__device__ __forceinline__ void tn_xcd_order(int& bx, int& by) {
  const int T = gridDim.x, S = gridDim.y;
  const int L = bx + T * by;
  const int full = (S & ~7) * T;
  if (L < full) {
    const int j = L >> 3;
    bx = j % T;
    by = (j / T) * 8 + (L & 7);
  } else {
    const int rT = (S & 7) * T, Lt = L - full;
    const int x = Lt & 7, q = Lt >> 3;
    const int start = x * (rT >> 3) + (x < (rT & 7) ? x : (rT & 7));
    const int pidx = start + q;
    bx = pidx % T;
    by = (S & ~7) + pidx / T;
  }
}

template <typename T, int BT, bool TR>  // BT x BT output tile of W
__global__ __launch_bounds__(256) void gemm_tn_kernel(const VsxGemm p, int rows_per_block) {
  constexpr int BMS = 32;                       // contraction rows per step
  constexpr int ES = sizeof(T);
  constexpr int VN = VT<T>::N;
  constexpr int LD = BT + 16;                   // LDS row stride (elements)
  constexpr int LDB = LD * ES;
  constexpr int CPR = BT / VN;                  // chunks per tile row
  constexpr int NCH = (BMS * CPR + 255) / 256;  // chunks per thread per operand
  constexpr int TILE_BYTES = BMS * LDB;
  constexpr int F = BT / 2 / 16;                // frags per wave per dim (2x2 waves)
  constexpr int MK = Frag<T>::MK;
  typedef typename VT<T>::vec vec;
  typedef typename Frag<T>::type frag_t;
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int p16 = lane & 15, kq = lane >> 4;
  const int z = blockIdx.z;

  const int tiles_k = (p.K + BT - 1) / BT;
  // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs in launch order (x fastest); all output tiles of
  // one split stream the SAME 32-row windows of X / Y, so they are placed on one XCD (one L2) and adjacent in time —
  // measured with FETCH_SIZE: without this every tile re-fetched its operands through the fabric
  int bx = blockIdx.x, by = blockIdx.y;
  tn_xcd_order(bx, by);
  const int tile_k = bx % tiles_k, tile_n = bx / tiles_k;
  const int n0 = tile_n * BT, k0 = tile_k * BT;
  // the 32-row contraction steps are dealt round-robin to the gridDim.y splits: at any instant the
  // splits stream one contiguous window of X / Y (blocked ranges would camp on a few HBM channels)
  (void)rows_per_block;
  const int mend = p.M;
  const int total_steps = (p.M + BMS - 1) / BMS;
  const int nsplit = gridDim.y;
  if (by >= total_steps) return;

  const T* X = reinterpret_cast<const T*>(p.B);
  const int x_coff = p.b_off[z];
  const int a_coff = p.a_coff[z];

  int crow[NCH], cch[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int cid = tid + i * 256;
    crow[i] = cid / CPR;
    cch[i] = cid % CPR;
  }
  vec xreg[NCH], yreg[NCH];
  int yb[NCH];
  auto load_tiles = [&](int ms) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int cid = tid + i * 256;
      int m = ms + crow[i];
      bool ok = cid < BMS * CPR && m < mend;
      int nn = n0 + cch[i] * VN;
      xreg[i] = (ok && nn < p.N) ? ldvec<T>(X + (size_t)m * p.ldb + x_coff + nn) : vzero<T>();
      int kk = k0 + cch[i] * VN;
      if (ok && kk < p.K) {
        RowCoord rc = decode_row(p, m);
        yreg[i] = load_a_chunk<T>(p, rc, m, kk, a_coff);
        yb[i] = rc.b;
      } else {
        yreg[i] = vzero<T>();
        yb[i] = -1;
      }
    }
  };
  auto store_tiles = [&](int buf) {
    char* Xs = smem + buf * 2 * TILE_BYTES;
    char* Ys = Xs + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int cid = tid + i * 256;
      if (cid < BMS * CPR) {
        *reinterpret_cast<vec*>(Xs + crow[i] * LDB + cch[i] * 16) = xreg[i];
        vec yv = yreg[i];
        if (p.pro != VSX_PRO_NONE && yb[i] >= 0) yv = apply_prologue<T>(p, yv, yb[i], k0 + cch[i] * VN);
        *reinterpret_cast<vec*>(Ys + crow[i] * LDB + cch[i] * 16) = yv;
      }
    }
  };

  f32x4 acc[F][F];
#pragma unroll
  for (int i = 0; i < F; ++i)
#pragma unroll
    for (int j = 0; j < F; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float csum = 0.f;  // bias-gradient partial for column n0 + tid (tile_k == 0 blocks only)
  const bool do_colsum = p.colsum != nullptr && tile_k == 0 && tid < BT;

  const int nsteps = (total_steps - by + nsplit - 1) / nsplit;
  load_tiles(by * BMS);
  store_tiles(0);
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    if (st + 1 < nsteps) load_tiles(((st + 1) * nsplit + by) * BMS);
    const char* Xs = smem + (st & 1) * 2 * TILE_BYTES;
    const char* Ys = Xs + TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < BMS / MK; ++kk) {
      frag_t xf[F], yf[F];
#pragma unroll
      for (int i = 0; i < F; ++i) xf[i] = lds_frag_mn<T, TR>(Xs, LDB, (wn * F + i) * 16, p16, kq, kk);
#pragma unroll
      for (int j = 0; j < F; ++j) yf[j] = lds_frag_mn<T, TR>(Ys, LDB, (wk * F + j) * 16, p16, kq, kk);
#pragma unroll
      for (int i = 0; i < F; ++i)
#pragma unroll
        for (int j = 0; j < F; ++j) acc[i][j] = mfma16(xf[i], yf[j], acc[i][j]);
    }
    if (do_colsum) {
#pragma unroll 8
      for (int r = 0; r < BMS; ++r) csum += to_f32<T>(*reinterpret_cast<const T*>(Xs + r * LDB + tid * ES));
    }
    if (st + 1 < nsteps) store_tiles((st + 1) & 1);
    __syncthreads();
  }

  float* W = reinterpret_cast<float*>(p.C) + p.c_coff[z];
#pragma unroll
  for (int i = 0; i < F; ++i)
#pragma unroll
    for (int j = 0; j < F; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int n = n0 + (wn * F + i) * 16 + kq * 4 + r;
        int k = k0 + (wk * F + j) * 16 + p16;
        if (n < p.N && k < p.K) atomicAdd(W + (size_t)n * p.ldc + k, acc[i][j][r]);
      }
  if (do_colsum && n0 + tid < p.N) atomicAdd(p.colsum + n0 + tid, csum);
}

// ------------------------------------------------------------------------------------------------
// gemm_tn, lean instantiation: plain row operands, M a multiple of 32, a 32-row step inside one sample.
// Same tiles / fragment reads / split order as gemm_tn_kernel; the operand addresses are a scalar base per step
// plus a per-lane offset computed once, column tails are clamped at load time (a clamped column only feeds
// outputs that are never written), the GRN prologue is a template parameter.
// ------------------------------------------------------------------------------------------------
// BTK_ != 0: rectangular output tile BT (N) x BTK_ (K).  A 256-wide side that spans the whole C-wide dimension of a weight
// gradient (dW2: N = C, dW1: K = C, C <= 256) makes the 4C-wide activation operand stream through exactly ONE workgroup per
// pixel split instead of two (PMC: the square-tile TN launches fetched 1.7x their algorithmic bytes) and halves the fragment
// reads per MFMA (12 transposing reads per 32 MFMAs instead of 8 per 16).
// PRO: 0 = plain operands; 1 = GRN prologue on the Y operand as it is staged (y = g * s[b, k] + beta[k], rounded to bf16 again);
// 2 (round 5) = the fc2 weight gradient of a whole-sample stage TOGETHER WITH the GRN statistics of the block backward.  The
// operand is not touched: Q_b = X_b^T . g_b is accumulated per SAMPLE, and behind the last step of a sample
//   * a second accumulator set takes acc2 += s[b, k] * Q_b       (dW[n, k] = sum_b s[b, k] Q_b[n, k] + beta[k] * sum_r x[r, n]),
//   * P[b, k] += sum_n W2[n, k] * Q_b[n, k] over the tile's rows n  (= sum_hw dz * g with dz = dout . W2: dz is linear in dout, so
//     the statistics that csrc/mlp.hip MODE 3 recomputes dz for — a full M x 4C x C contraction per block — fall out of the
//     per-sample tile that sits in the accumulators anyway; the W2 tile waits in LDS in accumulator order).
// The beta term is the rank-1 product (column sums of X) x beta, added once at the end; the column sums come from one extra MFMA
// per X fragment against a ones operand.  By itself the post-scaled form is as fast as the prologue form (the launch is bound by
// its operand stream, not by the prologue: 320 us plain vs 343 us at C = 384, B = 512); what it buys is the MODE 3 launch.
// WG = 512 (round 6): eight waves as a 4 (N) x 2 (K) grid on ONE workgroup per CU — 256 x 256 and 256 x 384 output tiles.  The
// split-K launches are bound by their operand stream L2 -> LDS (~10 TB/s chip-wide, section 3 item 18): a 128 x 256 tile moves
// (128 + 256) / (128 * 256) operand elements per MAC, a 256 x 384 tile 1.8 x fewer, and an operand whose whole width fits the tile
// is read exactly once by the launch.
template <typename T, int BT, bool TR, int PRO, int BMS = 32, int NBUF = 2, int BTK_ = 0, int WG = 256>
__global__ __launch_bounds__(WG, WG == 512 ? 1 : ((BTK_ != 0 || PRO == 2) ? 2 : ((BMS == 64 && TR && sizeof(T) == 2) ? 3 : 1))) void gemm_tn_fast_kernel(const VsxGemm p) {
  constexpr int ES = sizeof(T);
  constexpr int VN = VT<T>::N;
  constexpr int BTN = BT, BTK = BTK_ != 0 ? BTK_ : BT;
  constexpr int LDBX = (BTN + 16) * ES, LDBY = (BTK + 16) * ES;
  constexpr int CPRX = BTN / VN, CPRY = BTK / VN;
  constexpr int NCHX = (BMS * CPRX + WG - 1) / WG, NCHY = (BMS * CPRY + WG - 1) / WG;
  constexpr int WN = WG / 128;  // wave rows (N) x 2 wave columns (K)
  static_assert(WG == 256 || (WG == 512 && PRO == 0), "the eight-wave geometry serves the plain weight gradients");
  constexpr int TILE_X = BMS * LDBX, TILE_Y = BMS * LDBY;
  constexpr int FN_ = BTN / WN / 16, FK_ = BTK / 2 / 16;
  constexpr int MK = Frag<T>::MK;
  typedef typename VT<T>::vec vec;
  typedef typename Frag<T>::type frag_t;
  __shared__ __attribute__((aligned(16))) char smem[NBUF * (TILE_X + TILE_Y)];
  // PRO == 2: this workgroup's W2 tile (bf16) in accumulator order — lane t finds the 4 rows r of fragment (i, j) at [(i * FK_ + j) * 256 + t]
  __shared__ __attribute__((aligned(16))) uint2 w2img[PRO == 2 ? FN_ * FK_ * 256 : 1];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int p16 = lane & 15, kq = lane >> 4;
  const int z = blockIdx.z;
  const int tiles_k = (p.K + BTK - 1) / BTK;
  int bx = blockIdx.x, by = blockIdx.y;
  tn_xcd_order(bx, by);  // XCD-aware (see gemm_tn_kernel)
  const int tile_k = bx % tiles_k, tile_n = bx / tiles_k;
  const int n0 = tile_n * BTN, k0 = tile_k * BTK;
  const int total_steps = p.M / BMS;
  const int nsplit = gridDim.y;
  if (by >= total_steps) return;

  const char* Xb = reinterpret_cast<const char*>(p.B) + (size_t)p.b_off[z] * ES;
  const char* Yb = reinterpret_cast<const char*>(p.A) + (size_t)p.a_coff[z] * ES;
  // round 6: the Y operand gathered from 2 x 2 stride-2 patches (VSX_A_PATCH2: weight gradient of the downsampling projections).
  // Row m = output pixel of the gh x gw grid; with gw | BMS (dispatch) a step of BMS rows is BMS / gw whole grid rows, the input
  // pixel of row m is 4 gw (m / gw) + 2 (m % gw) whatever the sample, so a step advances the operand by 4 BMS input pixels and
  // the lane offsets below never change
  const bool ypatch = p.a_mode == VSX_A_PATCH2;
  // ... and with BMS | gw (wide grids: the 2048 x 2048 gate shape's 256 / 128 / 64-pixel rows) a step lies inside ONE grid row:
  // lane offsets 2 * crow, the step's first input pixel from the formula above (a scalar division per step)
  const bool ywide = ypatch && p.gw > BMS;
  const size_t xstep = (size_t)BMS * p.ldb * ES, ystep = (size_t)(ypatch ? 4 * BMS : BMS) * p.lda * ES;
  uint32_t xoff[NCHX], yoff[NCHY];
  int ldsx[NCHX], ldsy[NCHY], kcol[NCHY];
  bool livex[NCHX], livey[NCHY];
#pragma unroll
  for (int i = 0; i < NCHX; ++i) {
    const int cid = tid + i * WG, crow = cid / CPRX, cch = cid % CPRX;
    livex[i] = cid < BMS * CPRX;
    int nn = n0 + cch * VN;
    nn = nn < p.N ? nn : p.N - VN;
    xoff[i] = (uint32_t)(crow * p.ldb + nn) * ES;
    ldsx[i] = crow * LDBX + cch * 16;
  }
#pragma unroll
  for (int i = 0; i < NCHY; ++i) {
    const int cid = tid + i * WG, crow = cid / CPRY, cch = cid % CPRY;
    livey[i] = cid < BMS * CPRY;
    int kk = k0 + cch * VN;
    kk = kk < p.K ? kk : p.K - VN;
    kcol[i] = kk;
    if (ypatch) {
      const int tap = kk / p.cs, c = kk - tap * p.cs;
      const int pix = (ywide ? 2 * crow : (crow / p.gw) * (4 * p.gw) + 2 * (crow % p.gw)) + (tap >> 1) * (2 * p.gw) + (tap & 1);
      yoff[i] = (uint32_t)(pix * p.lda + c) * ES;
    } else
    yoff[i] = (uint32_t)(crow * p.lda + kk) * ES;
    ldsy[i] = crow * LDBY + cch * 16;
  }

  vec xreg[NCHX], yreg[NCHY];
  // An operand that ONE tile column / row covers completely is read exactly once by the launch: its loads are non-temporal, so
  // that the stream does not evict the other operand — which the other tiles of the split re-read — from the XCD's L2 (the TN
  // class hit the L2 in 50 % of its requests, profiles/r04_tcc_gemm.txt).  `tn_stream` flag (bits 13 / 14 of p.pro, set by the launcher): bit 0 = X, bit 1 = Y.
  const int tiles_n_ = (p.N + BTN - 1) / BTN;
  const bool x_once = (p.pro & 8192) && tiles_k == 1, y_once = (p.pro & 16384) && tiles_n_ == 1;
  auto load_tiles = [&](int step, vec* xr, vec* yr) {
    const char* Xs = Xb + (size_t)step * xstep;
    const char* Ys = Yb + (size_t)step * ystep;
    if (ywide) {
      const long m = (long)step * BMS;
      Ys = Yb + (size_t)(4 * p.gw * (m / p.gw) + 2 * (m % p.gw)) * (size_t)p.lda * ES;
    }
#pragma unroll
    for (int i = 0; i < NCHX; ++i)
      if (livex[i]) xr[i] = ldvec_stream<T>(reinterpret_cast<const T*>(Xs + xoff[i]), x_once);
#pragma unroll
    for (int i = 0; i < NCHY; ++i)
      if (livey[i]) yr[i] = ldvec_stream<T>(reinterpret_cast<const T*>(Ys + yoff[i]), y_once);
  };
  // GRN prologue operands: every chunk of this thread covers the SAME 8 (4) columns (256 % CPRY == 0), and the sample
  // index changes only every hw / BMS steps -> s[b, k..] and beta[k..] live in registers, reloaded on a sample change
  // (the first version re-read them from global memory for every chunk of every step)
  float gsr[VN], gbr[VN];
  int gcur = -1;
  if constexpr (PRO == 1) {
#pragma unroll
    for (int j = 0; j < VN; ++j) gbr[j] = p.grn_b[kcol[0] + j];
  }
  auto store_tiles = [&](int step, int buf, const vec* xr, const vec* yr) {
    char* Xs = smem + buf * (TILE_X + TILE_Y);
    char* Ys = Xs + TILE_X;
    if constexpr (PRO == 1) {
      const int bnow = (step * BMS) / p.hw;  // (uniform)
      if (bnow != gcur) {
        gcur = bnow;
        const float* gs = p.grn_s + (size_t)bnow * p.K + kcol[0];
#pragma unroll
        for (int j = 0; j < VN; ++j) gsr[j] = gs[j];
      }
    }
#pragma unroll
    for (int i = 0; i < NCHX; ++i) {
      if (livex[i]) {
        const vec xv = xr[i];
        *reinterpret_cast<vec*>(Xs + ldsx[i]) = xv;
      }
    }
#pragma unroll
    for (int i = 0; i < NCHY; ++i) {
      if (livey[i]) {
        vec yv = yr[i];
        if constexpr (PRO == 1) {
          float f[VN];
          unpack<T>(yv, f);
#pragma unroll
          for (int j = 0; j < VN; ++j) f[j] = fmaf(f[j], gsr[j], gbr[j]);
          yv = pack<T>(f);
        }
        *reinterpret_cast<vec*>(Ys + ldsy[i]) = yv;
      }
    }
  };

  f32x4 acc[FN_][FK_];
#pragma unroll
  for (int i = 0; i < FN_; ++i)
#pragma unroll
    for (int j = 0; j < FK_; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float csum = 0.f;
  const bool do_colsum = p.colsum != nullptr && tile_k == 0 && tid < BTN;
  f32x4 acc2[PRO == 2 ? FN_ : 1][PRO == 2 ? FK_ : 1];
  float sreg[PRO == 2 ? FK_ : 1];
  f32x4 acc1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};  // PRO == 2: column sums of X (fragments 2 wk, 2 wk + 1 of this wave's rows)
  frag_t ones_frag;
  const bool want_p = PRO == 2 && p.aux != nullptr && p.red0 != nullptr;
  if constexpr (PRO == 2) {
    union { unsigned short h[8]; frag_t v; } one;
#pragma unroll
    for (int j = 0; j < 8; ++j) one.h[j] = 0x3F80;
    ones_frag = one.v;
#pragma unroll
    for (int i = 0; i < FN_; ++i)
#pragma unroll
      for (int j = 0; j < FK_; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (want_p) {
      const unsigned short* W2 = reinterpret_cast<const unsigned short*>(p.aux);
#pragma unroll
      for (int i = 0; i < FN_; ++i)
#pragma unroll
        for (int j = 0; j < FK_; ++j) {
          const int k = k0 + (wk * FK_ + j) * 16 + p16;
          unsigned short w[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = n0 + (wn * FN_ + i) * 16 + kq * 4 + r;
            w[r] = (n < p.N && k < p.K) ? W2[(size_t)n * p.ldx + k] : (unsigned short)0;
          }
          w2img[(i * FK_ + j) * 256 + tid] = make_uint2((uint32_t)w[0] | ((uint32_t)w[1] << 16), (uint32_t)w[2] | ((uint32_t)w[3] << 16));
        }
    }
  }

  // steps of this split: interleaved with the other splits (by, by + nsplit, ...) or, with bit 10 of p.pro set by the
  // launcher, one contiguous range — the GRN prologue then reloads s[b, k..] once per hw / BMS steps instead of on every
  // step (a split stride of nsplit * BMS rows crosses a sample boundary each time on the small feature maps)
  const bool contig = (p.pro & 1024) != 0;
  // PRO == 2: a split is a range of WHOLE samples (sps steps each); the ranges may differ by one sample, so that the launcher is
  // free to pick the split count that fills its rounds of workgroups (36 tiles at C = 384 divide no power of two evenly)
  const int sps = PRO == 2 ? p.hw / BMS : 1;
  const int units = total_steps / sps;
  const int sq = units / nsplit, sr = units % nsplit;
  const int sbase = (by * sq + (by < sr ? by : sr)) * sps;
  const int nsteps = contig ? (sq + (by < sr ? 1 : 0)) * sps : (total_steps - by + nsplit - 1) / nsplit;
  auto step_of = [&](int st) { return contig ? sbase + st : st * nsplit + by; };
  auto mma_step = [&](const char* Xs, const char* Ys) {
#pragma unroll
    for (int kk = 0; kk < BMS / MK; ++kk) {
      // bf16: one MFMA contraction step = 32 tile rows (the transposing read addresses rows r, r + 16 of them)
      const char* Xk = Xs + (sizeof(T) == 2 ? kk * 32 * LDBX : 0);
      const char* Yk = Ys + (sizeof(T) == 2 ? kk * 32 * LDBY : 0);
      // (the Y fragments in groups of at most 6: with all 12 of the 384-wide tile live next to 192 accumulator registers the
      // kernel spilled 105 registers)
      constexpr int JH = FK_ > 8 ? 2 : 1, JW = FK_ / JH;
      static_assert(FK_ % JH == 0, "Y fragments split evenly");
      frag_t xf[FN_], yf[JW];
#pragma unroll
      for (int i = 0; i < FN_; ++i) xf[i] = lds_frag_mn<T, TR>(Xk, LDBX, (wn * FN_ + i) * 16, p16, kq, kk);
#pragma unroll
      for (int jh = 0; jh < JH; ++jh) {
#pragma unroll
        for (int j = 0; j < JW; ++j) yf[j] = lds_frag_mn<T, TR>(Yk, LDBY, (wk * FK_ + jh * JW + j) * 16, p16, kq, kk);
#pragma unroll
        for (int i = 0; i < FN_; ++i)
#pragma unroll
          for (int j = 0; j < JW; ++j) acc[i][jh * JW + j] = mfma16(xf[i], yf[j], acc[i][jh * JW + j]);
      }
      if constexpr (PRO == 2) {
        // column sums of X (beta term, bias gradient) on the matrix cores: X^T . 1 has them in every column; the two waves that
        // share an X row range take two fragments each (a scalar loop over the tile cost as much as the step's MFMAs)
        static_assert(PRO != 2 || FN_ == 4, "two X fragments per wave");
#pragma unroll
        for (int h = 0; h < 2; ++h) acc1[h] = mfma16(wk == 0 ? xf[h] : xf[2 + h], ones_frag, acc1[h]);
      }
    }
    if constexpr (PRO == 2) {
    } else if (do_colsum) {
#pragma unroll 8
      for (int r = 0; r < BMS; ++r) csum += to_f32<T>(*reinterpret_cast<const T*>(Xs + r * LDBX + tid * ES));
    }
  };
  // PRO == 2: what happens at the two ends of a sample
  auto sample_begin = [&](int gstep) {
    if constexpr (PRO == 2) {
      const float* gs = p.grn_s + (size_t)(gstep / sps) * p.K;
#pragma unroll
      for (int j = 0; j < FK_; ++j) {
        const int k = k0 + (wk * FK_ + j) * 16 + p16;
        sreg[j] = gs[k < p.K ? k : p.K - 1];
      }
    }
  };
  auto sample_end = [&](int gstep) {
    if constexpr (PRO == 2) {
      if (want_p) {
        float pj[FK_];
#pragma unroll
        for (int j = 0; j < FK_; ++j) {
          float a = 0.f;
#pragma unroll
          for (int i = 0; i < FN_; ++i) {
            const uint2 w = w2img[(i * FK_ + j) * 256 + tid];
            a = fmaf(acc[i][j][0], __uint_as_float(w.x << 16), a);
            a = fmaf(acc[i][j][1], __uint_as_float(w.x & 0xFFFF0000u), a);
            a = fmaf(acc[i][j][2], __uint_as_float(w.y << 16), a);
            a = fmaf(acc[i][j][3], __uint_as_float(w.y & 0xFFFF0000u), a);
          }
          a += __shfl_xor(a, 16, 64);
          a += __shfl_xor(a, 32, 64);
          pj[j] = a;
        }
        if (kq == 0) {
          float* P = p.red0 + (size_t)(gstep / sps) * p.K;
#pragma unroll
          for (int j = 0; j < FK_; ++j) {
            const int k = k0 + (wk * FK_ + j) * 16 + p16;
            if (k < p.K) atomicAdd(P + k, pj[j]);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < FN_; ++i)
#pragma unroll
        for (int j = 0; j < FK_; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc2[i][j][r] = fmaf(acc[i][j][r], sreg[j], acc2[i][j][r]);
            acc[i][j][r] = 0.f;
          }
    }
  };
  load_tiles(step_of(0), xreg, yreg);
  if constexpr (NBUF == 2) {
    store_tiles(step_of(0), 0, xreg, yreg);
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
      const int nxt = step_of(st + 1);
      if (st + 1 < nsteps) load_tiles(nxt, xreg, yreg);
      const char* Xs = smem + (st & 1) * (TILE_X + TILE_Y);
      mma_step(Xs, Xs + TILE_X);
      if (st + 1 < nsteps) store_tiles(nxt, (st + 1) & 1, xreg, yreg);
      __syncthreads();
    }
  } else {
    static_assert(PRO != 2 || NBUF == 1, "the post-scaled weight gradient lives on the single-buffer geometry");
    for (int st = 0; st < nsteps; ++st) {
      store_tiles(step_of(st), 0, xreg, yreg);
      __syncthreads();
      if (st + 1 < nsteps) load_tiles(step_of(st + 1), xreg, yreg);
      if constexpr (PRO == 2) {
        if ((sbase + st) % sps == 0) sample_begin(sbase + st);   // (the launcher aligns every split with whole samples)
      }
      mma_step(smem, smem + TILE_X);
      if constexpr (PRO == 2) {
        if ((sbase + st + 1) % sps == 0) sample_end(sbase + st);
      }
      __syncthreads();
    }
  }
  if constexpr (PRO == 2) {
    // beta term: dW[n, k] += beta[k] * (column sum of X over this workgroup's rows)[n]; the sums pass through LDS (the operand
    // tiles are dead).  The same sums are the bias gradient (tile_k == 0 workgroups only).
    float* cs = reinterpret_cast<float*>(smem);
    __syncthreads();
    if (p16 == 0) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) cs[(wn * FN_ + 2 * wk + h) * 16 + kq * 4 + r] = acc1[h][r];
    }
    __syncthreads();
    if (tid < BTN) csum = cs[tid];
#pragma unroll
    for (int j = 0; j < FK_; ++j) {
      const int k = k0 + (wk * FK_ + j) * 16 + p16;
      const float bk = p.grn_b[k < p.K ? k : p.K - 1];
#pragma unroll
      for (int i = 0; i < FN_; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaf(bk, cs[(wn * FN_ + i) * 16 + kq * 4 + r], acc2[i][j][r]);
    }
  }

  // per-sample mode (b_bstride != 0, set up by launch_tn): split `by` IS sample `by` (one contiguous step range per sample, one
  // workgroup per (sample, tile)), its product goes to its own output matrix with plain stores
  const bool per_sample = p.b_bstride != 0;
  const bool ps_atomic = (p.pro & 4096) != 0;  // several splits per sample (few samples, large maps): atomics into zeroed outputs
  const int sample = per_sample ? (int)(((long)sbase * BMS) / p.hw) : 0;
  float* W = reinterpret_cast<float*>(p.C) + p.c_coff[z] + (per_sample ? (size_t)sample * (size_t)p.b_bstride : 0);
#pragma unroll
  for (int i = 0; i < FN_; ++i)
#pragma unroll
    for (int j = 0; j < FK_; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + (wn * FN_ + i) * 16 + kq * 4 + r;
        const int k = k0 + (wk * FK_ + j) * 16 + p16;
        if (n < p.N && k < p.K) {
          if (per_sample && !ps_atomic) W[(size_t)n * p.ldc + k] = acc[i][j][r];
          else atomicAdd(W + (size_t)n * p.ldc + k, acc[i][j][r]);
        }
      }
  if (do_colsum && n0 + tid < p.N) {
    if (per_sample && !ps_atomic) p.colsum[(size_t)sample * p.N + n0 + tid] = csum;
    else atomicAdd(p.colsum + (per_sample ? (size_t)sample * p.N : 0) + n0 + tid, csum);
  }
}

// zero fill as an ordinary kernel node (the captured step stays free of memset nodes)
__global__ __launch_bounds__(256) void tn_zero_kernel(float* __restrict__ p, long n) {
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) *reinterpret_cast<float4*>(p + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  else for (; i < n; ++i) p[i] = 0.f;
}

// Split count for a split-K launch of `tiles` output tiles whose workgroups fit `per_cu` to a CU: the count in [want * 2 / 3,
// want * 4 / 3] whose tiles x splits fill their last round of 256 x per_cu workgroups best (round 5, tools/rounds.py: the
// stage-3 weight gradients ran 1.12 rounds — the last eighth of their workgroups alone on the chip — stage 2's dW1 0.84).  A tie
// goes to the count nearest `want`.  tn_fill = 0 switches it off.
static int fill_splits(int tiles, int per_cu, int want, int max_splits) {
  if (!g_vsx_tn_fill) return want;
  const int slots = 256 * per_cu;
  int lo = want * 2 / 3, hi = want * 4 / 3;
  lo = lo < 1 ? 1 : lo;
  hi = hi > max_splits ? max_splits : hi;
  int best = want < lo ? lo : (want > hi ? hi : want);
  double best_fill = -1.0;
  for (int sp = lo; sp <= hi; ++sp) {
    const long wgs = (long)tiles * sp;
    const long rounds = (wgs + slots - 1) / slots;
    const double fill = (double)wgs / (double)(rounds * slots);
    const int d = sp > want ? sp - want : want - sp, bd = best > want ? best - want : want - best;
    if (fill > best_fill + 1e-9 || (fill > best_fill - 1e-9 && d < bd)) { best_fill = fill; best = sp; }
  }
  return best;
}

template <typename T, int BT, bool TR>
static int launch_tn(const VsxGemm* p, hipStream_t s) {
  g_vsx_last_kernel = "gemm_tn_fast";  // (the generic kernel overrides this at its launch below)
  int tiles = vsx_cdiv(p->N, BT) * vsx_cdiv(p->K, BT);
  int nz = p->nz > 0 ? p->nz : 1;
  // split the pixel (contraction) axis so that the launch fills 256 CUs, but keep the number of
  // same-address atomics (= splits) small: they serialise at ~0.2 us each
  int want = vsx_cdiv(g_vsx_tn_want, tiles * nz);
  int max_splits = vsx_cdiv(p->M, 256);
  int splits = want < 1 ? 1 : (want > max_splits ? max_splits : want);
  // few tiles (skinny weight matrices, e.g. the head's 8x32): the atomics spread over few addresses anyway,
  // parallelism matters more
  int cap = tiles * nz <= 4 ? 384 : (tiles * nz <= 24 ? 96 : 48);
  if (splits > cap) splits = cap;
  int rpb = 0;
  if (splits > vsx_cdiv(p->M, 32)) splits = vsx_cdiv(p->M, 32);
  if (splits >= 8) splits &= ~7;  // multiple of 8: the kernel maps whole splits onto XCDs
  if (p->b_bstride != 0) {
    // one product PER SAMPLE: C[b] (b_bstride elements apart) = X_b^T . Y_b over the hw rows of sample b, colsum[b][N] likewise
    // (plain stores: the caller need not zero the outputs).  Used by the block backward: the per-sample products dout_b^T g_b
    // give the fc2 weight gradient AND the GRN statistics P, S without ever forming dz (vsx_grn_q_reduce).
    if constexpr (sizeof(T) == 2 && BT == 128 && TR) {
      VSX_CHECK(g_vsx_nt_fast && p->a_mode == VSX_A_ROWS && p->pro == VSX_PRO_NONE && p->hw > 0 && p->hw % 64 == 0 && p->M % p->hw == 0 &&
                    nz == 1 && (unsigned long long)64 * (p->lda > p->ldb ? p->lda : p->ldb) * sizeof(T) < (1ull << 31),
                "vsx_gemm_tn: per-sample outputs (b_bstride) need plain bf16 row operands, no prologue, hw %% 64 == 0");
      VsxGemm pq = *p;
      pq.pro |= 1024;  // contiguous step range per split: a whole sample, or 1 / ks of one
      pq.pro |= (g_vsx_tn_stream & 3) << 13;
      const int nb = p->M / p->hw;
      // N = 192 too since round 4: a quarter of the 256-wide tile idles, but the 4C-wide operand is read once instead of twice and,
      // with its loads non-temporal (tn_stream), the launch is 4 % faster than on 128 x 128 tiles (328 -> 315 us at B = 512)
      const bool n_full = p->N >= 192 && p->N <= 256 && p->K >= 128;
      const bool big = TR && (g_vsx_tn_rect & 64) && n_full && p->K >= 512;  // round 6, bit 6 (off: 1 210 -> 1 340 us at C = 224, 308 -> 320 at C = 192): eight-wave 256 x 256 tiles, one workgroup per CU
      const int t2 = big ? vsx_cdiv(p->K, 256) : (n_full ? vsx_cdiv(p->N, 256) * vsx_cdiv(p->K, 128) : tiles);
      // few samples with large maps (the 2048^2 gate shape: 8 samples of 262 144 rows): ks splits per sample so that the launch
      // still fills the chip; their partial products meet in zero-filled outputs through atomics (<= ks adds per address)
      int ks = 1;
      const int spp = p->hw / 64;  // 64-row steps per sample
      while ((long)nb * ks * t2 < 768 && ks < 64 && spp % (2 * ks) == 0) ks *= 2;
      if (g_vsx_tn_fill && ks > 1) {
        // ... and of the power-of-two counts from there up to 4 x, the one that fills its last round of workgroups best (2 per CU for
        // the 256-wide tiles, 3 for the square ones): 8 samples x 36 tiles x 4 ran 1.5 rounds at the gate shape's C = 384 blocks
        const int slots = 256 * (big ? 1 : (n_full ? 2 : 3));
        int best = ks;
        double bf = -1.0;
        for (int k2 = ks; k2 <= 4 * ks && k2 <= 64 && spp % k2 == 0; k2 *= 2) {
          const long wgs = (long)nb * k2 * t2, rounds = (wgs + slots - 1) / slots;
          const double f = (double)wgs / (double)(rounds * slots);
          if (f > bf + 1e-9) { bf = f; best = k2; }
        }
        ks = best;
      }
      if (ks > 1) {
        pq.pro |= 4096;
        const long nq = (long)nb * p->b_bstride, nc = p->colsum ? (long)nb * p->N : 0;
        hipLaunchKernelGGL(tn_zero_kernel, dim3(vsx_cdiv(nq, 1024L)), dim3(256), 0, s, reinterpret_cast<float*>(p->C) + p->c_coff[0], nq);
        if (nc) hipLaunchKernelGGL(tn_zero_kernel, dim3(vsx_cdiv(nc, 1024L)), dim3(256), 0, s, p->colsum, nc);
      }
      if (big) {
        dim3 g2(t2, nb * ks, 1);
        hipLaunchKernelGGL((gemm_tn_fast_kernel<T, 256, TR, false, 32, 2, 256, 512>), g2, dim3(512), 0, s, pq);
      } else if (n_full) {
        dim3 g2(t2, nb * ks, 1);
        hipLaunchKernelGGL((gemm_tn_fast_kernel<T, 256, TR, false, 64, 1, 128>), g2, dim3(256), 0, s, pq);
      } else {
        dim3 g2(tiles, nb * ks, 1);
        hipLaunchKernelGGL((gemm_tn_fast_kernel<T, BT, TR, false, 64, 1>), g2, dim3(256), 0, s, pq);
      }
      VSX_LAUNCH_CHECK();
      return 0;
    } else {
      vsx_set_error("vsx_gemm_tn: per-sample outputs (b_bstride) exist for bf16 operands with outputs of at least 96 x 96 only");
      return 1;
    }
  }
  if constexpr (sizeof(T) == 2 && BT == 128 && TR) {
    // GRN-prologue weight gradient + GRN backward statistics in one launch (kernel header, PRO == 2): the caller asks for it by
    // passing the bf16 fc2 weight as `aux` ([N, ldx]) and the statistics target as `red0` ([M / hw, K], zeroed); whole samples per
    // split, 64-row steps inside one sample.  (Without aux / red0 the prologue kernel below is as fast: nothing to gain.)
    if (p->pro == VSX_PRO_GRN && p->aux != nullptr && p->red0 != nullptr) {
      VSX_CHECK(g_vsx_nt_fast && p->a_mode == VSX_A_ROWS && nz == 1 && p->hw > 0 && p->hw % 64 == 0 && p->M % p->hw == 0 && p->N >= 96 &&
                    p->K >= 128 && p->N % 8 == 0 && p->K % 8 == 0 && p->ldx >= p->K &&
                    (unsigned long long)64 * (p->lda > p->ldb ? p->lda : p->ldb) * sizeof(T) < (1ull << 31),
                "vsx_gemm_tn: the weight gradient with GRN statistics (aux = W2, red0 = P) needs plain bf16 row operands, hw %% 64 == 0, N >= 96, K >= 128");
      const int nb = p->M / p->hw;
      // two workgroups per CU (68 KB of LDS each): the split count that fills `tn_p2_rounds` rounds of 512 workgroups as evenly as
      // whole tiles allow (36 tiles at C = 384: 28 splits = 1008 workgroups; the power-of-two split of the first version ran 2.25
      // rounds).  tn_p2_rounds = 0: that first version (the largest divisor of the batch below 2 * tn_want / tiles).
      int sp = 1;
      if (g_vsx_tn_p2_rounds > 0) {
        sp = (g_vsx_tn_p2_rounds * 512) / tiles;
      } else {
        const int wantp = vsx_cdiv(g_vsx_tn_want * 2, tiles);
        for (int d = wantp < nb ? wantp : nb; d >= 1; --d)
          if (nb % d == 0) { sp = d; break; }
      }
      sp = sp < 1 ? 1 : (sp > nb ? nb : sp);
      VsxGemm pq = *p;
      pq.pro = 1024 | ((g_vsx_tn_stream & 3) << 13);
      dim3 g2(tiles, sp, 1);
      g_vsx_last_kernel = "gemm_tn_fast";
      hipLaunchKernelGGL((gemm_tn_fast_kernel<T, 128, TR, 2, 64, 1>), g2, dim3(256), 0, s, pq);
      VSX_LAUNCH_CHECK();
      return 0;
    }
  }
  dim3 grid(tiles, splits, nz);
  const bool patch_ok = (g_vsx_nt_fast & 2) && p->a_mode == VSX_A_PATCH2 && p->pro == VSX_PRO_NONE && p->gw > 0 && (32 % p->gw == 0 || p->gw % 64 == 0) &&
                        p->cs % VT<T>::N == 0 && (unsigned long long)(512 + 4 * p->gw) * p->lda * sizeof(T) < (1ull << 31);
  const bool fast = g_vsx_nt_fast && (p->a_mode == VSX_A_ROWS || patch_ok) && p->M % 32 == 0 && p->N >= VT<T>::N && p->K >= VT<T>::N &&
                    (p->pro == VSX_PRO_NONE || (p->pro == VSX_PRO_GRN && p->hw > 0 && p->hw % 32 == 0)) &&
                    (unsigned long long)32 * (p->lda > p->ldb ? p->lda : p->ldb) * sizeof(T) < (1ull << 31);
  if (fast) {
    VsxGemm pq = *p;
    if (g_vsx_tn_contig) pq.pro |= 1024;  // kernel-side flag bit (the prologue kind is a template parameter there)
    pq.pro |= (g_vsx_tn_stream & 3) << 13;
    if constexpr (sizeof(T) == 2 && BT == 128) {
      if (g_vsx_tn_wide && p->M % 64 == 0 && (p->pro == VSX_PRO_NONE || p->hw % 64 == 0) && p->M / 64 >= 2 * splits) {
        if (g_vsx_tn_fill && TR) {  // 128 x 128 tiles, 64-row steps, one LDS buffer: 160 - 168 registers, three workgroups per CU
          int spf = fill_splits(tiles * nz, 3, splits, p->M / 128);
          if (spf > cap) spf = cap;
          grid.y = splits = spf < 1 ? 1 : spf;
        }
        if constexpr (TR) {
          // round 6, tn_rect bit 4: eight-wave workgroups on 256 x 256 / 256 x 384 tiles (one per CU) for the plain weight gradients
          // whose K side fits (or divides into) such a tile: dW1 of the C = 192 / 224 (256 wide), C = 384 / 768 (384 wide) blocks
          if ((g_vsx_tn_rect & 16) && p->pro == VSX_PRO_NONE && p->a_mode == VSX_A_ROWS && p->N >= 512 && p->M % 32 == 0) {
            // (a 256 x 384 tile — the K side of the C = 384 / 768 gradients in one tile — needs 192 accumulator registers and spilled
            // 60 - 105: 228 -> 410 us.  Those launches take 192-wide K tiles instead: tn_rect bit 5)
            const int btk = (p->K >= 192 && p->K <= 256) ? 256 : (((g_vsx_tn_rect & 32) && p->K > 256 && p->K % 192 == 0) ? 192 : 0);
            if (btk) {
              const int t3 = vsx_cdiv(p->N, 256) * vsx_cdiv(p->K, btk);
              int sp3 = vsx_cdiv(g_vsx_tn_want3, t3 * nz);
              if (sp3 > p->M / 64) sp3 = p->M / 64;
              if (g_vsx_tn_fill) sp3 = fill_splits(t3 * nz, 1, sp3, p->M / 64);
              if (sp3 < 1) sp3 = 1;
              dim3 g3(t3, sp3, nz);
              if (btk == 192) hipLaunchKernelGGL((gemm_tn_fast_kernel<T, 256, TR, false, 32, 2, 192, 512>), g3, dim3(512), 0, s, pq);
              else hipLaunchKernelGGL((gemm_tn_fast_kernel<T, 256, TR, false, 32, 2, 256, 512>), g3, dim3(512), 0, s, pq);
              VSX_LAUNCH_CHECK();
              return 0;
            }
          }
          // rectangular tiles when one side of the weight gradient fits a single 256-wide tile (see the kernel's header)
          // measured (tools/perf_nt.py, B = 512): C = 224 -9 % (dW1) / -14 % (dW2); C = 192 +4..8 % (a quarter of the
          // 256-wide tile idles) -> only when the tile is >= 7/8 full
          // bit 1: 256x128 tiles also when 256 divides N exactly and there is no prologue (dW1 of the C = 192 / 384 / 768
          // stages: -5..-11 % on those launches); bit 2 (off): the mirrored 128x256 tiles for the GRN-prologue operand —
          // measured 1.7..2x SLOWER (twice the prologue work per workgroup at 2 workgroups per CU)
          const bool n_div = (g_vsx_tn_rect & 2) && p->N % 256 == 0 && p->K >= 128 && p->pro == VSX_PRO_NONE;
          const bool k_div = (g_vsx_tn_rect & 4) && !n_div && p->K % 256 == 0 && p->N >= 128;
          const bool n_full = ((g_vsx_tn_rect & 1) && p->N >= 224 && p->N <= 256 && p->K >= 256) || n_div;
          const bool k_full = ((g_vsx_tn_rect & 1) && !n_full && p->K >= 224 && p->K <= 256 && p->N >= 256) || (k_div && !n_full);
          if (n_full || k_full) {
            const int t2 = n_full ? vsx_cdiv(p->N, 256) * vsx_cdiv(p->K, 128) : vsx_cdiv(p->N, 128) * vsx_cdiv(p->K, 256);
            int want2 = vsx_cdiv(g_vsx_tn_want2, t2 * nz), sp2 = want2 < 1 ? 1 : (want2 > max_splits ? max_splits : want2);
            if (sp2 > p->M / 128) sp2 = p->M / 128;
            if (g_vsx_tn_fill) sp2 = fill_splits(t2 * nz, 2, sp2, p->M / 128);  // 256 registers: two workgroups per CU
            else if (sp2 >= 8) sp2 &= ~7;
            if (sp2 < 1) sp2 = 1;
            dim3 g2(t2, sp2, nz);
            if (n_full) {
              if (p->pro == VSX_PRO_GRN)
                hipLaunchKernelGGL((gemm_tn_fast_kernel<T, 256, TR, true, 64, 1, 128>), g2, dim3(256), 0, s, pq);
              else
                hipLaunchKernelGGL((gemm_tn_fast_kernel<T, 256, TR, false, 64, 1, 128>), g2, dim3(256), 0, s, pq);
            } else {
              if (p->pro == VSX_PRO_GRN)
                hipLaunchKernelGGL((gemm_tn_fast_kernel<T, 128, TR, true, 64, 1, 256>), g2, dim3(256), 0, s, pq);
              else
                hipLaunchKernelGGL((gemm_tn_fast_kernel<T, 128, TR, false, 64, 1, 256>), g2, dim3(256), 0, s, pq);
            }
            VSX_LAUNCH_CHECK();
            return 0;
          }
        }
        if (p->pro == VSX_PRO_GRN)
          hipLaunchKernelGGL((gemm_tn_fast_kernel<T, BT, TR, true, 64, 1>), grid, dim3(256), 0, s, pq);
        else
          hipLaunchKernelGGL((gemm_tn_fast_kernel<T, BT, TR, false, 64, 1>), grid, dim3(256), 0, s, pq);
        VSX_LAUNCH_CHECK();
        return 0;
      }
    }
    if (p->pro == VSX_PRO_GRN)
      hipLaunchKernelGGL((gemm_tn_fast_kernel<T, BT, TR, true>), grid, dim3(256), 0, s, pq);
    else
      hipLaunchKernelGGL((gemm_tn_fast_kernel<T, BT, TR, false>), grid, dim3(256), 0, s, pq);
    VSX_LAUNCH_CHECK();
    return 0;
  }
  g_vsx_last_kernel = "gemm_tn_generic";
  hipLaunchKernelGGL((gemm_tn_kernel<T, BT, TR>), grid, dim3(256), 0, s, *p, rpb);
  VSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t vsx_gemm_tn(const VsxGemm* p, int32_t dtype, vsx_stream_t stream) {
  if (int e = check_common(p, dtype, "vsx_gemm_tn")) return e;
  VSX_CHECK(p->epi == VSX_EPI_NONE && p->c_mode == VSX_A_ROWS, "vsx_gemm_tn: no epilogue / scatter modes");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // 128x128 tiles (4x the MFMA work per barrier of a 64x64 tile) whenever the output is at least one such tile
  // and there are enough pixel rows to split; 64x64 only for genuinely small weight matrices
  long t128 = (long)vsx_cdiv(p->N, 128) * vsx_cdiv(p->K, 128);
  bool small = (p->N < 96 || p->K < 96) || (t128 < 24 && p->M < 65536);
  if (p->b_bstride != 0) small = false;  // per-sample outputs live on the 128-wide lean instantiations
  if (p->pro == VSX_PRO_GRN && p->aux != nullptr && p->red0 != nullptr) {  // ... and so does the weight gradient with GRN statistics
    VSX_CHECK(dtype == VSX_BF16 && g_vsx_tn_tr, "vsx_gemm_tn: the weight gradient with GRN statistics (aux = W2, red0 = P) is a bf16 kernel");
    small = false;
  }
  if (dtype == VSX_BF16) {
    if (g_vsx_tn_tr) return small ? launch_tn<bf16_t, 64, true>(p, s) : launch_tn<bf16_t, 128, true>(p, s);
    return small ? launch_tn<bf16_t, 64, false>(p, s) : launch_tn<bf16_t, 128, false>(p, s);
  }
  return small ? launch_tn<float, 64, false>(p, s) : launch_tn<float, 128, false>(p, s);
}
