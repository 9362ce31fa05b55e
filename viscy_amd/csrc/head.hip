// PixelToVoxelHead tail: InstanceNorm3d + PReLU + 1x1x1 Conv3d + pixel
// shuffle into the (B, C, Z, Y, X) output stack, forward and backward.  The 3x3x3 convolution
// itself runs on the MFMA GEMM (gemm.hip, VSX_A_CONV3) whose epilogue accumulates the
// per-(b, channel) sum / sum-of-squares this file turns into InstanceNorm statistics.
//
// Layout: U[b, y, x, z, o] (channels-last, o = Cmid contiguous); one thread per voxel (b, y, x, z),
// lanes along x·z so both the 16-byte U reads and the 8-byte stores into the X-contiguous output
// rows are coalesced.
#include "vsx_common.h"
#include "../../include/vsx.h"

#define HEAD_MAX_CMID 64
#define HEAD_MAX_CO4 16

struct HeadDims {
  int B, H2, W2, Z, Cmid, Cout;  // Cout = output channels (Cout*4 rows of the 1x1x1 conv)
};

template <typename T, int CMID>
__device__ __forceinline__ void head_norm_act(const T* __restrict__ u, const float* __restrict__ mu,
                                              const float* __restrict__ rs, float alpha, float* nh, float* a) {
  constexpr int VN = VT<T>::N;
#pragma unroll
  for (int c = 0; c < CMID; c += VN) {
    float v[VN];
    unpack<T>(ldvec<T>(u + c), v);
#pragma unroll
    for (int j = 0; j < VN; ++j) {
      float n = (v[j] - mu[c + j]) * rs[c + j];
      nh[c + j] = n;
      a[c + j] = n > 0.f ? n : alpha * n;
    }
  }
}

// stats → mean / rstd in LDS for sample b
__device__ __forceinline__ void head_load_stats(const float* __restrict__ ssum, const float* __restrict__ ssq, int b,
                                                int Cmid, float count, float eps, float* mu, float* rs) {
  for (int i = threadIdx.x; i < Cmid; i += blockDim.x) {
    float m = ssum[b * Cmid + i] / count;
    float var = ssq[b * Cmid + i] / count - m * m;
    mu[i] = m;
    rs[i] = rsqrtf(fmaxf(var, 0.f) + eps);
  }
}

template <typename T, int CMID, int CO4>
__global__ __launch_bounds__(256) void head_out_fwd_kernel(const T* __restrict__ U, const float* __restrict__ ssum,
                                                           const float* __restrict__ ssq, const float* __restrict__ w2,
                                                           const float* __restrict__ b2, const float* __restrict__ alpha_p,
                                                           float* __restrict__ out, HeadDims d, float eps) {
  __shared__ float mu[HEAD_MAX_CMID], rs[HEAD_MAX_CMID], w2s[HEAD_MAX_CO4 * HEAD_MAX_CMID], b2s[HEAD_MAX_CO4];
  const int b = blockIdx.y;
  constexpr int co4 = CO4;
  head_load_stats(ssum, ssq, b, CMID, (float)d.Z * d.H2 * d.W2, eps, mu, rs);
  for (int i = threadIdx.x; i < co4 * CMID; i += 256) w2s[i] = w2[i];
  if (threadIdx.x < co4) b2s[threadIdx.x] = b2[threadIdx.x];
  __syncthreads();
  const float alpha = alpha_p[0];
  const int nvox = d.H2 * d.W2 * d.Z;
  const int vox = blockIdx.x * 256 + threadIdx.x;  // (y, x, z) with z fastest
  if (vox >= nvox) return;
  const int z = vox % d.Z;
  const int px = vox / d.Z;
  const int x = px % d.W2, y = px / d.W2;
  float nh[CMID], a[CMID];
  head_norm_act<T, CMID>(U + ((size_t)b * nvox + vox) * CMID, mu, rs, alpha, nh, a);
  const int H = 2 * d.H2, W = 2 * d.W2;
  _Pragma("unroll") for (int co = 0; co < CO4 / 4; ++co) {
    float v[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float acc = b2s[co * 4 + s];
      _Pragma("unroll") for (int c = 0; c < CMID; ++c) acc = fmaf(w2s[(co * 4 + s) * CMID + c], a[c], acc);
      v[s] = acc;
    }
    float* o = out + ((((size_t)b * d.Cout + co) * d.Z + z) * H + 2 * y) * W + 2 * x;
    *reinterpret_cast<float2*>(o) = make_float2(v[0], v[1]);
    *reinterpret_cast<float2*>(o + W) = make_float2(v[2], v[3]);
  }
}

// backward pass 1: per voxel recompute n̂ / a, gather dv from dout, dA = W2^T dv, dn = dA * prelu'(n̂);
// writes a (T) and dv (T) for the 1x1x1 weight-gradient GEMM; accumulates S1 = Σ dn, S2 = Σ dn·n̂ per
// (b, channel) and the PReLU-slope gradient.
template <typename T, int CMID, int CO4>
__global__ __launch_bounds__(256) void head_out_bwd1_kernel(const T* __restrict__ U, const float* __restrict__ ssum,
                                                            const float* __restrict__ ssq, const float* __restrict__ w2,
                                                            const float* __restrict__ alpha_p,
                                                            const float* __restrict__ dout, T* __restrict__ act,
                                                            T* __restrict__ dvout, float* __restrict__ S1,
                                                            float* __restrict__ S2, float* __restrict__ dalpha,
                                                            HeadDims d, float eps, int vox_per_thread) {
  constexpr int VN = VT<T>::N;
  __shared__ float mu[HEAD_MAX_CMID], rs[HEAD_MAX_CMID], w2s[HEAD_MAX_CO4 * HEAD_MAX_CMID];
  __shared__ float part[4][2 * HEAD_MAX_CMID + 1];  // per-wave partials (no LDS atomics: plain stores, fixed order)
  __shared__ typename VT<T>::vec stage[CMID / VN][256];
  const int b = blockIdx.y;
  constexpr int co4 = CO4;
  head_load_stats(ssum, ssq, b, CMID, (float)d.Z * d.H2 * d.W2, eps, mu, rs);
  for (int i = threadIdx.x; i < co4 * CMID; i += 256) w2s[i] = w2[i];
  __syncthreads();
  const float alpha = alpha_p[0];
  const int nvox = d.H2 * d.W2 * d.Z;
  const int H = 2 * d.H2, W = 2 * d.W2;
  const int wv = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * (2 * HEAD_MAX_CMID + 1); i += 256) (&part[0][0])[i] = 0.f;
  __syncthreads();
  // Channels are processed 8 at a time and every partial sum is reduced across the wave immediately: keeping
  // 2 x CMID running sums per thread (the first version) cost 256 VGPRs + 230 AGPRs = one wave per SIMD, 2.07 ms.
  for (int it = 0; it < vox_per_thread; ++it) {
    const int vox = (it * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
    const bool live = vox < nvox;
    const int vv = live ? vox : nvox - 1;
    const int z = vv % d.Z;
    const int px = vv / d.Z;
    const int x = px % d.W2, y = px / d.W2;
    const size_t row = (size_t)b * nvox + vv;
    float dv[CO4];
    _Pragma("unroll") for (int co = 0; co < CO4 / 4; ++co) {
      const float* o = dout + ((((size_t)b * d.Cout + co) * d.Z + z) * H + 2 * y) * W + 2 * x;
      float2 t0 = *reinterpret_cast<const float2*>(o);
      float2 t1 = *reinterpret_cast<const float2*>(o + W);
      dv[co * 4 + 0] = round_to<T>(t0.x); dv[co * 4 + 1] = round_to<T>(t0.y);
      dv[co * 4 + 2] = round_to<T>(t1.x); dv[co * 4 + 3] = round_to<T>(t1.y);
    }
    if (live) {
      _Pragma("unroll") for (int k = 0; k < CO4; k += VN) stvec<T>(dvout + row * co4 + k, pack<T>(dv + k));
    }
    float da = 0.f;
    // the activation row is parked in this thread's LDS slots chunk by chunk and written to HBM in one go after the
    // channel loop (stores issued between the reductions reached HBM as partial lines: 2.4x the write traffic; keeping
    // the row in registers instead needs an unrolled channel loop = the 486-register version again)
    _Pragma("unroll 1") for (int c0 = 0; c0 < CMID; c0 += 8) {
      float u[8], nh[8], a[8];
      _Pragma("unroll") for (int c = 0; c < 8; c += VN) unpack<T>(ldvec<T>(U + row * CMID + c0 + c), u + c);
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {
        nh[j] = (u[j] - mu[c0 + j]) * rs[c0 + j];
        a[j] = nh[j] > 0.f ? nh[j] : alpha * nh[j];
      }
      _Pragma("unroll") for (int c = 0; c < 8; c += VN) stage[(c0 + c) / VN][threadIdx.x] = pack<T>(a + c);
      float dA[8];
      _Pragma("unroll") for (int j = 0; j < 8; ++j) dA[j] = 0.f;
      _Pragma("unroll") for (int k = 0; k < CO4; ++k) {
        const float4 wa = *reinterpret_cast<const float4*>(w2s + k * CMID + c0);  // same address in every lane: LDS broadcast
        const float4 wb = *reinterpret_cast<const float4*>(w2s + k * CMID + c0 + 4);
        dA[0] = fmaf(wa.x, dv[k], dA[0]); dA[1] = fmaf(wa.y, dv[k], dA[1]);
        dA[2] = fmaf(wa.z, dv[k], dA[2]); dA[3] = fmaf(wa.w, dv[k], dA[3]);
        dA[4] = fmaf(wb.x, dv[k], dA[4]); dA[5] = fmaf(wb.y, dv[k], dA[5]);
        dA[6] = fmaf(wb.z, dv[k], dA[6]); dA[7] = fmaf(wb.w, dv[k], dA[7]);
      }
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {
        float dn = nh[j] > 0.f ? dA[j] : alpha * dA[j];
        float dnn = dn * nh[j];
        if (nh[j] <= 0.f) da += dA[j] * nh[j];
        if (!live) { dn = 0.f; dnn = 0.f; }
        const float t1 = wave_sum_to_lane63(dn), t2 = wave_sum_to_lane63(dnn);
        if ((threadIdx.x & 63) == 63) {
          part[wv][c0 + j] += t1;
          part[wv][CMID + c0 + j] += t2;
        }
      }
    }
    if (live) {
      _Pragma("unroll") for (int c = 0; c < CMID / VN; ++c) stvec<T>(act + row * CMID + c * VN, stage[c][threadIdx.x]);
    }
    if (!live) da = 0.f;
    da = wave_sum_to_lane63(da);
    if ((threadIdx.x & 63) == 63) part[wv][2 * CMID] += da;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * CMID + 1; i += 256) {
    const float v = (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]);
    float* dst = i < CMID ? S1 + b * CMID + i : (i < 2 * CMID ? S2 + b * CMID + (i - CMID) : dalpha);
    atomicAdd(dst, v);
  }
}

// backward pass 1 with the 1x1x1 weight gradient folded in (bf16).  The weight gradient dW2[k][c] = Σ_vox dv[k]·a[c] is a
// GEMM whose contraction runs over the voxels, i.e. over the LANES of this kernel (one thread per voxel): each wave parks its
// 64 activation rows and dv rows transposed in LDS ([channel][voxel], 2-byte stores, 128 contiguous bytes per instruction),
// reads them back as 16x16x32 MFMA fragments (8 consecutive voxels of one channel per lane) and keeps dW2 (+ the bias
// gradient through a ones column) in MFMA accumulators across the voxel loop.  The activation tensor ([M5, Cmid]: 2.7 GB per
// step at B = 512) is never written and the separate skinny TN GEMM over it (M = 41.9 M, N = 8, K = 32: 2.3 ms at
// 1.5 TB/s) disappears.  Per-sample partial sums go to dwp[b][CO4*CMID + CO4] with a few atomics per address.
typedef __attribute__((ext_vector_type(4))) float head_f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 head_bf16x8;

template <int CMID, int CO4>
__global__ __launch_bounds__(256) void head_out_bwd1_wgrad_kernel(const bf16_t* __restrict__ U, const float* __restrict__ ssum,
                                                                  const float* __restrict__ ssq, const float* __restrict__ w2,
                                                                  const float* __restrict__ alpha_p,
                                                                  const float* __restrict__ dout, bf16_t* __restrict__ dvout,
                                                                  float* __restrict__ S1, float* __restrict__ S2,
                                                                  float* __restrict__ dalpha, float* __restrict__ dwp,
                                                                  HeadDims d, float eps, int vox_per_thread) {
  typedef bf16_t T;
  constexpr int VN = 8;
  constexpr int NF = CMID / 16;  // 16-channel MFMA column fragments
  constexpr int ROW = 64 + 8;    // one wave's voxels + 16 bytes of padding (elements)
  constexpr int NW = CO4 * CMID + CO4;
  __shared__ float mu[HEAD_MAX_CMID], rs[HEAD_MAX_CMID], w2s[HEAD_MAX_CO4 * HEAD_MAX_CMID];
  __shared__ float part[4][2 * HEAD_MAX_CMID + 1];
  __shared__ __attribute__((aligned(16))) uint16_t aT[4][CMID][ROW];
  __shared__ __attribute__((aligned(16))) uint16_t dT[4][CO4][ROW];
  static_assert(sizeof(float) * 4 * CO4 * (CMID + 1) <= sizeof(uint16_t) * 4 * CMID * ROW, "wred must fit in aT");
  float (*wred)[CO4][CMID + 1] = reinterpret_cast<float (*)[CO4][CMID + 1]>(&aT[0][0][0]);  // after the voxel loop
  const int b = blockIdx.y;
  constexpr int co4 = CO4;
  head_load_stats(ssum, ssq, b, CMID, (float)d.Z * d.H2 * d.W2, eps, mu, rs);
  for (int i = threadIdx.x; i < co4 * CMID; i += 256) w2s[i] = w2[i];
  for (int i = threadIdx.x; i < 4 * (2 * HEAD_MAX_CMID + 1); i += 256) (&part[0][0])[i] = 0.f;
  __syncthreads();
  const float alpha = alpha_p[0];
  const int nvox = d.H2 * d.W2 * d.Z;
  const int H = 2 * d.H2, W = 2 * d.W2;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int p16 = lane & 15, kq = lane >> 4;
  head_f32x4 acc[NF + 1];
#pragma unroll
  for (int f = 0; f <= NF; ++f) acc[f] = (head_f32x4){0.f, 0.f, 0.f, 0.f};
  const uint32_t one2 = p16 == 0 ? 0x3F803F80u : 0u;  // bf16 1.0 pairs: column 0 of the extra fragment sums dv (bias gradient)
  const uint4 ones_bits = make_uint4(one2, one2, one2, one2);
  const head_bf16x8 ones = *reinterpret_cast<const head_bf16x8*>(&ones_bits);
  // per-lane partial sums of S1 = sum dn, S2 = sum dn * n^ and of the PReLU-slope gradient over this thread's voxels: reduced across
  // the wave ONCE after the voxel loop (the first version reduced every channel of every voxel row: 64 wave reductions of 6
  // dependent DPP operations per 64 voxels — 40 % of the kernel's VALU work)
  float s1a[CMID], s2a[CMID], daa = 0.f;
#pragma unroll
  for (int c = 0; c < CMID; ++c) { s1a[c] = 0.f; s2a[c] = 0.f; }
  for (int it = 0; it < vox_per_thread; ++it) {
    const int vox = (it * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
    const bool live = vox < nvox;
    const int vv = live ? vox : nvox - 1;
    const int z = vv % d.Z;
    const int px = vv / d.Z;
    const int x = px % d.W2, y = px / d.W2;
    const size_t row = (size_t)b * nvox + vv;
    float dv[CO4];
    _Pragma("unroll") for (int co = 0; co < CO4 / 4; ++co) {
      const float* o = dout + ((((size_t)b * d.Cout + co) * d.Z + z) * H + 2 * y) * W + 2 * x;
      float2 t0 = *reinterpret_cast<const float2*>(o);
      float2 t1 = *reinterpret_cast<const float2*>(o + W);
      dv[co * 4 + 0] = round_to<T>(t0.x); dv[co * 4 + 1] = round_to<T>(t0.y);
      dv[co * 4 + 2] = round_to<T>(t1.x); dv[co * 4 + 3] = round_to<T>(t1.y);
    }
    if (live) {
      _Pragma("unroll") for (int k = 0; k < CO4; k += VN) stvec<T>(dvout + row * co4 + k, pack<T>(dv + k));
    }
    // dead lanes contribute zero rows to the voxel contraction
    _Pragma("unroll") for (int k = 0; k < CO4; ++k) dT[wv][k][lane] = live ? (uint16_t)(__float_as_uint(dv[k]) >> 16) : (uint16_t)0;
    float da = 0.f;
    _Pragma("unroll") for (int c0 = 0; c0 < CMID; c0 += 8) {
      float u[8], nh[8], a[8];
      unpack<T>(ldvec<T>(U + row * CMID + c0), u);
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {
        nh[j] = (u[j] - mu[c0 + j]) * rs[c0 + j];
        a[j] = nh[j] > 0.f ? nh[j] : alpha * nh[j];
        aT[wv][c0 + j][lane] = (uint16_t)f32_to_bf16_bits(a[j]);
      }
      float dA[8];
      _Pragma("unroll") for (int j = 0; j < 8; ++j) dA[j] = 0.f;
      _Pragma("unroll") for (int k = 0; k < CO4; ++k) {
        const float4 wa = *reinterpret_cast<const float4*>(w2s + k * CMID + c0);  // same address in every lane: LDS broadcast
        const float4 wb = *reinterpret_cast<const float4*>(w2s + k * CMID + c0 + 4);
        dA[0] = fmaf(wa.x, dv[k], dA[0]); dA[1] = fmaf(wa.y, dv[k], dA[1]);
        dA[2] = fmaf(wa.z, dv[k], dA[2]); dA[3] = fmaf(wa.w, dv[k], dA[3]);
        dA[4] = fmaf(wb.x, dv[k], dA[4]); dA[5] = fmaf(wb.y, dv[k], dA[5]);
        dA[6] = fmaf(wb.z, dv[k], dA[6]); dA[7] = fmaf(wb.w, dv[k], dA[7]);
      }
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {
        float dn = nh[j] > 0.f ? dA[j] : alpha * dA[j];
        float dnn = dn * nh[j];
        if (nh[j] <= 0.f) da += dA[j] * nh[j];
        if (live) {
          s1a[c0 + j] += dn;
          s2a[c0 + j] += dnn;
        }
      }
    }
    if (live) daa += da;
    __syncthreads();  // the wave's transposed rows are complete (block-wide barrier: trip counts are uniform)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      head_bf16x8 af = *reinterpret_cast<const head_bf16x8*>(&dT[wv][p16 < CO4 ? p16 : 0][kb * 32 + kq * 8]);
      if (p16 >= CO4) af = (head_bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const head_bf16x8 bf = *reinterpret_cast<const head_bf16x8*>(&aT[wv][f * 16 + p16][kb * 32 + kq * 8]);
        acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[f], 0, 0, 0);
      }
      acc[NF] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, ones, acc[NF], 0, 0, 0);
    }
    __syncthreads();  // fragments read before the next iteration overwrites the rows
  }
#pragma unroll
  for (int c = 0; c < CMID; ++c) {
    const float t1 = wave_sum_to_lane63(s1a[c]), t2 = wave_sum_to_lane63(s2a[c]);
    if (lane == 63) {
      part[wv][c] = t1;
      part[wv][CMID + c] = t2;
    }
  }
  daa = wave_sum_to_lane63(daa);
  if (lane == 63) part[wv][2 * CMID] = daa;
  // accumulator lane (p16, kq), register r holds D[m = kq*4 + r][n = p16]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = kq * 4 + r;
    if (m < CO4) {
#pragma unroll
      for (int f = 0; f < NF; ++f) wred[wv][m][f * 16 + p16] = acc[f][r];
      if (p16 == 0) wred[wv][m][CMID] = acc[NF][r];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * CMID + 1; i += 256) {
    const float v = (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]);
    float* dst = i < CMID ? S1 + b * CMID + i : (i < 2 * CMID ? S2 + b * CMID + (i - CMID) : dalpha);
    atomicAdd(dst, v);
  }
  for (int i = threadIdx.x; i < CO4 * (CMID + 1); i += 256) {
    const int m = i / (CMID + 1), n = i - m * (CMID + 1);
    const float v = (wred[0][m][n] + wred[1][m][n]) + (wred[2][m][n] + wred[3][m][n]);
    atomicAdd(dwp + (size_t)b * NW + (n < CMID ? m * CMID + n : CO4 * CMID + m), v);
  }
}

__global__ __launch_bounds__(256) void head_zero_kernel(float* __restrict__ p, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

// dW[i] += Σ_r dwp[r][i] (i < NWgt), db[i - NWgt] += ... for the remaining columns.  Block = 64 columns x 4 row slots over a
// 64-row slab (same structure as norm.hip's reduce_rows; kept local to this TU)
__global__ __launch_bounds__(256) void head_wgrad_reduce_kernel(const float* __restrict__ dwp, float* __restrict__ dW,
                                                                float* __restrict__ db, int R, int NWgt, int NB) {
  __shared__ float red[4][64];
  const int N = NWgt + NB;
  const int nl = threadIdx.x & 63, slot = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + nl;
  const int r0 = blockIdx.y * 64;
  const int r1 = r0 + 64 < R ? r0 + 64 : R;
  float a = 0.f;
  if (n < N) {
    for (int r = r0 + slot; r < r1; r += 4) a += dwp[(size_t)r * N + n];
  }
  red[slot][nl] = a;
  __syncthreads();
  if (slot == 0 && n < N) atomicAdd(n < NWgt ? dW + n : db + (n - NWgt), red[0][nl] + red[1][nl] + red[2][nl] + red[3][nl]);
}

// backward pass 2: dU = rstd * (dn - S1/cnt - n̂ * S2/cnt)
template <typename T, int CMID, int CO4>
__global__ __launch_bounds__(256) void head_out_bwd2_kernel(const T* __restrict__ U, const float* __restrict__ ssum,
                                                            const float* __restrict__ ssq, const float* __restrict__ w2,
                                                            const float* __restrict__ alpha_p,
                                                            const T* __restrict__ dvin, const float* __restrict__ S1,
                                                            const float* __restrict__ S2, T* __restrict__ dU,
                                                            HeadDims d, float eps) {
  constexpr int VN = VT<T>::N;
  __shared__ float mu[HEAD_MAX_CMID], rs[HEAD_MAX_CMID], w2s[HEAD_MAX_CO4 * HEAD_MAX_CMID];
  __shared__ float m1[HEAD_MAX_CMID], m2[HEAD_MAX_CMID];
  const int b = blockIdx.y;
  constexpr int co4 = CO4;
  const float cnt = (float)d.Z * d.H2 * d.W2;
  head_load_stats(ssum, ssq, b, CMID, cnt, eps, mu, rs);
  for (int i = threadIdx.x; i < co4 * CMID; i += 256) w2s[i] = w2[i];
  for (int i = threadIdx.x; i < CMID; i += 256) {
    m1[i] = S1[b * CMID + i] / cnt;
    m2[i] = S2[b * CMID + i] / cnt;
  }
  __syncthreads();
  const float alpha = alpha_p[0];
  const int nvox = d.H2 * d.W2 * d.Z;
  const int vox = blockIdx.x * 256 + threadIdx.x;
  if (vox >= nvox) return;
  const size_t row = (size_t)b * nvox + vox;
  float dv[CO4];
  _Pragma("unroll") for (int k = 0; k < CO4; k += VN) unpack<T>(ldvec<T>(dvin + row * co4 + k), dv + k);
  _Pragma("unroll") for (int c = 0; c < CMID; c += VN) {
    float v[VN], o[VN];
    unpack<T>(ldvec<T>(U + row * CMID + c), v);
#pragma unroll
    for (int j = 0; j < VN; ++j) {
      float n = (v[j] - mu[c + j]) * rs[c + j];
      float dA = 0.f;
      _Pragma("unroll") for (int k = 0; k < CO4; ++k) dA = fmaf(w2s[k * CMID + c + j], dv[k], dA);
      float dn = n > 0.f ? dA : alpha * dA;
      o[j] = rs[c + j] * (dn - m1[c + j] - n * m2[c + j]);
    }
    stvec<T>(dU + row * CMID + c, pack<T>(o));
  }
}

// ------------------------------------------------------------------------------------------------ round 6: row-tiled passes
// The three passes above give one thread one voxel with z fastest: a lane's 16-byte loads of its U row sit 64 bytes apart and
// a wave's 8-byte stores into the X-contiguous output rows come in 104-byte runs (13 pixels x 5 planes per wave); the 1x1x1
// contraction is 256 VALU FMAs per voxel against weights broadcast from LDS.  The row-tiled passes (bf16, 64 | W2, Z <= 8):
//   * a workgroup owns 64 consecutive x of one image row, all Z planes = Z waves, wave z <-> plane z, and 64*Z*Cmid*2 contiguous
//     bytes of U, which enter LDS with fully coalesced 16-byte loads (row pitch + 16 bytes: the fragment reads below are
//     conflict-free at Z = 5);
//   * the 1x1x1 convolution and its transpose run on the matrix cores with the VOXELS as the MFMA's N dimension: lane
//     (p16, kq) of a 16-voxel fragment holds voxel x = 16 xb + p16 and, in the accumulator, the four values m = 4 kq + r — for
//     the forward the 2 x 2 sub-pixels of output channel co = kq (two 8-byte stores per lane, 16 lanes = one full 128-byte line
//     of an output row), for the backward the channels c = 16 cb + 4 kq + r of dA = W2^T dv.  The activation enters as bf16 (what
//     autocast hands the reference's Conv3d), the weights as two bf16 fragments hi + lo (16 mantissa bits), fp32 accumulation;
//   * dv (the gathered output gradient) is laid out by the lanes that loaded it: lane (x, co) reads its 2 x 2 sub-pixels
//     (128-byte lines again) and they ARE its slice k = 4 co + s of the MFMA's K dimension — no shuffle;
//   * the voxel contraction of the weight gradient takes both operands with the transposing LDS read (ds_read_b64_tr_b16)
//     from row-major tiles (the U tile itself and the 16-byte dv rows) — the 2-byte transposing stores of the thread-per-voxel
//     pass are gone;
//   * S1 / S2 / dalpha partial sums live in the lane that owns the channel (8 per lane instead of 2 x Cmid per thread).
typedef __attribute__((ext_vector_type(4))) short head_s16x4;
constexpr int HR_TX = 64;
#ifndef HR_DBG
#define HR_DBG 0  // timing experiments (results wrong): 1 = no final atomics, 2 = no weight-gradient contraction, 4 = no per-voxel pass
#endif

template <int CMID>
struct HeadRows {
  static constexpr int RP = CMID * 2 + 16;  // LDS bytes per voxel row
  static constexpr int CPR = CMID / 8;      // 16-byte chunks per row
};

// 64*Z*CMID contiguous bf16 of U -> LDS rows (row = xl * Z + z): every thread moves exactly CPR 16-byte chunks (nt = 64 Z threads,
// nt rows).  Issue and commit are separate so that the loads of all chunks (and of the next tile, in the pass that loops over
// tiles) are in flight together: as ONE loop with a run-time trip count the compiler emitted load / s_waitcnt vmcnt(0) /
// ds_write per chunk — four exposed round trips per tile.
template <int CMID>
__device__ __forceinline__ void hr_issue(const bf16_t* __restrict__ src, int tid, int nt, vsx_u32x4* regs) {
  const vsx_u32x4* s4 = reinterpret_cast<const vsx_u32x4*>(src);
#pragma unroll
  for (int i = 0; i < HeadRows<CMID>::CPR; ++i) regs[i] = __builtin_nontemporal_load(s4 + tid + i * nt);
}
template <int CMID>
__device__ __forceinline__ void hr_commit(const vsx_u32x4* regs, unsigned char* tile, int tid, int nt) {
  typedef HeadRows<CMID> G;
#pragma unroll
  for (int i = 0; i < G::CPR; ++i) {
    const int j = tid + i * nt;
    const int r = j / G::CPR, p = j - r * G::CPR;
    *reinterpret_cast<vsx_u32x4*>(tile + r * G::RP + p * 16) = regs[i];
  }
}

__device__ __forceinline__ head_bf16x8 hr_pack8(const float* f) {
  union { uint4 u; head_bf16x8 v; } t;
  t.u = make_uint4(f32x2_to_bf16x2_bits(f[0], f[1]), f32x2_to_bf16x2_bits(f[2], f[3]), f32x2_to_bf16x2_bits(f[4], f[5]),
                   f32x2_to_bf16x2_bits(f[6], f[7]));
  return t.v;
}

// 8 voxels (contraction slots of a 32-row step, see gemm.hip lds_frag_mn_bf16) of column col0 + p16 out of a row-major LDS tile
__device__ __forceinline__ head_bf16x8 hr_tr(const unsigned char* tile, int ldb, int col0, int p16, int kq) {
  const unsigned char* a0 = tile + (kq * 4 + (p16 >> 2)) * ldb + (col0 + (p16 & 3) * 4) * 2;
  const unsigned char* a1 = a0 + 16 * ldb;
  union { struct { head_s16x4 lo, hi; } s; head_bf16x8 v; } u;
  u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((head_s16x4 __attribute__((address_space(3)))*)(a0));
  u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((head_s16x4 __attribute__((address_space(3)))*)(a1));
  return u.v;
}

template <int CMID, int CO4>
__global__ __launch_bounds__(512) void head_out_fwd_rows_kernel(const bf16_t* __restrict__ U, const float* __restrict__ ssum,
                                                                const float* __restrict__ ssq, const float* __restrict__ w2,
                                                                const float* __restrict__ b2, const float* __restrict__ alpha_p,
                                                                float* __restrict__ out, HeadDims d, float eps) {
  typedef HeadRows<CMID> G;
  constexpr int KB = CMID / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char hr_smem[];
  float* mu = reinterpret_cast<float*>(hr_smem);
  float* rs = mu + CMID;
  unsigned char* tile = hr_smem + 2 * CMID * sizeof(float);
  const int Z = d.Z, nt = 64 * Z;
  const int b = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * HR_TX;
  const int tid = threadIdx.x, lane = tid & 63, z = tid >> 6;
  const int p16 = lane & 15, kq = lane >> 4;
  vsx_u32x4 ureg[G::CPR];
  hr_issue<CMID>(U + (((size_t)b * d.H2 + y) * d.W2 + x0) * Z * CMID, tid, nt, ureg);
  head_load_stats(ssum, ssq, b, CMID, (float)d.Z * d.H2 * d.W2, eps, mu, rs);
  // A = W2 rows (m = output index), rows past CO4 zero, as TWO bf16 fragments hi + lo (lo = bf16(W2 - hi)): the weights enter with
  // 16 mantissa bits for one more MFMA per fragment on an idle pipe (bf16 weights alone moved the per-stage gradient error of the
  // bf16 engine from 0.6 - 0.9 x to 1.0 - 1.2 x the autocast yardstick of tests/test_gpu_model.py: every gradient passes here)
  head_bf16x8 wf[KB], wl[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    float t[8], tl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      t[j] = p16 < CO4 ? w2[p16 * CMID + kb * 32 + kq * 8 + j] : 0.f;
      tl[j] = t[j] - round_bf16(t[j]);
    }
    wf[kb] = hr_pack8(t);
    wl[kb] = hr_pack8(tl);
  }
  float bias[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bias[r] = kq * 4 + r < CO4 ? b2[kq * 4 + r] : 0.f;
  const float alpha = alpha_p[0];
  hr_commit<CMID>(ureg, tile, tid, nt);
  __syncthreads();
  float mc[KB][8], rc[KB][8];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      mc[kb][j] = mu[kb * 32 + kq * 8 + j];
      rc[kb][j] = rs[kb * 32 + kq * 8 + j];
    }
  const int H = 2 * d.H2, W = 2 * d.W2;
#pragma unroll
  for (int xb = 0; xb < HR_TX / 16; ++xb) {
    const unsigned char* row = tile + ((xb * 16 + p16) * Z + z) * G::RP;
    head_f32x4 acc = (head_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      float u[8];
      unpack<bf16_t>(*reinterpret_cast<const uint4*>(row + (kb * 32 + kq * 8) * 2), u);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float n = (u[j] - mc[kb][j]) * rc[kb][j];
        u[j] = n > 0.f ? n : alpha * n;
      }
      const head_bf16x8 af = hr_pack8(u);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[kb], af, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kb], af, acc, 0, 0, 0);
    }
    if (kq * 4 < CO4) {
      const int x = x0 + xb * 16 + p16;
      float* o = out + ((((size_t)b * d.Cout + kq) * d.Z + z) * H + 2 * y) * W + 2 * x;
      *reinterpret_cast<float2*>(o) = make_float2(acc[0] + bias[0], acc[1] + bias[1]);
      *reinterpret_cast<float2*>(o + W) = make_float2(acc[2] + bias[2], acc[3] + bias[3]);
    }
  }
}

// backward pass 2 on row tiles: dU = rstd * (dn - S1/cnt - n^ * S2/cnt), dn = prelu'(n^) * (W2^T dv)
template <int CMID, int CO4>
__global__ __launch_bounds__(512) void head_out_bwd2_rows_kernel(const bf16_t* __restrict__ U, const float* __restrict__ ssum,
                                                                 const float* __restrict__ ssq, const float* __restrict__ w2,
                                                                 const float* __restrict__ alpha_p,
                                                                 const bf16_t* __restrict__ dvin, const float* __restrict__ S1,
                                                                 const float* __restrict__ S2, bf16_t* __restrict__ dU,
                                                                 HeadDims d, float eps) {
  typedef HeadRows<CMID> G;
  constexpr int NCB = CMID / 16, DVR = CO4 * 2;  // channel fragments; bytes per dv row
  extern __shared__ __attribute__((aligned(16))) unsigned char hr_smem[];
  float* mu = reinterpret_cast<float*>(hr_smem);
  float* rs = mu + CMID;
  float* m1 = rs + CMID;
  float* m2 = m1 + CMID;
  unsigned char* tile = hr_smem + 4 * CMID * sizeof(float);
  const int Z = d.Z, nt = 64 * Z;
  unsigned char* dvs = tile + nt * G::RP;
  const int b = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * HR_TX;
  const int tid = threadIdx.x, lane = tid & 63, z = tid >> 6;
  const int p16 = lane & 15, kq = lane >> 4;
  const float cnt = (float)d.Z * d.H2 * d.W2;
  const size_t vox0 = (((size_t)b * d.H2 + y) * d.W2 + x0) * Z;
  vsx_u32x4 ureg[G::CPR], dreg[CO4 / 8];
  hr_issue<CMID>(U + vox0 * CMID, tid, nt, ureg);
#pragma unroll
  for (int i = 0; i < CO4 / 8; ++i) dreg[i] = __builtin_nontemporal_load(reinterpret_cast<const vsx_u32x4*>(dvin + vox0 * CO4) + tid + i * nt);
  head_load_stats(ssum, ssq, b, CMID, cnt, eps, mu, rs);
  for (int i = tid; i < CMID; i += nt) {
    m1[i] = S1[b * CMID + i] / cnt;
    m2[i] = S2[b * CMID + i] / cnt;
  }
  // A = W2^T: lane (p16 = channel within the fragment, kq) holds W2[k = 8 kq + j][16 cb + p16], k >= CO4 zero
  // (hi + lo bf16 fragments: dv IS bf16, so dA = W2^T dv comes out as with fp32 weights — see the forward)
  head_bf16x8 wf[NCB], wl[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    float t[8], tl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      t[j] = kq * 8 + j < CO4 ? w2[(kq * 8 + j) * CMID + cb * 16 + p16] : 0.f;
      tl[j] = t[j] - round_bf16(t[j]);
    }
    wf[cb] = hr_pack8(t);
    wl[cb] = hr_pack8(tl);
  }
  const float alpha = alpha_p[0];
  hr_commit<CMID>(ureg, tile, tid, nt);
#pragma unroll
  for (int i = 0; i < CO4 / 8; ++i) reinterpret_cast<vsx_u32x4*>(dvs)[tid + i * nt] = dreg[i];
  __syncthreads();
  float mc[NCB][4], rc[NCB][4], m1c[NCB][4], m2c[NCB][4];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = cb * 16 + kq * 4 + r;
      mc[cb][r] = mu[c]; rc[cb][r] = rs[c]; m1c[cb][r] = m1[c]; m2c[cb][r] = m2[c];
    }
#pragma unroll
  for (int xb = 0; xb < HR_TX / 16; ++xb) {
    const int rloc = (xb * 16 + p16) * Z + z;
    unsigned char* row = tile + rloc * G::RP;
    union { uint4 u; head_bf16x8 v; } bv;
    bv.u = kq * 8 < CO4 ? *reinterpret_cast<const uint4*>(dvs + rloc * DVR + kq * 16) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      head_f32x4 dA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[cb], bv.v, (head_f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      dA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cb], bv.v, dA, 0, 0, 0);
      uint2* up = reinterpret_cast<uint2*>(row + (cb * 16 + kq * 4) * 2);
      const uint2 uu = *up;
      const float u[4] = {bf16_bits_to_f32(uu.x & 0xFFFFu), bf16_bits_to_f32(uu.x >> 16), bf16_bits_to_f32(uu.y & 0xFFFFu),
                          bf16_bits_to_f32(uu.y >> 16)};
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float n = (u[r] - mc[cb][r]) * rc[cb][r];
        const float dn = n > 0.f ? dA[r] : alpha * dA[r];
        o[r] = rc[cb][r] * (dn - m1c[cb][r] - n * m2c[cb][r]);
      }
      *up = make_uint2(f32x2_to_bf16x2_bits(o[0], o[1]), f32x2_to_bf16x2_bits(o[2], o[3]));
    }
  }
  __syncthreads();
  uint4* d4 = reinterpret_cast<uint4*>(dU + vox0 * CMID);
#pragma unroll
  for (int i = 0; i < G::CPR; ++i) {
    const int j = tid + i * nt;
    const int r = j / G::CPR, p = j - r * G::CPR;
    d4[j] = *reinterpret_cast<const uint4*>(tile + r * G::RP + p * 16);
  }
}

// backward pass 1 + the 1x1x1 weight / bias gradient on row tiles.  Everything this pass has to deliver is a contraction over
// the voxels of dv with a function of the normalised activation alone — dz = W2^T dv is linear in dv, so with f = prelu'(n^)
// in {1, alpha}, m = [n^ <= 0], neg = min(n^, 0), a = prelu(n^) = f n^:
//     dW2[k][c]  = sum_v dv[v][k] a[v][c]                                         (the weight gradient, as before)
//     S2[c] = sum_v dn n^ = sum_v f n^ (W2^T dv)[c] = sum_k W2[k][c] dW2[k][c]    (no work of its own)
//     S1[c] = sum_v dn    = sum_k W2[k][c] (db2[k] + (alpha - 1) M[k][c]),        M = dv^T m
//     dalpha = sum_{v,c} dA n^ [n^ <= 0] = sum_{k,c} W2[k][c] N[k][c],            N = dv^T neg
// — three MFMA contractions (a, neg and the 0 / 1 mask as bf16 B operands read straight out of the U tile with the transposing
// LDS read: lane (c, kq) holds 8 voxels of ONE channel, so mean / rstd are two scalars per lane) and a per-workgroup
// epilogue on [CO4][CMID] matrices.  The first row-tiled version formed dA per voxel and accumulated S1 / S2 / dalpha on the
// VALU as the thread-per-voxel pass does: ~100 VALU instructions per 16 voxels and wave = 1.50 ms, the same as the old
// kernel, 0.89 ms with that loop removed (profiles/r06_head_rows.txt); this form needs ~6 VALU instructions per element.
// W2 enters S1 / S2 / dalpha in fp32, as it enters pass 2's dA = W2^T dv (hi + lo bf16 fragments), whose mean S1 / cnt subtracts.
template <int CMID, int CO4>
__device__ __forceinline__ void head_out_bwd1_rows_body(const bf16_t* __restrict__ U, const float* __restrict__ ssum,
                                                        const float* __restrict__ ssq, const float* __restrict__ w2,
                                                        const float* __restrict__ alpha_p, const float* __restrict__ dout,
                                                        bf16_t* __restrict__ dvout, float* __restrict__ S1,
                                                        float* __restrict__ S2, float* __restrict__ dalpha,
                                                        float* __restrict__ dwp, HeadDims d, float eps, int tiles_per_wg) {
  typedef HeadRows<CMID> G;
  constexpr int NF = CMID / 16, DVR = CO4 * 2, COUT = CO4 / 4;
  constexpr int NW = CO4 * CMID + CO4;
  constexpr int NACC = 3 * CO4 * CMID + CO4;  // per-wave partials: A | N | M as [CO4][CMID], then db[CO4]
  extern __shared__ __attribute__((aligned(16))) unsigned char hr_smem[];
  float* mu = reinterpret_cast<float*>(hr_smem);
  float* rs = mu + CMID;
  unsigned char* tile = hr_smem + 2 * CMID * sizeof(float);
  const int Z = d.Z, nt = 64 * Z;
  unsigned char* dvs = tile + nt * G::RP;   // two buffers of nt rows
  float* red = reinterpret_cast<float*>(tile);  // epilogue scratch over the U tile (and the dv rows) once the tile loop is done
  static_assert(NACC * sizeof(float) <= 64 * (G::RP + 2 * DVR), "the partials of a wave fit in its share of the tiles");
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, z = tid >> 6;
  const int p16 = lane & 15, kq = lane >> 4;
  head_load_stats(ssum, ssq, b, CMID, (float)d.Z * d.H2 * d.W2, eps, mu, rs);
  const float alpha = alpha_p[0];
  const uint32_t one2 = p16 == 0 ? 0x3F803F80u : 0u;  // ones in column 0: the extra fragment sums dv over the voxels (bias gradient)
  union { uint4 u; head_bf16x8 v; } ones;
  ones.u = make_uint4(one2, one2, one2, one2);
  head_f32x4 accA[NF], accN[NF], accM[NF], accB = (head_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    accA[f] = (head_f32x4){0.f, 0.f, 0.f, 0.f};
    accN[f] = (head_f32x4){0.f, 0.f, 0.f, 0.f};
    accM[f] = (head_f32x4){0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  float mc[NF], rc[NF];  // this lane's channel c = 16 f + p16
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    mc[f] = mu[f * 16 + p16];
    rc[f] = rs[f * 16 + p16];
  }
  const float am1 = alpha - 1.f;
  const int txn = d.W2 / HR_TX, ntile = d.H2 * txn;
  const int H = 2 * d.H2, W = 2 * d.W2;
  const int t_begin = blockIdx.x * tiles_per_wg, t_end = min(ntile, t_begin + tiles_per_wg);
  // Software pipeline over the tiles: the U chunks and the dout sub-pixels of tile t + 1 are requested before tile t is
  // computed and land in registers meanwhile (CPR x 4 + 16 registers).
  vsx_u32x4 ureg[G::CPR];
  float dv4[HR_TX / 16][4];
  auto request = [&](int t) {
    const int y = t / txn, x0 = (t - y * txn) * HR_TX;
    hr_issue<CMID>(U + (((size_t)b * d.H2 + y) * d.W2 + x0) * Z * CMID, tid, nt, ureg);
    // this lane's slice of dv: the 2 x 2 sub-pixels of output channel kq at (x, z)
#pragma unroll
    for (int xb = 0; xb < HR_TX / 16; ++xb) {
      if (kq < COUT) {
        const float* o = dout + ((((size_t)b * d.Cout + kq) * d.Z + z) * H + 2 * y) * W + 2 * (x0 + xb * 16 + p16);
        const vsx_v2f t0 = __builtin_nontemporal_load(reinterpret_cast<const vsx_v2f*>(o));
        const vsx_v2f t1 = __builtin_nontemporal_load(reinterpret_cast<const vsx_v2f*>(o + W));
        dv4[xb][0] = t0.x; dv4[xb][1] = t0.y; dv4[xb][2] = t1.x; dv4[xb][3] = t1.y;
      } else {
        dv4[xb][0] = dv4[xb][1] = dv4[xb][2] = dv4[xb][3] = 0.f;
      }
    }
  };
  if (t_begin < t_end) request(t_begin);
  for (int t = t_begin; t < t_end; ++t) {
    const int y = t / txn, x0 = (t - y * txn) * HR_TX;
    const size_t vox0 = (((size_t)b * d.H2 + y) * d.W2 + x0) * Z;
    // (the previous tile's last readers of the U tile are behind the barrier in front of its dv copy-out)
    hr_commit<CMID>(ureg, tile, tid, nt);
    unsigned char* dvt = dvs + (t & 1) * (nt * DVR);  // dv rows are double-buffered: the previous tile's are still being copied out
    if (kq < COUT) {
#pragma unroll
      for (int xb = 0; xb < HR_TX / 16; ++xb)  // rounded as the reference's bf16 gradient
        *reinterpret_cast<uint2*>(dvt + ((xb * 16 + p16) * Z + z) * DVR + kq * 8) =
            make_uint2(f32x2_to_bf16x2_bits(dv4[xb][0], dv4[xb][1]), f32x2_to_bf16x2_bits(dv4[xb][2], dv4[xb][3]));
    }
    __syncthreads();  // U tile and dv rows complete
    if (t + 1 < t_end) request(t + 1);
    // contraction over this wave's 64 voxels (rows xl * Z + z of both tiles), two 32-row steps
#pragma unroll
    for (int kb = 0; kb < ((HR_DBG & 2) ? 0 : 2); ++kb) {
      const unsigned char* t0 = tile + ((kb * 32) * Z + z) * G::RP;
      const unsigned char* v0 = dvt + ((kb * 32) * Z + z) * DVR;
      // every lane supplies its own slot address (the transposition takes column p16's values from the reads of the lanes
      // with (q & 3) == p16 / 4, whatever their own column); lanes past CO4 read into the next row and are zeroed
      head_bf16x8 af = hr_tr(v0, Z * DVR, 0, p16, kq);
      if (p16 >= CO4) af = (head_bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        union { head_bf16x8 v; uint32_t w[4]; } raw, fa, fn, fm;
        raw.v = hr_tr(t0, Z * G::RP, f * 16, p16, kq);
        if (!(HR_DBG & 4)) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float n0 = (bf16_bits_to_f32(raw.w[i] & 0xFFFFu) - mc[f]) * rc[f];
            const float n1 = (bf16_bits_to_f32(raw.w[i] >> 16) - mc[f]) * rc[f];
            const float g0 = fminf(n0, 0.f), g1 = fminf(n1, 0.f);
            fa.w[i] = f32x2_to_bf16x2_bits(fmaf(am1, g0, n0), fmaf(am1, g1, n1));  // prelu(n^) = n^ + (alpha - 1) min(n^, 0)
            fn.w[i] = f32x2_to_bf16x2_bits(g0, g1);
            // mask [n^ <= 0] as bf16 1.0 / 0.0 (the derivative at n^ = 0 is alpha, as torch's prelu backward has it)
            fm.w[i] = (n0 > 0.f ? 0u : 0x3F80u) | (n1 > 0.f ? 0u : 0x3F800000u);
          }
        } else {
          fa = raw; fn = raw; fm = raw;
        }
        accA[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, fa.v, accA[f], 0, 0, 0);
        accN[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, fn.v, accN[f], 0, 0, 0);
        accM[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, fm.v, accM[f], 0, 0, 0);
      }
      accB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, ones.v, accB, 0, 0, 0);
    }
    __syncthreads();  // dv rows of every wave are in LDS, every wave is done with the U tile
    {
      uint4* o4 = reinterpret_cast<uint4*>(dvout + vox0 * CO4);
#pragma unroll
      for (int i = 0; i < CO4 / 8; ++i) o4[tid + i * nt] = reinterpret_cast<const uint4*>(dvt)[tid + i * nt];
    }
  }
  // ---- epilogue: per-wave partials -> LDS, over the tiles
  __syncthreads();  // every wave has copied its share of the last dv rows out
  // accumulator lane (p16, kq), register r holds D[m = kq*4 + r][n = p16]
  float* my = red + z * NACC;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = kq * 4 + r;
    if (m < CO4) {
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        my[m * CMID + f * 16 + p16] = accA[f][r];
        my[CO4 * CMID + m * CMID + f * 16 + p16] = accN[f][r];
        my[2 * CO4 * CMID + m * CMID + f * 16 + p16] = accM[f][r];
      }
      if (p16 == 0) my[3 * CO4 * CMID + m] = accB[r];
    }
  }
  __syncthreads();
  // sum over the waves, in place into wave 0's slots
  for (int i = tid; i < NACC; i += nt) {
    float v = red[i];
    for (int w = 1; w < Z; ++w) v += red[w * NACC + i];
    red[i] = v;
  }
  __syncthreads();
  if (HR_DBG & 1) return;
  const float* RA = red;
  const float* RN = red + CO4 * CMID;
  const float* RM = red + 2 * CO4 * CMID;
  const float* RB = red + 3 * CO4 * CMID;
  for (int i = tid; i < NW; i += nt) atomicAdd(dwp + (size_t)b * NW + i, i < CO4 * CMID ? RA[i] : RB[i - CO4 * CMID]);
  if (tid < CMID) {  // wave 0 (CMID <= 64): S1 / S2 of channel c, this channel's share of dalpha
    const int c = tid;
    float s1 = 0.f, s2 = 0.f, da = 0.f;
#pragma unroll
    for (int k = 0; k < CO4; ++k) {
      const float wb = w2[k * CMID + c];
      s1 = fmaf(wb, fmaf(am1, RM[k * CMID + c], RB[k]), s1);
      s2 = fmaf(wb, RA[k * CMID + c], s2);
      da = fmaf(wb, RN[k * CMID + c], da);
    }
    atomicAdd(S1 + b * CMID + c, s1);
    atomicAdd(S2 + b * CMID + c, s2);
    da = group_sum<CMID < 64 ? CMID : 64>(da);
    if (tid == 0) atomicAdd(dalpha, da);
  }
}

// The Cmid = 32 instantiation is capped at 128 registers (4 waves per SIMD = three 5-wave workgroups per CU instead of two:
// 1.20 -> 1.00 ms at B = 512; 3 spilled registers); Cmid = 64 needs 190 and keeps them.
#define HR_BWD1_ARGS                                                                                                         \
  const bf16_t *__restrict__ U, const float *__restrict__ ssum, const float *__restrict__ ssq, const float *__restrict__ w2, \
      const float *__restrict__ alpha_p, const float *__restrict__ dout, bf16_t *__restrict__ dvout, float *__restrict__ S1, \
      float *__restrict__ S2, float *__restrict__ dalpha, float *__restrict__ dwp, HeadDims d, float eps, int tiles_per_wg
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void head_out_bwd1_rows32_kernel(HR_BWD1_ARGS) {
  head_out_bwd1_rows_body<32, 8>(U, ssum, ssq, w2, alpha_p, dout, dvout, S1, S2, dalpha, dwp, d, eps, tiles_per_wg);
}
__global__ __launch_bounds__(512) void head_out_bwd1_rows64_kernel(HR_BWD1_ARGS) {
  head_out_bwd1_rows_body<64, 16>(U, ssum, ssq, w2, alpha_p, dout, dvout, S1, S2, dalpha, dwp, d, eps, tiles_per_wg);
}
#undef HR_BWD1_ARGS

static bool head_rows_ok(const HeadDims& d, int dtype, int bit) {
  return dtype == VSX_BF16 && (g_vsx_head_rows >> bit & 1) && d.W2 % HR_TX == 0 && d.Z >= 1 && d.Z <= 8 &&
         ((d.Cmid == 32 && d.Cout == 2) || (d.Cmid == 64 && d.Cout == 4)) && d.H2 <= 65535 && d.B <= 65535 &&
         4 * d.Cmid * 4 + 64 * d.Z * (d.Cmid * 2 + 16 + 16 * d.Cout) + 16 <= 65536;  // static LDS limit of a plain launch
}

#define HEAD_DISPATCH(KERNEL, TT, ...)                                                                         \
  do {                                                                                                          \
    if (Cmid == 16 && Cout == 1) hipLaunchKernelGGL((KERNEL<TT, 16, 4>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
    else if (Cmid == 32 && Cout == 2) hipLaunchKernelGGL((KERNEL<TT, 32, 8>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
    else if (Cmid == 48 && Cout == 3) hipLaunchKernelGGL((KERNEL<TT, 48, 12>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<TT, 64, 16>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);         \
  } while (0)

static int head_check(const char* who, HeadDims d, int dtype) {
  int vn = dtype == VSX_BF16 ? 8 : 4;
  VSX_CHECK(d.B > 0 && d.H2 > 0 && d.W2 > 0 && d.Z > 0, "%s: bad dims", who);
  VSX_CHECK(d.Cmid > 0 && d.Cmid <= HEAD_MAX_CMID && d.Cmid % vn == 0, "%s: Cmid=%d must be <=%d and a multiple of %d", who,
            d.Cmid, HEAD_MAX_CMID, vn);
  VSX_CHECK(d.Cout > 0 && d.Cout * 4 <= HEAD_MAX_CO4 && (d.Cout * 4) % vn == 0,
            "%s: out_channels*4=%d must be <=%d and a multiple of %d", who, d.Cout * 4, HEAD_MAX_CO4, vn);
  VSX_CHECK(d.Cmid == 16 * d.Cout, "%s: only head_expansion_ratio=4 is built (Cmid=%d, out_channels=%d)", who, d.Cmid,
            d.Cout);
  return 0;
}

/* K13 (norm + act) + K14: MONAI Convolution ADN (InstanceNorm3d eps 1e-5 → PReLU), nn.Conv3d(mid, 4*out, 1),
 * transpose + nn.PixelShuffle(2) + transpose (viscy_models/components/heads.py:617-625,638-641).
 * U: [B, H2, W2, Z, Cmid] conv output; ssum/ssq: [B, Cmid] from the conv GEMM epilogue;
 * out: (B, Cout, Z, 2*H2, 2*W2) fp32. */
extern "C" int32_t vsx_head_out_fwd(const void* U, const float* ssum, const float* ssq, const float* w2, const float* b2,
                                    const float* alpha, float* out, int32_t B, int32_t H2, int32_t W2, int32_t Z,
                                    int32_t Cmid, int32_t Cout, float eps, int32_t dtype, vsx_stream_t stream) {
  HeadDims d{B, H2, W2, Z, Cmid, Cout};
  if (int e = head_check("vsx_head_out_fwd", d, dtype)) return e;
  VSX_CHECK(U && ssum && ssq && w2 && b2 && alpha && out, "vsx_head_out_fwd: null pointer");
  if (head_rows_ok(d, dtype, 0)) {
    const size_t lds = 2 * Cmid * sizeof(float) + (size_t)64 * Z * (Cmid * 2 + 16);
    dim3 g(W2 / HR_TX, H2, B);
    if (Cmid == 32)
      hipLaunchKernelGGL((head_out_fwd_rows_kernel<32, 8>), g, dim3(64 * Z), lds, (hipStream_t)stream, (const bf16_t*)U, ssum, ssq,
                         w2, b2, alpha, out, d, eps);
    else
      hipLaunchKernelGGL((head_out_fwd_rows_kernel<64, 16>), g, dim3(64 * Z), lds, (hipStream_t)stream, (const bf16_t*)U, ssum,
                         ssq, w2, b2, alpha, out, d, eps);
    VSX_LAUNCH_CHECK();
    return 0;
  }
  dim3 grid(vsx_cdiv((long)H2 * W2 * Z, 256), B);
  if (dtype == VSX_BF16)
    HEAD_DISPATCH(head_out_fwd_kernel, bf16_t, (const bf16_t*)U, ssum, ssq,
                       w2, b2, alpha, out, d, eps);
  else
    HEAD_DISPATCH(head_out_fwd_kernel, float, (const float*)U, ssum, ssq,
                       w2, b2, alpha, out, d, eps);
  VSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t vsx_head_out_bwd1(const void* U, const float* ssum, const float* ssq, const float* w2,
                                     const float* alpha, const float* dout, void* act, void* dv, float* S1, float* S2,
                                     float* dalpha, int32_t B, int32_t H2, int32_t W2, int32_t Z, int32_t Cmid,
                                     int32_t Cout, float eps, int32_t dtype, vsx_stream_t stream) {
  HeadDims d{B, H2, W2, Z, Cmid, Cout};
  if (int e = head_check("vsx_head_out_bwd1", d, dtype)) return e;
  VSX_CHECK(U && ssum && ssq && w2 && alpha && dout && act && dv && S1 && S2 && dalpha, "vsx_head_out_bwd1: null pointer");
  long nvox = (long)H2 * W2 * Z;
  int vpt = vsx_cdiv(nvox, 256L * 512);
  if (vpt < 1) vpt = 1;
  dim3 grid(vsx_cdiv(nvox, 256L * vpt), B);
  if (dtype == VSX_BF16)
    HEAD_DISPATCH(head_out_bwd1_kernel, bf16_t, (const bf16_t*)U, ssum, ssq,
                       w2, alpha, dout, (bf16_t*)act, (bf16_t*)dv, S1, S2, dalpha, d, eps, vpt);
  else
    HEAD_DISPATCH(head_out_bwd1_kernel, float, (const float*)U, ssum, ssq,
                       w2, alpha, dout, (float*)act, (float*)dv, S1, S2, dalpha, d, eps, vpt);
  VSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t vsx_head_out_bwd1_wgrad(const void* U, const float* ssum, const float* ssq, const float* w2,
                                           const float* alpha, const float* dout, void* dv, float* S1, float* S2,
                                           float* dalpha, float* dW2, float* db2, float* scratch, int32_t B, int32_t H2,
                                           int32_t W2, int32_t Z, int32_t Cmid, int32_t Cout, float eps, int32_t dtype,
                                           vsx_stream_t stream) {
  HeadDims d{B, H2, W2, Z, Cmid, Cout};
  if (int e = head_check("vsx_head_out_bwd1_wgrad", d, dtype)) return e;
  VSX_CHECK(dtype == VSX_BF16, "vsx_head_out_bwd1_wgrad: bf16 only (fp32 uses vsx_head_out_bwd1 + vsx_gemm_tn)");
  VSX_CHECK(Cmid % 16 == 0, "vsx_head_out_bwd1_wgrad: Cmid=%d must be a multiple of 16", Cmid);
  VSX_CHECK(U && ssum && ssq && w2 && alpha && dout && dv && S1 && S2 && dalpha && dW2 && db2 && scratch,
            "vsx_head_out_bwd1_wgrad: null pointer");
  const long nvox = (long)H2 * W2 * Z;
  // enough workgroups to fill the chip, few enough per sample that the per-sample partial sums see a handful of atomics
  long bps = vsx_cdiv(8192L, (long)B);
  if (bps < 32) bps = 32;
  if (g_vsx_head_bps > 0) bps = g_vsx_head_bps;  // tests: few workgroups per sample = several tiles / voxel windows per workgroup
  long vpt = nvox / (256L * bps);
  if (vpt < 1) vpt = 1;
  dim3 grid(vsx_cdiv(nvox, 256L * vpt), B);
  const int co4 = 4 * Cout, NW = co4 * Cmid + co4;
  hipStream_t st = (hipStream_t)stream;
  // zero-fill by an ordinary kernel node (not hipMemsetAsync: keeps the captured training step free of memset nodes)
  hipLaunchKernelGGL(head_zero_kernel, dim3(vsx_cdiv((long)B * NW, 256L)), dim3(256), 0, st, scratch, (long)B * NW);
  VSX_LAUNCH_CHECK();
#define HEAD_WG(CM, C4)                                                                                                   \
  hipLaunchKernelGGL((head_out_bwd1_wgrad_kernel<CM, C4>), grid, dim3(256), 0, st, (const bf16_t*)U, ssum, ssq, w2, alpha, \
                     dout, (bf16_t*)dv, S1, S2, dalpha, scratch, d, eps, (int)vpt)
  if (head_rows_ok(d, dtype, 2)) {
    // row tiles: the same number of workgroups per sample as above, each a run of consecutive 64-pixel row pieces
    const long ntile = (long)H2 * (W2 / HR_TX);
    long tpw = ntile / bps;
    if (tpw < 1) tpw = 1;
    const size_t lds = 2 * Cmid * sizeof(float) + (size_t)64 * Z * (Cmid * 2 + 16) + 2 * (size_t)64 * Z * 8 * Cout + 16;
    dim3 g(vsx_cdiv(ntile, tpw), B);
    if (Cmid == 32)
      hipLaunchKernelGGL(head_out_bwd1_rows32_kernel, g, dim3(64 * Z), lds, st, (const bf16_t*)U, ssum, ssq, w2, alpha, dout,
                         (bf16_t*)dv, S1, S2, dalpha, scratch, d, eps, (int)tpw);
    else
      hipLaunchKernelGGL(head_out_bwd1_rows64_kernel, g, dim3(64 * Z), lds, st, (const bf16_t*)U, ssum, ssq, w2, alpha, dout,
                         (bf16_t*)dv, S1, S2, dalpha, scratch, d, eps, (int)tpw);
  }
  else if (Cmid == 32 && Cout == 2) HEAD_WG(32, 8);  // bf16 rows of dv are 16-byte vectors: 4*Cout is a multiple of 8 (head_check)
  else if (Cmid == 64 && Cout == 4) HEAD_WG(64, 16);
  else VSX_CHECK(false, "vsx_head_out_bwd1_wgrad: unsupported (Cmid=%d, out_channels=%d)", Cmid, Cout);
#undef HEAD_WG
  VSX_LAUNCH_CHECK();
  hipLaunchKernelGGL(head_wgrad_reduce_kernel, dim3(vsx_cdiv(NW, 64), vsx_cdiv(B, 64)), dim3(256), 0, st, scratch, dW2, db2, B,
                     co4 * Cmid, co4);
  VSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t vsx_head_out_bwd2(const void* U, const float* ssum, const float* ssq, const float* w2,
                                     const float* alpha, const void* dv, const float* S1, const float* S2, void* dU,
                                     int32_t B, int32_t H2, int32_t W2, int32_t Z, int32_t Cmid, int32_t Cout, float eps,
                                     int32_t dtype, vsx_stream_t stream) {
  HeadDims d{B, H2, W2, Z, Cmid, Cout};
  if (int e = head_check("vsx_head_out_bwd2", d, dtype)) return e;
  VSX_CHECK(U && ssum && ssq && w2 && alpha && dv && S1 && S2 && dU, "vsx_head_out_bwd2: null pointer");
  if (head_rows_ok(d, dtype, 1)) {
    const size_t lds = 4 * Cmid * sizeof(float) + (size_t)64 * Z * (Cmid * 2 + 16) + (size_t)64 * Z * 8 * Cout;
    dim3 g(W2 / HR_TX, H2, B);
    if (Cmid == 32)
      hipLaunchKernelGGL((head_out_bwd2_rows_kernel<32, 8>), g, dim3(64 * Z), lds, (hipStream_t)stream, (const bf16_t*)U, ssum, ssq,
                         w2, alpha, (const bf16_t*)dv, S1, S2, (bf16_t*)dU, d, eps);
    else
      hipLaunchKernelGGL((head_out_bwd2_rows_kernel<64, 16>), g, dim3(64 * Z), lds, (hipStream_t)stream, (const bf16_t*)U, ssum,
                         ssq, w2, alpha, (const bf16_t*)dv, S1, S2, (bf16_t*)dU, d, eps);
    VSX_LAUNCH_CHECK();
    return 0;
  }
  dim3 grid(vsx_cdiv((long)H2 * W2 * Z, 256), B);
  if (dtype == VSX_BF16)
    HEAD_DISPATCH(head_out_bwd2_kernel, bf16_t, (const bf16_t*)U, ssum, ssq,
                       w2, alpha, (const bf16_t*)dv, S1, S2, (bf16_t*)dU, d, eps);
  else
    HEAD_DISPATCH(head_out_bwd2_kernel, float, (const float*)U, ssum, ssq,
                       w2, alpha, (const float*)dv, S1, S2, (float*)dU, d, eps);
  VSX_LAUNCH_CHECK();
  return 0;
}
