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
  if (Cmid == 32 && Cout == 2) HEAD_WG(32, 8);  // bf16 rows of dv are 16-byte vectors: 4*Cout is a multiple of 8 (head_check)
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
