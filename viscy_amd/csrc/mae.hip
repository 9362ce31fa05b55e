// Masked-autoencoder pre-training pieces of the FCMAE path (viscy_models/unet/fcmae.py:95-141 masked_patchify /
// masked_unpatchify, cytoland/engine.py:104-125 MaskedMSELoss).
//
// Feature maps are channels-last row matrices [B*H*W, C]; the reference's boolean-index gather / zero-filled scatter of the
// unmasked tokens become ONE row-permutation kernel driven by an int32 row map:
//     dst[r, :] = map[r] >= 0 ? src[map[r], :] (+ add[r, :]) : 0
//   gather  (patchify)   : map = dense row of compact row r                (all >= 0)
//   scatter (unpatchify) : map = compact row of dense row r, -1 if masked  (zero fill in the same pass: no memset)
//   mask    (x *= unmasked): map[r] = r or -1
// HBM-bound: one thread moves one 16-byte vector, lanes run along the contiguous channel axis (coalesced on both sides; a
// row is C*esize >= 64 contiguous bytes).
#include "vsx_common.h"
#include "../../include/vsx.h"

template <int VB>  // bytes per thread: 16, 8, 4 or 2
struct RawVec;
template <>
struct RawVec<16> { typedef uint4 t; };
template <>
struct RawVec<8> { typedef uint2 t; };
template <>
struct RawVec<4> { typedef uint32_t t; };
template <>
struct RawVec<2> { typedef uint16_t t; };

template <typename T>
__device__ __forceinline__ void add_words(uint32_t* v, const uint32_t* a, int nw);
template <>
__device__ __forceinline__ void add_words<float>(uint32_t* v, const uint32_t* a, int nw) {
  for (int i = 0; i < nw; ++i) v[i] = __float_as_uint(__uint_as_float(v[i]) + __uint_as_float(a[i]));
}
template <>
__device__ __forceinline__ void add_words<bf16_t>(uint32_t* v, const uint32_t* a, int nw) {
  for (int i = 0; i < nw; ++i) {
    const float lo = bf16_bits_to_f32(v[i] & 0xffffu) + bf16_bits_to_f32(a[i] & 0xffffu);
    const float hi = bf16_bits_to_f32(v[i] >> 16) + bf16_bits_to_f32(a[i] >> 16);
    v[i] = f32x2_to_bf16x2_bits(lo, hi);
  }
}

template <typename T, int VB>
__global__ __launch_bounds__(256) void rows_select_kernel(const void* __restrict__ src_, const int* __restrict__ map,
                                                          const void* __restrict__ add_, void* __restrict__ dst_, long n_out,
                                                          int vec_per_row) {
  typedef typename RawVec<VB>::t vec;
  const vec* __restrict__ src = (const vec*)src_;
  const vec* __restrict__ add = (const vec*)add_;
  vec* __restrict__ dst = (vec*)dst_;
  const long total = n_out * vec_per_row;
  for (long gid = (long)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (long)gridDim.x * 256) {
    const long r = gid / vec_per_row;
    const int c = (int)(gid - r * vec_per_row);
    const int m = map[r];
    vec v;
    if (m >= 0) {
      v = src[(long)m * vec_per_row + c];
      if constexpr (VB >= 4) {
        if (add) {
          vec a = add[gid];
          add_words<T>((uint32_t*)&v, (const uint32_t*)&a, VB / 4);
        }
      }
    } else {
      __builtin_memset(&v, 0, sizeof(v));
    }
    dst[gid] = v;
  }
}

template <typename T>
static int32_t rows_select_launch(const void* src, const int* map, const void* add, void* dst, long n_out, int C,
                                  hipStream_t stream) {
  const int rb = C * (int)sizeof(T);
  const long cap = 256L * 64 * 8;  // grid-stride beyond 8 waves' worth of blocks per CU
#define VSX_RS(VB)                                                                                                  \
  {                                                                                                                 \
    const int vpr = rb / VB;                                                                                        \
    long nb = (n_out * vpr + 255) / 256;                                                                            \
    if (nb > cap) nb = cap;                                                                                         \
    hipLaunchKernelGGL((rows_select_kernel<T, VB>), dim3((unsigned)nb), dim3(256), 0, stream, src, map, add, dst, n_out, vpr); \
  }
  if (rb % 16 == 0) VSX_RS(16)
  else if (rb % 8 == 0) VSX_RS(8)
  else if (rb % 4 == 0) VSX_RS(4)
  else {
    VSX_CHECK(add == nullptr, "vsx_rows_select: the fused add needs rows that are a multiple of 4 bytes (C=%d)", C);
    VSX_RS(2)
  }
#undef VSX_RS
  VSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t vsx_rows_select(const void* src, const int32_t* map, const void* add, void* dst, int64_t n_out, int32_t C,
                                   int32_t dtype, vsx_stream_t stream) {
  VSX_CHECK(src && map && dst && n_out > 0 && C > 0, "vsx_rows_select: bad arguments");
  VSX_CHECK(src != dst, "vsx_rows_select: in-place permutation is not supported");
  if (dtype == VSX_BF16) return rows_select_launch<bf16_t>(src, map, add, dst, n_out, C, (hipStream_t)stream);
  return rows_select_launch<float>(src, map, add, dst, n_out, C, (hipStream_t)stream);
}

// ------------------------------------------------------------------ MaskedMSELoss (engine.py:104-125)
//   loss = sum_{b,c,y,x} mask[b,y,x] * mean_z (p - o)^2 / sum(mask)          (mask: (B,1,H,W), 1 = masked = reconstructed)
// acc[0] += sum mask*(p-o)^2 (over all z), acc[1] += sum(mask)   -- <= 2048 same-address atomics per launch
__global__ __launch_bounds__(256) void masked_mse_sum_kernel(const float* __restrict__ P, const float* __restrict__ O,
                                                            const uint8_t* __restrict__ mask, float* __restrict__ acc, int CZ,
                                                            long HW, long total4) {
  float s = 0.f, cnt = 0.f;
  const long hw4 = HW / 4;
  for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < total4; g += (long)gridDim.x * 256) {
    const long plane = g / hw4;  // (b, c, z) plane
    const long p4 = g - plane * hw4;
    const long b = plane / CZ;
    const uint32_t m = *(const uint32_t*)(mask + b * HW + p4 * 4);
    if (m == 0) continue;
    const float4 p = ((const float4*)P)[g], o = ((const float4*)O)[g];
    const float d0 = p.x - o.x, d1 = p.y - o.y, d2 = p.z - o.z, d3 = p.w - o.w;
    const float m0 = (m & 0xffu) ? 1.f : 0.f, m1 = (m & 0xff00u) ? 1.f : 0.f, m2 = (m & 0xff0000u) ? 1.f : 0.f,
                m3 = (m & 0xff000000u) ? 1.f : 0.f;
    s += m0 * d0 * d0 + m1 * d1 * d1 + m2 * d2 * d2 + m3 * d3 * d3;
    if (plane - b * CZ == 0) cnt += m0 + m1 + m2 + m3;
  }
  s = wave_sum(s);
  cnt = wave_sum(cnt);
  __shared__ float red[2][4];
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[0][wv] = s;
    red[1][wv] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float ts = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const float tc = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    if (ts != 0.f) atomicAdd(acc, ts);
    if (tc != 0.f) atomicAdd(acc + 1, tc);
  }
}

__global__ void masked_mse_finalize_kernel(const float* __restrict__ acc, float* __restrict__ loss, float inv_z) {
  loss[0] = acc[0] * inv_z / acc[1];
}

// dP = gout * 2 (p - o) mask / (Z * sum(mask))
__global__ __launch_bounds__(256) void masked_mse_bwd_kernel(const float* __restrict__ P, const float* __restrict__ O,
                                                            const uint8_t* __restrict__ mask, const float* __restrict__ acc,
                                                            const float* __restrict__ gout, float* __restrict__ dP, int CZ,
                                                            long HW, long total4, float inv_z) {
  const float k = 2.f * inv_z * gout[0] / acc[1];
  const long hw4 = HW / 4;
  for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < total4; g += (long)gridDim.x * 256) {
    const long plane = g / hw4;
    const long p4 = g - plane * hw4;
    const long b = plane / CZ;
    const uint32_t m = *(const uint32_t*)(mask + b * HW + p4 * 4);
    float4 r = {0.f, 0.f, 0.f, 0.f};
    if (m != 0) {
      const float4 p = ((const float4*)P)[g], o = ((const float4*)O)[g];
      r.x = (m & 0xffu) ? k * (p.x - o.x) : 0.f;
      r.y = (m & 0xff00u) ? k * (p.y - o.y) : 0.f;
      r.z = (m & 0xff0000u) ? k * (p.z - o.z) : 0.f;
      r.w = (m & 0xff000000u) ? k * (p.w - o.w) : 0.f;
    }
    ((float4*)dP)[g] = r;
  }
}

extern "C" int32_t vsx_masked_mse_fwd(const float* pred, const float* orig, const uint8_t* mask, float* acc, float* loss,
                                      int32_t B, int32_t C, int32_t Z, int64_t HW, vsx_stream_t stream) {
  VSX_CHECK(pred && orig && mask && acc && loss && B > 0 && C > 0 && Z > 0 && HW > 0, "vsx_masked_mse_fwd: bad arguments");
  VSX_CHECK(HW % 4 == 0, "vsx_masked_mse_fwd: H*W=%ld must be a multiple of 4", (long)HW);
  const long total4 = (long)B * C * Z * (HW / 4);
  long nb = (total4 + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipMemsetAsync(acc, 0, 2 * sizeof(float), (hipStream_t)stream);
  hipLaunchKernelGGL(masked_mse_sum_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, pred, orig, mask, acc, C * Z,
                     (long)HW, total4);
  hipLaunchKernelGGL(masked_mse_finalize_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, acc, loss, 1.f / (float)Z);
  VSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t vsx_masked_mse_bwd(const float* pred, const float* orig, const uint8_t* mask, const float* acc,
                                      const float* gout, float* dpred, int32_t B, int32_t C, int32_t Z, int64_t HW,
                                      vsx_stream_t stream) {
  VSX_CHECK(pred && orig && mask && acc && gout && dpred && B > 0 && C > 0 && Z > 0 && HW > 0, "vsx_masked_mse_bwd: bad arguments");
  VSX_CHECK(HW % 4 == 0, "vsx_masked_mse_bwd: H*W=%ld must be a multiple of 4", (long)HW);
  const long total4 = (long)B * C * Z * (HW / 4);
  long nb = (total4 + 255) / 256;
  if (nb > 256L * 64) nb = 256L * 64;
  hipLaunchKernelGGL(masked_mse_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, pred, orig, mask, acc, gout,
                     dpred, C * Z, (long)HW, total4, 1.f / (float)Z);
  VSX_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ per-sample row scaling (stochastic-depth backward)
template <typename T>
__global__ __launch_bounds__(256) void scale_rows_samples_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                                 T* __restrict__ out, long M, int cv, int hw) {
  constexpr int VN = VT<T>::N;
  const long total = M * cv;
  for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long)gridDim.x * 256) {
    const long m = g / cv;
    const float s = scale[m / hw];
    float f[VN];
    unpack<T>(ldvec<T>(x + g * VN), f);
#pragma unroll
    for (int j = 0; j < VN; ++j) f[j] *= s;
    stvec<T>(out + g * VN, pack<T>(f));
  }
}

extern "C" int32_t vsx_scale_rows_samples(const void* x, const float* scale, void* out, int64_t M, int32_t C, int32_t hw,
                                          int32_t dtype, vsx_stream_t stream) {
  const int vn = dtype == VSX_BF16 ? 8 : 4;
  VSX_CHECK(x && scale && out && M > 0 && C > 0 && hw > 0 && C % vn == 0, "vsx_scale_rows_samples: bad arguments (C=%d)", C);
  const long total = (long)M * (C / vn);
  long nb = (total + 255) / 256;
  if (nb > 256L * 64) nb = 256L * 64;
  if (dtype == VSX_BF16)
    hipLaunchKernelGGL(scale_rows_samples_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, scale,
                       (bf16_t*)out, (long)M, C / vn, hw);
  else
    hipLaunchKernelGGL(scale_rows_samples_kernel<float>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const float*)x, scale,
                       (float*)out, (long)M, C / vn, hw);
  VSX_LAUNCH_CHECK();
  return 0;
}
