// viscy_amd — shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
// wave = 64 lanes, 16-byte vector global/LDS accesses, fp32 accumulation everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

typedef __bf16 bf16_t;

#define VSX_F32 0
#define VSX_BF16 1

// ------------------------------------------------------------------ error plumbing (host)
void vsx_set_error(const char* fmt, ...);
extern int g_vsx_head_bps;                   // api.hip: workgroups per sample of head backward pass 1 (0 = sized from the batch)
extern int g_vsx_head_rows;                  // api.hip: row-tiled MFMA passes of the head tail (head.hip)
extern int g_vsx_det_reduce;                 // api.hip: fixed-order forward sums (vsx_set_flag("det_reduce", 1))
extern thread_local float* g_vsx_det_ws;     // api.hip: vsx_det_workspace
extern thread_local long g_vsx_det_ws_floats;
// out[g * N + n] += sum over r < rows_per_group, IN ORDER, of ws[(g * rows_per_group + r) * ld + col0 + n]   (norm.hip)
int vsx_det_group_sum(const float* ws, int ld, int col0, float* out, int groups, int rows_per_group, int N, hipStream_t s);
extern thread_local const char* g_vsx_last_kernel;  // api.hip: set by the GEMM dispatchers, read by vsx_last_kernel()
#define VSX_CHECK(cond, ...)            \
  do {                                  \
    if (!(cond)) {                      \
      vsx_set_error(__VA_ARGS__);       \
      return 1;                         \
    }                                   \
  } while (0)
#define VSX_LAUNCH_CHECK()                                               \
  do {                                                                   \
    hipError_t e_ = hipGetLastError();                                   \
    if (e_ != hipSuccess) {                                              \
      vsx_set_error("%s:%d launch: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return 2;                                                          \
    }                                                                    \
  } while (0)

static inline int vsx_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ------------------------------------------------------------------ element traits
template <typename T>
struct VT;
template <>
struct VT<float> {
  static constexpr int N = 4;  // elements per 16-byte vector
  typedef float4 vec;
};
template <>
struct VT<bf16_t> {
  static constexpr int N = 8;
  typedef uint4 vec;
};

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t hi16) { return __uint_as_float(hi16 << 16); }
// gfx950 converts in hardware (v_cvt_pk_bf16_f32: round-to-nearest-even, two values per instruction); the software
// sequence it replaces cost ~9 VALU ops per element and made the GEMM epilogues issue-bound (PMC: 1443 VALU / wave)
typedef __bf16 vsx_bf16x2 __attribute__((ext_vector_type(2)));
typedef float vsx_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f32x2_to_bf16x2_bits(float lo, float hi) {
  vsx_f32x2 f = {lo, hi};
  vsx_bf16x2 h = __builtin_convertvector(f, vsx_bf16x2);
  return *reinterpret_cast<uint32_t*>(&h);
}
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  __bf16 h = (__bf16)f;
  return (uint32_t) * reinterpret_cast<uint16_t*>(&h);
}
__device__ __forceinline__ float round_bf16(float f) { return bf16_bits_to_f32(f32_to_bf16_bits(f)); }

template <typename T>
__device__ __forceinline__ float to_f32(T v);
template <>
__device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) {
  return bf16_bits_to_f32((uint32_t) * reinterpret_cast<const uint16_t*>(&v));
}
template <typename T>
__device__ __forceinline__ T from_f32(float f);
template <>
__device__ __forceinline__ float from_f32<float>(float f) { return f; }
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float f) {
  uint16_t b = (uint16_t)f32_to_bf16_bits(f);
  return *reinterpret_cast<bf16_t*>(&b);
}
// value after a round trip through storage type T (identity for fp32)
template <typename T>
__device__ __forceinline__ float round_to(float f);
template <>
__device__ __forceinline__ float round_to<float>(float f) { return f; }
template <>
__device__ __forceinline__ float round_to<bf16_t>(float f) { return round_bf16(f); }

template <typename T>
__device__ __forceinline__ void unpack(const typename VT<T>::vec& v, float* f);
template <>
__device__ __forceinline__ void unpack<float>(const float4& v, float* f) {
  f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
}
template <>
__device__ __forceinline__ void unpack<bf16_t>(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
template <typename T>
__device__ __forceinline__ typename VT<T>::vec pack(const float* f);
template <>
__device__ __forceinline__ float4 pack<float>(const float* f) { return make_float4(f[0], f[1], f[2], f[3]); }
template <>
__device__ __forceinline__ uint4 pack<bf16_t>(const float* f) {
  uint4 v;
  v.x = f32x2_to_bf16x2_bits(f[0], f[1]);
  v.y = f32x2_to_bf16x2_bits(f[2], f[3]);
  v.z = f32x2_to_bf16x2_bits(f[4], f[5]);
  v.w = f32x2_to_bf16x2_bits(f[6], f[7]);
  return v;
}
template <typename T>
__device__ __forceinline__ typename VT<T>::vec vzero();
template <>
__device__ __forceinline__ float4 vzero<float>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <>
__device__ __forceinline__ uint4 vzero<bf16_t>() { return make_uint4(0u, 0u, 0u, 0u); }

template <typename T>
__device__ __forceinline__ typename VT<T>::vec ldvec(const T* p) {
  return *reinterpret_cast<const typename VT<T>::vec*>(p);
}
template <typename T>
__device__ __forceinline__ void stvec(T* p, const typename VT<T>::vec& v) {
  *reinterpret_cast<typename VT<T>::vec*>(p) = v;
}

// 16-byte store that streams past the caches when `nt` is set (outputs a later launch reads, never this one).
// Compiler builtin, NOT inline assembly: round 1 issued `global_store_dwordx4 ... nt` from an asm statement, which hipcc
// treats as one opaque instruction — it neither counts the store in its s_waitcnt bookkeeping nor pads the wait states a
// 16-byte store needs before its data registers may be rewritten (the store reads its four data VGPRs over several
// cycles; hipcc's next instruction could overwrite them).  That is a "rare garbage in a few lanes" hazard by construction
// and the only thing the streaming-access paths had that the default paths did not; __builtin_nontemporal_store emits the
// same instruction with the compiler's own hazard handling.
typedef uint32_t vsx_u32x4 __attribute__((ext_vector_type(4)));
typedef float vsx_f32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stvec_stream(bf16_t* p, const uint4& v, bool nt) {
  const vsx_u32x4 w = {v.x, v.y, v.z, v.w};
  if (nt) __builtin_nontemporal_store(w, reinterpret_cast<vsx_u32x4*>(p));
  else *reinterpret_cast<vsx_u32x4*>(p) = w;
}
__device__ __forceinline__ void stvec_stream(float* p, const float4& v, bool nt) {
  const vsx_f32x4s w = {v.x, v.y, v.z, v.w};
  if (nt) __builtin_nontemporal_store(w, reinterpret_cast<vsx_f32x4s*>(p));
  else *reinterpret_cast<vsx_f32x4s*>(p) = w;
}

// 16-byte load of a line this launch is the last reader of (`nt`: global_load_dwordx4 ... nt)
template <typename T>
__device__ __forceinline__ typename VT<T>::vec ldvec_stream(const T* p, bool nt) {
  if (!nt) return ldvec<T>(p);
  const vsx_u32x4 w = __builtin_nontemporal_load(reinterpret_cast<const vsx_u32x4*>(p));
  typename VT<T>::vec v;
  __builtin_memcpy(&v, &w, 16);
  return v;
}

// ------------------------------------------------------------------ math
// exact-erf GELU (nn.GELU default) and its derivative.  erf via Abramowitz-Stegun 7.1.26
// (|abs err| <= 1.5e-7, below fp32 round-off of the surrounding arithmetic): one v_rcp, one v_exp
// and a 5-term Horner chain instead of the ~50-instruction branchy libm erff; exp(-x^2/2) is shared
// between the erf tail and the Gaussian pdf of the derivative.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float e = __expf(-z * z);  // = exp(-x^2 / 2)
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));  // v_rcp_f32 (1 ulp), no IEEE division sequence
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float erf_abs = 1.0f - p * t * e;
  const float erf_v = copysignf(erf_abs, x);
  cdf = 0.5f * (1.0f + erf_v);
  pdf = 0.3989422804014327f * e;
}
// two elements at a time: gfx950 issues v_pk_{mul,add,fma}_f32 at the scalar-op rate, so the polynomial / combination part
// of the GELU costs half as many VALU cycles per element (the exp and rcp stay one per element).  The elementwise passes
// that evaluate it for every 4C-wide activation (fc1 epilogue, GRN+GELU backward) are VALU-bound, not HBM-bound, without
// this: ~30 -> ~18 issue cycles per element.
typedef float vsx_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_parts2(vsx_v2f x, vsx_v2f& cdf, vsx_v2f& pdf) {
  const vsx_v2f ax = {fabsf(x.x), fabsf(x.y)};
  const vsx_v2f z = ax * 0.70710678118654752f;
  const vsx_v2f a = -(z * z) * 1.4426950408889634f;  // exp(-z^2) = 2^(-z^2 log2 e)
  const vsx_v2f e = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
  const vsx_v2f den = z * 0.3275911f + 1.0f;
  const vsx_v2f t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
  vsx_v2f p = t * 1.061405429f + (-1.453152027f);
  p = p * t + 1.421413741f;
  p = p * t + (-0.284496736f);
  p = p * t + 0.254829592f;
  const vsx_v2f ea = 1.0f - p * t * e;
  const vsx_v2f ev = {copysignf(ea.x, x.x), copysignf(ea.y, x.y)};
  cdf = ev * 0.5f + 0.5f;
  pdf = e * 0.3989422804014327f;
}

__device__ __forceinline__ float gelu_f(float x) {
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  return x * cdf;
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  return cdf + x * pdf;
}
__device__ __forceinline__ void gelu_both(float x, float& g, float& dg) {
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  g = x * cdf;
  dg = cdf + x * pdf;
}

// ------------------------------------------------------------------ wave helpers (wave64)
// Butterfly sums.  The four steps inside a 16-lane row are DPP-modified VALU adds (quad_perm x2, row_half_mirror,
// row_mirror: no LDS, no wait); only the row-crossing steps (xor 16 / 32) go through ds_bpermute.  The all-__shfl_xor
// version was 6 ds_bpermute + 6 s_waitcnt per sum — head_out_bwd1 issued 408 of them per thread.
template <int CTRL>
__device__ __forceinline__ float dpp_xadd(float v) {
  const int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false);
  return v + __int_as_float(r);
}
template <int G>
__device__ __forceinline__ float group_sum(float v) {  // sum over aligned groups of G lanes (G pow2 <= 64), result in every lane
  if (G >= 2) v = dpp_xadd<0xB1>(v);    // quad_perm [1,0,3,2]  (xor 1)
  if (G >= 4) v = dpp_xadd<0x4E>(v);    // quad_perm [2,3,0,1]  (xor 2)
  if (G >= 8) v = dpp_xadd<0x141>(v);   // row_half_mirror: lane i <-> 7 - i of each 8 (quads already hold their totals)
  if (G >= 16) v = dpp_xadd<0x140>(v);  // row_mirror: lane i <-> 15 - i
  if (G >= 32) v += __shfl_xor(v, 16, 64);
  if (G >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }
// wave total delivered to lane 63 only, without LDS traffic: the two row-crossing steps are the gfx9 DPP row broadcasts
// (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) instead of two ds_bpermute + s_waitcnt.  For
// reductions whose result one lane accumulates (head_out_bwd1: 64 of them per voxel).
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v = group_sum<16>(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));  // row_bcast:15
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));  // row_bcast:31
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
