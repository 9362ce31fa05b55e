// Round 5 follow-up to mfma_valu_overlap.hip.  That probe found "one MFMA wave + one VALU wave per SIMD = the SUM of the two" —
// but hipcc had SLP-packed its fmaf chains into v_pk_fma_f32, which MI355X_MICROARCH.md ("price of one filler beside MFMAs")
// lists as an anti-lever next to matrix instructions.  This probe pins the instruction with inline asm and asks two questions:
//   A. do the MFMAs of one wave overlap with the VALU instructions of the OTHER wave on the same SIMD, per VALU opcode?
//   B. inside ONE wave, how many VALU instructions of each kind fit in the shadow of a v_mfma_f32_16x16x32_bf16?
// One 512-thread workgroup per CU (2 waves per SIMD: w and w + 4), 256 workgroups, cycles at 2.4 GHz per loop iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(8))) __bf16 b8;
typedef __attribute__((ext_vector_type(2))) float f2;

#define VOP_FMA 0
#define VOP_PKFMA 1
#define VOP_PKADD 2
#define VOP_CVT 3
#define VOP_AND 4
#define VOP_PKMUL 5
#define VOP_MUL 6
#define VOP_BFI 7

template <int OP>
__device__ __forceinline__ void vop(float& a, float& b, f2& p) {
  if constexpr (OP == VOP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  else if constexpr (OP == VOP_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p));
  else if constexpr (OP == VOP_PKADD) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p));
  else if constexpr (OP == VOP_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a) : "v"(b));
  else if constexpr (OP == VOP_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b));
  else if constexpr (OP == VOP_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p));
  else if constexpr (OP == VOP_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
  else asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(a) : "v"(b));
}

// mode 0: all waves 64 MFMAs; mode 1: all waves NV VALU; mode 2: waves 0-3 MFMA, 4-7 VALU; mode 3: every wave 64 x (1 MFMA + K VALU)
template <int OP, int K>
__global__ __launch_bounds__(512) void probe(float* out, int mode, int iters) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  b8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
  float v[8], w = 1.0001f;
  f2 pv[8];
  for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 0.001f + i; pv[i] = (f2){v[i], v[i] + 1.f}; }
  const bool do_mfma = mode == 0 || (mode == 2 && wave < 4);
  const bool do_valu = mode == 1 || (mode == 2 && wave >= 4);
  if (mode == 3) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b));
#pragma unroll
          for (int k = 0; k < K; ++k) vop<OP>(v[(c * K + k) & 7], w, pv[(c * K + k) & 7]);
        }
    }
  } else if (do_mfma) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b));
    }
  } else if (do_valu) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 32; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) vop<OP>(v[i], w, pv[i]);
    }
  }
  float s = 0;
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  for (int i = 0; i < 8; ++i) s += v[i] + pv[i].x + pv[i].y;
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int OP, int K>
float run(float* out, int mode, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<OP, K>), dim3(256), dim3(512), 0, 0, out, mode, 10); hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<OP, K>), dim3(256), dim3(512), 0, 0, out, mode, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3f * 2.4e9f / iters;
}

template <int OP>
void report(float* out, const char* name) {
  const int iters = 2000;
  const float m0 = run<OP, 1>(out, 0, iters), m1 = run<OP, 1>(out, 1, iters), m2 = run<OP, 1>(out, 2, iters);
  printf("%-18s | all-MFMA (2x64) %6.0f | all-VALU (2x256) %6.0f = %.2f cyc/instr | 1 MFMA wave + 1 VALU wave per SIMD %6.0f  (max %.0f, sum %.0f)\n",
         name, m0, m1, m1 / 512.f, m2, m0 / 2 > m1 / 2 ? m0 / 2 : m1 / 2, m0 / 2 + m1 / 2);
  const float k1 = run<OP, 1>(out, 3, iters), k2 = run<OP, 2>(out, 3, iters), k3 = run<OP, 3>(out, 3, iters), k4 = run<OP, 4>(out, 3, iters),
              k6 = run<OP, 6>(out, 3, iters), k8 = run<OP, 8>(out, 3, iters);
  printf("%-18s | same wave, 2 waves/SIMD, 64 x (1 MFMA + K VALU): K=1 %6.0f  K=2 %6.0f  K=3 %6.0f  K=4 %6.0f  K=6 %6.0f  K=8 %6.0f   (MFMA alone %.0f)\n",
         name, k1, k2, k3, k4, k6, k8, m0);
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  report<VOP_FMA>(out, "v_fma_f32");
  report<VOP_MUL>(out, "v_mul_f32");
  report<VOP_PKFMA>(out, "v_pk_fma_f32");
  report<VOP_PKADD>(out, "v_pk_add_f32");
  report<VOP_PKMUL>(out, "v_pk_mul_f32");
  report<VOP_CVT>(out, "v_cvt_pk_bf16_f32");
  report<VOP_AND>(out, "v_and_b32");
  report<VOP_BFI>(out, "v_bfi_b32");
  return 0;
}
