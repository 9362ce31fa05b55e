// VALU issue-rate probe for gfx950: v_fma_f32 vs v_pk_fma_f32 vs v_dot2c_f32_bf16 vs v_perm_b32 (ops per clock per CU).
// hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef __attribute__((ext_vector_type(2))) float f2;
constexpr int NACC = 16, ITERS = 4096;

template <int MODE>
__global__ __launch_bounds__(256) void probe(const unsigned* in, float* out) {
  unsigned a = in[threadIdx.x], b = in[threadIdx.x + 256];
  float acc[NACC];
  f2 acc2[NACC];
  unsigned pa[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) { acc[i] = (float)i; acc2[i] = f2{(float)i, 1.f}; pa[i] = a + i; }
  const float fa = __uint_as_float(a), fb = __uint_as_float(b);
  const f2 va = {fa, fb}, vb = {fb, fa};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (MODE == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(fa), "v"(fb));  // (plain C gets SLP-packed into v_pk_fma_f32)
      if (MODE == 1) acc2[i] = __builtin_elementwise_fma(va, vb, acc2[i]);
      if (MODE == 2) acc[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a), __builtin_bit_cast(bf2, b), acc[i], false);
      if (MODE == 3) pa[i] = __builtin_amdgcn_perm(pa[i], b, 0x07060100);
    }
    asm volatile("" ::: "memory");
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i] + acc2[i].x + acc2[i].y + __uint_as_float(pa[i]);
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, const unsigned* in, float* out, double per_inst) {
  const int grid = 256 * 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, in, out);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, in, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double insts = 5.0 * grid * 4 /*waves*/ * (double)NACC * ITERS;  // wave-instructions
  const double t = ms * 1e-3;
  // wave-instructions per second per SIMD (1024 SIMDs); at 2.4 GHz → cycles per wave-instruction
  const double per_simd = insts / t / 1024.0;
  printf("%-18s %8.3f ms  %.2f cycles/wave-inst @2.4GHz   %.1f T%s/s\n", name, ms / 5, 2.4e9 / per_simd,
         insts * 64 * per_inst / t / 1e12, "op");
}

int main() {
  unsigned* in; float* out;
  hipMalloc(&in, 4096); hipMalloc(&out, 256 * 8 * 256 * 4);
  hipMemset(in, 0x3f, 4096);
  run<0>("v_fma_f32", in, out, 1);
  run<1>("v_pk_fma_f32", in, out, 2);
  run<2>("v_dot2c_f32_bf16", in, out, 2);
  run<3>("v_perm_b32", in, out, 1);
  return 0;
}
