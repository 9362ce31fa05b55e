// MFMA issue rate vs. number of independent accumulator chains (v_mfma_f32_16x16x32_bf16, 2 waves per SIMD and 1 wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(8))) __bf16 b8;
template <int NC>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
  f4 acc[NC];
  for (int c = 0; c < NC; ++c) acc[c] = (f4){0, 0, 0, 0};
  b8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 64 / NC; ++r)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);   // 64 MFMAs per iteration
  }
  float s = 0;
  for (int c = 0; c < NC; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NC>
void run(float* out, int threads) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  hipLaunchKernelGGL(probe<NC>, dim3(256), dim3(threads), 0, 0, out, 10); hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<NC>, dim3(256), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int waves_per_simd = threads / 256;
  printf("%2d chains, %d wave(s) per SIMD: %7.3f ms -> %5.1f ns per MFMA per SIMD = %5.1f cycles at 2.4 GHz -> %6.0f TFLOP/s chip\n", NC, waves_per_simd, ms,
         ms * 1e6 / (iters * 64.0 * waves_per_simd), ms * 1e6 / (iters * 64.0 * waves_per_simd) * 2.4,
         256.0 * 4 * waves_per_simd * iters * 64.0 * 16384 / (ms * 1e-3) / 1e12);
}
int main() {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  for (int t : {256, 512, 1024}) { run<1>(out, t); run<2>(out, t); run<4>(out, t); run<8>(out, t); run<16>(out, t); }
  return 0;
}
