// Do MFMA and VALU instructions of DIFFERENT waves on one SIMD overlap on gfx950?  One workgroup of 512 threads per CU (2 waves per
// SIMD: waves w and w + 4).  mode 0: every wave runs 64 MFMAs per iteration; mode 1: every wave runs 256 VALU FMAs per iteration;
// mode 2: waves 0-3 the MFMA loop, waves 4-7 the VALU loop (one of each per SIMD).  If the two pipes overlap, mode 2 takes
// max(mode 0, mode 1) / 2 each ... i.e. about half of either; if they share the SIMD it takes (mode 0 + mode 1) / 2.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(8))) __bf16 b8;

__device__ __forceinline__ float mfma_loop(int iters) {
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  b8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
  }
  float s = 0;
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  return s;
}
__device__ __forceinline__ float valu_loop(int iters) {
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 32; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  return s;
}
__global__ __launch_bounds__(512) void probe(float* out, int mode, int iters) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float s;
  if (mode == 0 || (mode == 2 && wave < 4)) s = mfma_loop(iters);
  else s = valu_loop(iters);
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  const char* names[] = {"all 8 waves: 64 MFMAs / iteration", "all 8 waves: 256 VALU / iteration", "waves 0-3 MFMA | waves 4-7 VALU"};
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, out, mode, 10); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, out, mode, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-36s %8.3f ms  = %7.1f cycles (2.4 GHz) per iteration\n", names[mode], ms, ms * 1e-3 * 2.4e9 / iters);
  }
  return 0;
}
