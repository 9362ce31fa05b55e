// How fast can this part WRITE to HBM?  hipMemset vs plain / non-temporal 16-byte stores, one-shot vs persistent grids.
// hipcc --offload-arch=gfx950 -O3 tools/micro/write_rate.hip -o tools/micro/write_rate && tools/micro/write_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void fill(u32x4* __restrict__ dst, size_t n, unsigned v) {
  const u32x4 val = {v, v + 1, v + 2, v + 3};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (MODE == 0) dst[i] = val;
    if (MODE == 1) __builtin_nontemporal_store(val, dst + i);
  }
}
// read + write (copy) for reference
__global__ __launch_bounds__(256) void copyk(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void readk(const u32x4* __restrict__ src, unsigned* out, size_t n) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const u32x4 v = src[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <typename F>
double time_ms(F f, int reps = 5) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f();
  (void)hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) f();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  const size_t bytes = (size_t)4 << 30, n = bytes / 16;
  u32x4 *a, *b; unsigned* o;
  (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMalloc(&o, 4);
  (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 2, bytes);
  double ms = time_ms([&] { (void)hipMemsetAsync(a, 3, bytes, 0); });
  printf("hipMemset                     %7.3f ms  %6.2f TB/s written\n", ms, bytes / ms / 1e9);
  for (int grid : {2048, 8192, 65536, (int)(n / 256)}) {
    ms = time_ms([&] { hipLaunchKernelGGL(fill<0>, dim3(grid), dim3(256), 0, 0, a, n, 7u); });
    printf("plain 16-B stores  grid %8d %7.3f ms  %6.2f TB/s written\n", grid, ms, bytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(fill<1>, dim3(grid), dim3(256), 0, 0, a, n, 7u); });
    printf("nontemporal stores grid %8d %7.3f ms  %6.2f TB/s written\n", grid, ms, bytes / ms / 1e9);
  }
  for (int grid : {8192, (int)(n / 256)}) {
    ms = time_ms([&] { hipLaunchKernelGGL(copyk, dim3(grid), dim3(256), 0, 0, a, b, n); });
    printf("copy (r + w)       grid %8d %7.3f ms  %6.2f TB/s each way\n", grid, ms, bytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(readk, dim3(grid), dim3(256), 0, 0, a, o, n); });
    printf("read only          grid %8d %7.3f ms  %6.2f TB/s read\n", grid, ms, bytes / ms / 1e9);
  }
  return 0;
}
