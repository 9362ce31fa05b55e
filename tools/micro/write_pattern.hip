// Does the WRITE PATTERN decide the HBM write rate?  A [M][W] bf16 matrix (row = W*2 bytes) written
//   (a) linearly (one 16-byte vector per thread, consecutive threads consecutive addresses),
//   (b) as GEMM tiles: workgroup (tile_m, tile_n) writes 128 rows x 256-byte row pieces (tile_n fastest in the grid),
//   (c) as the fused-MLP kernel does: workgroup = 128 rows, loops over 128-byte column slices (each wave 16 rows x 128 B),
//   (d) like (c) but every workgroup writes whole rows at a time from a staged tile (row-contiguous, 16 rows per pass).
// hipcc --offload-arch=gfx950 -O3 tools/micro/write_pattern.hip -o tools/micro/write_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void lin(u32x4* dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = u32x4{1, 2, 3, 4};
}
// (b) tiles of 128 rows x 256 B; grid = tiles_m * tiles_n, tile_n fastest
__global__ __launch_bounds__(256) void gemm_tiles(char* dst, int M, int rowB) {
  const int tiles_n = rowB / 256;
  const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n;
  const int ch = threadIdx.x & 15, r0 = threadIdx.x >> 4;  // 16 threads cover a 256-byte row piece
  for (int r = r0; r < 128; r += 16)
    *reinterpret_cast<u32x4*>(dst + (size_t)(tm * 128 + r) * rowB + tn * 256 + ch * 16) = u32x4{1, 2, 3, 4};
}
// (c) workgroup = 128 rows, slices of 128 B: wave w rows 16w..16w+15, 8 lanes per row piece, 2 passes of 8 rows
__global__ __launch_bounds__(512) void slices(char* dst, int M, int rowB) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ch = lane & 7, rr = lane >> 3;
  const size_t row0 = (size_t)blockIdx.x * 128 + wave * 16;
  for (int s = 0; s < rowB / 128; ++s)
    for (int p = 0; p < 2; ++p)
      *reinterpret_cast<u32x4*>(dst + (row0 + p * 8 + rr) * rowB + s * 128 + ch * 16) = u32x4{1, 2, 3, 4};
}
// (d) workgroup = 128 rows written as whole rows: 512 threads cover 8192 B per pass
__global__ __launch_bounds__(512) void rows(char* dst, int M, int rowB) {
  char* base = dst + (size_t)blockIdx.x * 128 * rowB;
  const size_t total = (size_t)128 * rowB;
  for (size_t o = (size_t)threadIdx.x * 16; o < total; o += 512 * 16) *reinterpret_cast<u32x4*>(base + o) = u32x4{1, 2, 3, 4};
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
  const int M = 2097152;
  for (int W : {384, 896, 1536}) {
    const int rowB = W * 2;
    const size_t bytes = (size_t)M * rowB;
    char* d; (void)hipMalloc(&d, bytes); (void)hipMemset(d, 0, bytes);
    double ms = time_ms([&] { hipLaunchKernelGGL(lin, dim3((unsigned)(bytes / 16 / 256)), dim3(256), 0, 0, (u32x4*)d, bytes / 16); });
    printf("W=%4d (%.2f GB) linear            %7.3f ms %6.2f TB/s\n", W, bytes / 1e9, ms, bytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(gemm_tiles, dim3((M / 128) * (rowB / 256)), dim3(256), 0, 0, d, M, rowB); });
    printf("W=%4d            GEMM tiles 128x256B %7.3f ms %6.2f TB/s\n", W, ms, bytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(slices, dim3(M / 128), dim3(512), 0, 0, d, M, rowB); });
    printf("W=%4d            128-B column slices %7.3f ms %6.2f TB/s\n", W, ms, bytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(rows, dim3(M / 128), dim3(512), 0, 0, d, M, rowB); });
    printf("W=%4d            whole rows          %7.3f ms %6.2f TB/s\n", W, ms, bytes / ms / 1e9);
    (void)hipFree(d);
  }
  return 0;
}
