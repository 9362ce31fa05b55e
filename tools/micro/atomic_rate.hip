// VERDICT r5 item 1b, the "form dW1 inside the dh pass" idea priced before it is built: every workgroup of the dh pass would add
// its partial of dW1 (4C x C fp32 = 0.8 MB at C = 224) into a per-XCD private, L2-resident copy.  How fast does the chip take
// fp32 atomic adds into an L2-resident 0.8 MB region per XCD, issued as fully coalesced 64-lane instructions (the best case)?
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/atomic_rate.hip -o tools/micro/bin/atomic_rate && tools/micro/bin/atomic_rate
// Compared with: plain 16-byte streaming stores of the same byte count (what writing dh costs).
#include <hip/hip_runtime.h>
#include <cstdio>

// each workgroup adds `elems` floats (its partial of dW1) into copy (blockIdx.x % 8): workgroups are dealt round-robin to the XCDs
template <int VEC>
__global__ __launch_bounds__(512) void add_partial(float* __restrict__ acc, int elems, int copies) {
  float* dst = acc + (size_t)(blockIdx.x % copies) * elems;
  const float v = 1.0f + blockIdx.x * 1e-6f;
  if (VEC == 1) {
    for (int i = threadIdx.x; i < elems; i += 512) unsafeAtomicAdd(dst + i, v);
  } else {  // packed: two bf16-free fp32 adds per lane through 64-bit... gfx950 has global_atomic_pk_add only for 16-bit types,
            // so "VEC 2" = two consecutive floats per lane (two instructions, 8-byte lane stride: half the coalescing)
    for (int i = threadIdx.x * 2; i < elems; i += 1024) { unsafeAtomicAdd(dst + i, v); unsafeAtomicAdd(dst + i + 1, v); }
  }
}
__global__ __launch_bounds__(512) void store_partial(float4* __restrict__ out, int elems4) {
  float4* dst = out + (size_t)blockIdx.x * elems4;
  const float4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  for (int i = threadIdx.x; i < elems4; i += 512) dst[i] = v;
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
  const int C = 224, H4 = 896, elems = C * H4;          // 200 704 floats = 0.8 MB
  const int wgs = 8192;                                 // B = 512 x 64 x 64 rows / 256 rows per workgroup
  float* acc; float4* out;
  (void)hipMalloc(&acc, (size_t)64 * elems * 4);
  (void)hipMalloc(&out, (size_t)wgs * (256 * H4 * 2));  // the dh tile of a workgroup: 256 rows x 4C bf16 = 458 752 B
  (void)hipMemset(acc, 0, (size_t)64 * elems * 4);
  const double gbytes = (double)wgs * elems * 4 / 1e9;
  printf("partial of dW1 per workgroup: %.1f KB; dh tile per workgroup: %.1f KB; %d workgroups -> %.2f GB of atomic adds vs %.2f GB of dh\n",
         elems * 4 / 1e3, 256 * H4 * 2 / 1e3, wgs, gbytes, (double)wgs * 256 * H4 * 2 / 1e9);
  for (int copies : {8, 16, 64}) {
    double ms = time_ms([&] { hipLaunchKernelGGL(add_partial<1>, dim3(wgs), dim3(512), 0, 0, acc, elems, copies); });
    printf("atomic adds, %2d copies (one per XCD x %d), coalesced: %8.3f ms = %7.1f GB/s = %6.1f G adds/s\n", copies, copies / 8, ms,
           gbytes / (ms * 1e-3), gbytes / 4 / (ms * 1e-3));
  }
  {
    double ms = time_ms([&] { hipLaunchKernelGGL(add_partial<2>, dim3(wgs), dim3(512), 0, 0, acc, elems, 8); });
    printf("atomic adds,  8 copies, 8-byte lane stride:          %8.3f ms = %7.1f GB/s\n", ms, gbytes / (ms * 1e-3));
  }
  {
    const int e4 = 256 * H4 * 2 / 16;
    double ms = time_ms([&] { hipLaunchKernelGGL(store_partial, dim3(wgs), dim3(512), 0, 0, out, e4); });
    printf("plain stores of the dh tiles (%.2f GB):               %8.3f ms = %7.1f GB/s\n", (double)wgs * e4 * 16 / 1e9, ms,
           (double)wgs * e4 * 16 / 1e9 / (ms * 1e-3));
  }
  return 0;
}
