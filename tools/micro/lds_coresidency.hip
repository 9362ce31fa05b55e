// Do two co-resident workgroups on one CU see disjoint LDS when each allocates more than 64 KB?  (Round 4: mlp_fused MODE 6 produced
// wrong rows at 77.8 / 81.9 KB per workgroup once two of them shared a CU, and not at 65.5 KB.)
// Every workgroup fills its whole LDS allocation with a tag (plain ds_write, or LDS-DMA global_load_lds for the first 24 KB), spins
// so that neighbours overlap in time, then checks every word.  Build: hipcc --offload-arch=gfx950 -O2 lds_coresidency.hip -o lds_co
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

extern __shared__ unsigned lds[];

__global__ __launch_bounds__(512) void probe(unsigned* bad, const unsigned* src, int words, int spin, int use_dma, unsigned* where) {
  const unsigned tag = (blockIdx.x + 1) * 0x01000193u;
  for (int i = threadIdx.x; i < words; i += 512) lds[i] = tag ^ (unsigned)i;
  if (use_dma) {
    // overwrite 24 KB (use_dma = 1: the first, 2: the LAST 24 KB of the allocation) by LDS-DMA from a buffer that holds the pattern
    const int dma_off = use_dma == 2 ? words * 4 - 24576 : 0;
    for (int i = threadIdx.x; i < 6144; i += 512) lds[dma_off / 4 + i] = 0xdeadbeefu;
    __syncthreads();
    const char* s = reinterpret_cast<const char*>(src) + (size_t)blockIdx.x * 24576;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int p = wave; p < 24; p += 8)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + p * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(lds) + dma_off + p * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  // keep the workgroup resident for a while; re-write periodically so that an overlapping neighbour gets clobbered both ways
  unsigned acc = 0;
  for (int r = 0; r < spin; ++r) {
    for (int i = threadIdx.x; i < words; i += 512) acc += lds[i];
    __syncthreads();
  }
  unsigned nb = 0, first = 0xffffffffu;
  for (int i = threadIdx.x; i < words; i += 512)
    {
      const int dma_off_w = use_dma == 2 ? words - 6144 : 0;
      const bool in_dma = use_dma && i >= dma_off_w && i < dma_off_w + 6144;
      const unsigned want = in_dma ? (tag ^ (unsigned)(i - dma_off_w)) : (tag ^ (unsigned)i);
      if (lds[i] != want) { ++nb; if (first == 0xffffffffu) first = i; }
    }
  if (nb) { atomicAdd(bad, nb); atomicMin(where, first); }
  if (acc == 0x12345678u) bad[1] = acc;
}

int main() {
  unsigned *bad, *src, *where;
  hipMalloc(&bad, 8); hipMalloc(&where, 4);
  const int nblk = 2048;
  hipMalloc(&src, (size_t)nblk * 24576);
  std::vector<unsigned> h((size_t)nblk * 6144);
  for (int b = 0; b < nblk; ++b) for (int i = 0; i < 6144; ++i) h[(size_t)b * 6144 + i] = ((b + 1) * 0x01000193u) ^ (unsigned)i;  // (offset-corrected on the device for the high window)
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int dma = 0; dma < 3; ++dma)
    for (int kb : {48, 60, 64, 65, 66, 70, 72, 76, 78, 80}) {
      const int bytes = kb * 1024;
      unsigned z[2] = {0, 0}, w = 0xffffffffu;
      hipMemcpy(bad, z, 8, hipMemcpyHostToDevice); hipMemcpy(where, &w, 4, hipMemcpyHostToDevice);
      int occ = 0;
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe, 512, bytes);
      hipLaunchKernelGGL(probe, dim3(nblk), dim3(512), bytes, 0, bad, src, bytes / 4, 40, dma, where);
      hipError_t e = hipDeviceSynchronize();
      hipMemcpy(z, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&w, where, 4, hipMemcpyDeviceToHost);
      printf("%s fill, %2d KB per workgroup (occupancy query: %d per CU): %u corrupted words, first at word %u (byte %u)  [%s]\n",
             dma == 2 ? "DMAhi " : (dma ? "DMA+ds" : "ds    "), kb, occ, z[0], w, w * 4, hipGetErrorString(e));
    }
  return 0;
}
