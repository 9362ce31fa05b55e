// Memory-system probe for the GEMM A-operand access pattern: a workgroup (256 threads) walks a 128-row tile of a
// row-major [M][K] bf16 matrix in K-slabs of PIECE bytes per row, DEPTH slabs in flight, with a barrier per slab
// (like the GEMM main loop) but no LDS/MFMA work.  Reports GB/s.  Build: hipcc --offload-arch=gfx950 -O3 stream_tiles.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int PIECE, int DEPTH>
__global__ __launch_bounds__(256) void walk(const char* __restrict__ A, int M, int ldb /*row bytes*/, int kbytes, uint32_t* out,
                                            int ntn /*tiles along N sharing the same rows (re-reads)*/) {
  extern __shared__ char pad[];  // occupancy control only
  constexpr int CPR = PIECE / 16;           // 16-byte chunks per row piece
  constexpr int NL = 128 * CPR / 256;       // loads per thread per slab
  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);  // XCD-aware: the ntn re-readers of a tile share one L2
  const int tile = bid / ntn;
  const int m0 = tile * 128;
  uint32_t off[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    int cid = threadIdx.x + i * 256;
    off[i] = (uint32_t)(m0 + cid / CPR) * (uint32_t)ldb + (cid % CPR) * 16;
  }
  const int nslab = kbytes / PIECE;
  uint4 r[DEPTH][NL];
  uint32_t acc = 0;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      int s = d < nslab ? d : nslab - 1;
      r[d][i] = *reinterpret_cast<const uint4*>(A + (size_t)s * PIECE + off[i]);
    }
  for (int s0 = 0; s0 < nslab; s0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int i = 0; i < NL; ++i) acc ^= r[d][i].x ^ r[d][i].y ^ r[d][i].z ^ r[d][i].w;   // consume (waits for slab s0+d)
      int s = s0 + d + DEPTH;
      s = s < nslab ? s : nslab - 1;
#pragma unroll
      for (int i = 0; i < NL; ++i) r[d][i] = *reinterpret_cast<const uint4*>(A + (size_t)s * PIECE + off[i]);
      __syncthreads();
    }
  }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

template <int PIECE, int DEPTH>
float run(const char* A, int M, int K, int ntn, uint32_t* out, int lds) {
  int tiles = M / 128 * ntn;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((walk<PIECE, DEPTH>), dim3(tiles), dim3(256), lds, 0, A, M, K * 2, K * 2, out, ntn);
  hipEventRecord(e0);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((walk<PIECE, DEPTH>), dim3(tiles), dim3(256), lds, 0, A, M, K * 2, K * 2, out, ntn);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5 * 1e3f;
}

int main() {
  const int M = 524288;
  char* A; uint32_t* out;
  hipMalloc(&A, (size_t)M * 1792 + 4096); hipMalloc(&out, 1 << 22);
  hipMemset(A, 1, (size_t)M * 1792);
  for (int K : {896, 384, 224}) {
    for (int ntn : {1, 2, 4, 8}) {
      for (int lds : {49152}) {
        double bytes = (double)M * K * 2;
        float a = run<64, 1>(A, M, K, ntn, out, lds), b = run<64, 2>(A, M, K, ntn, out, lds), c = run<64, 4>(A, M, K, ntn, out, lds);
        float d = run<128, 1>(A, M, K, ntn, out, lds), e = run<128, 2>(A, M, K, ntn, out, lds);
        float f = K % 128 == 0 ? run<256, 1>(A, M, K, ntn, out, lds) : 0.f;
        printf("K=%4d ntn=%d lds=%5d | unique GB/s: 64B x1 %6.0f  x2 %6.0f  x4 %6.0f | 128B x1 %6.0f  x2 %6.0f | 256B x1 %6.0f\n", K, ntn, lds,
               bytes / a / 1e3, bytes / b / 1e3, bytes / c / 1e3, bytes / d / 1e3, bytes / e / 1e3, f > 0 ? bytes / f / 1e3 : 0.0);
      }
    }
  }
  return 0;
}
