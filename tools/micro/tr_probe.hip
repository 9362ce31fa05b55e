#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(int* out, int pitch) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  // lane (r = (l&15)>>2, cseg = l&3) of 16-lane group g = l>>4 supplies the address of 4 contiguous elements
  const int g = l >> 4, r = (l & 15) >> 2, cseg = l & 3;
  short* p = &lds[(g * 4 + r) * pitch + cseg * 4];
  v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  int* d; hipMalloc(&d, 64 * 4 * 4);
  for (int pitch : {16, 48}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, pitch);
    int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pitch %d\n", pitch);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" r%d c%-2d", h[l*4+j] / pitch, h[l*4+j] % pitch); printf("\n"); }
  }
  return 0;
}
