// Depthwise 7x7 convolution, channels-last — forward, data gradient, weight gradient.
// VALU stencil (no MFMA: there is no channel contraction).  Lanes run along the contiguous channel
// axis (16-byte vectors), each thread owns a strip of TW output pixels along X and slides the
// 7-tap window over a register-held input row, so every loaded vector feeds up to 7 FMAs per
// channel; the 7x halo re-reads along Y are served by L1/L2.
#include "vsx_common.h"
#include "../../include/vsx.h"

template <typename T, int TW, bool FLIP>
__global__ __launch_bounds__(256) void dwconv7_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, const T* __restrict__ add,
                                                      T* __restrict__ y, int B, int H, int W, int C) {
  constexpr int VN = VT<T>::N;
  const int ncv = C / VN;
  const int nstrip = (W + TW - 1) / TW;
  const long total = (long)B * H * nstrip * ncv;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int cv = (int)(gid % ncv);
  long r = gid / ncv;
  const int xs = (int)(r % nstrip);
  r /= nstrip;
  const int yy0 = (int)(r % H);
  const int b = (int)(r / H);
  const int x0 = xs * TW, c0 = cv * VN;

  float acc[TW][VN];
#pragma unroll
  for (int o = 0; o < TW; ++o)
#pragma unroll
    for (int j = 0; j < VN; ++j) acc[o][j] = bias ? bias[c0 + j] : 0.f;

  for (int ky = 0; ky < 7; ++ky) {
    const int yy = yy0 + ky - 3;
    if (yy < 0 || yy >= H) continue;
    float wk[7][VN];
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
      const int tap = FLIP ? (6 - ky) * 7 + (6 - kx) : ky * 7 + kx;
#pragma unroll
      for (int j = 0; j < VN; ++j) wk[kx][j] = w[(size_t)tap * C + c0 + j];
    }
    const T* row = x + (((size_t)b * H + yy) * W) * C + c0;
#pragma unroll
    for (int i = 0; i < TW + 6; ++i) {
      const int xx = x0 + i - 3;
      if (xx < 0 || xx >= W) continue;
      float v[VN];
      unpack<T>(ldvec<T>(row + (size_t)xx * C), v);
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const int o = i - kx;
        if (o >= 0 && o < TW) {
#pragma unroll
          for (int j = 0; j < VN; ++j) acc[o][j] = fmaf(v[j], wk[kx][j], acc[o][j]);
        }
      }
    }
  }
#pragma unroll
  for (int o = 0; o < TW; ++o) {
    const int xx = x0 + o;
    if (xx < W) {
      const size_t off = (((size_t)b * H + yy0) * W + xx) * C + c0;
      if (add) {
        float a[VN];
        unpack<T>(ldvec<T>(add + off), a);
#pragma unroll
        for (int j = 0; j < VN; ++j) acc[o][j] += a[j];
      }
      stvec<T>(y + off, pack<T>(acc[o]));
    }
  }
}

// weight gradient: dw[ky*7+kx][c] += sum_{b,y,x} dy[b,y,x,c] * x[b,y+ky-3,x+kx-3,c];  db[c] += sum dy
// thread = (channel vector, chunk of (b,y) rows) for the block's ky (blockIdx.y); loops x with 7 x VN
// register accumulators.  A block is [256/NCVB row-chunk slots][NCVB channel vectors]; slots are
// reduced through LDS atomics so that each block issues one global atomic per (tap, channel).
template <typename T>
__global__ __launch_bounds__(256) void dwconv7_wgrad_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            float* __restrict__ dw, float* __restrict__ db, int B,
                                                            int H, int W, int C, int rows_per_chunk, int ncvb,
                                                            int ncvblocks) {
  constexpr int VN = VT<T>::N;
  __shared__ float red[8 * VN * 64];  // [(7 taps + bias)][VN][ncvb <= 64]
  const int ncv = C / VN;
  const int nrows = B * H;
  const int ky = blockIdx.y;
  const int cvl = threadIdx.x % ncvb, slot = threadIdx.x / ncvb, nslot = 256 / ncvb;
  const int cvblock = blockIdx.x % ncvblocks, chunkblock = blockIdx.x / ncvblocks;
  const int cv = cvblock * ncvb + cvl;
  const int chunk = chunkblock * nslot + slot;
  for (int i = threadIdx.x; i < 8 * VN * ncvb; i += 256) red[i] = 0.f;
  __syncthreads();
  const int c0 = cv * VN;
  float acc[7][VN], bsum[VN];
#pragma unroll
  for (int k = 0; k < 7; ++k)
#pragma unroll
    for (int j = 0; j < VN; ++j) acc[k][j] = 0.f;
#pragma unroll
  for (int j = 0; j < VN; ++j) bsum[j] = 0.f;
  const int rbeg = chunk * rows_per_chunk;
  const int rend = rbeg + rows_per_chunk < nrows ? rbeg + rows_per_chunk : nrows;
  if (cv < ncv) {
    for (int rr = rbeg; rr < rend; ++rr) {
      const int b = rr / H, y = rr - b * H;
      const int yy = y + ky - 3;
      const bool row_ok = yy >= 0 && yy < H;
      if (!row_ok && ky != 3) continue;
      const T* dyr = dy + ((size_t)rr * W) * C + c0;
      const T* xr = x + (((size_t)b * H + (row_ok ? yy : 0)) * W) * C + c0;
      for (int xx = 0; xx < W; ++xx) {
        float d[VN];
        unpack<T>(ldvec<T>(dyr + (size_t)xx * C), d);
        if (ky == 3) {
#pragma unroll
          for (int j = 0; j < VN; ++j) bsum[j] += d[j];
        }
        if (row_ok) {
#pragma unroll
          for (int kx = 0; kx < 7; ++kx) {
            const int xs = xx + kx - 3;
            if (xs >= 0 && xs < W) {
              float v[VN];
              unpack<T>(ldvec<T>(xr + (size_t)xs * C), v);
#pragma unroll
              for (int j = 0; j < VN; ++j) acc[kx][j] = fmaf(d[j], v[j], acc[kx][j]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int kx = 0; kx < 7; ++kx)
#pragma unroll
      for (int j = 0; j < VN; ++j) atomicAdd(&red[(kx * VN + j) * ncvb + cvl], acc[kx][j]);
    if (ky == 3) {
#pragma unroll
      for (int j = 0; j < VN; ++j) atomicAdd(&red[(7 * VN + j) * ncvb + cvl], bsum[j]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 8 * VN * ncvb; i += 256) {
    const int l = i % ncvb, kj = i / ncvb;
    const int j = kj % VN, kx = kj / VN;
    const int c = (cvblock * ncvb + l) * VN + j;
    if (c >= C) continue;
    if (kx < 7)
      atomicAdd(dw + (size_t)(ky * 7 + kx) * C + c, red[i]);
    else if (ky == 3 && db)
      atomicAdd(db + c, red[i]);
  }
}

template <typename T>
static int dw_launch(const void* x, const float* w, const float* bias, const void* add, void* y, int B, int H, int W,
                     int C, bool flip, hipStream_t s) {
  constexpr int VN = VT<T>::N;
  constexpr int TW = 8;
  long total = (long)B * H * vsx_cdiv(W, TW) * (C / VN);
  dim3 grid(vsx_cdiv(total, 256));
  if (flip)
    hipLaunchKernelGGL((dwconv7_kernel<T, TW, true>), grid, dim3(256), 0, s, (const T*)x, w, bias, (const T*)add, (T*)y,
                       B, H, W, C);
  else
    hipLaunchKernelGGL((dwconv7_kernel<T, TW, false>), grid, dim3(256), 0, s, (const T*)x, w, bias, (const T*)add,
                       (T*)y, B, H, W, C);
  VSX_LAUNCH_CHECK();
  return 0;
}

/* K3: timm ConvNeXtBlock.conv_dw (nn.Conv2d(C, C, 7, padding=3, groups=C)).  w is the prepared
 * tap-major fp32 copy [49][C] of conv_dw.weight[C,1,7,7]; y = conv(x) + bias [+ add]. */
extern "C" int32_t vsx_dwconv7_fwd(const void* x, const float* w, const float* bias, const void* add, void* y, int32_t B,
                                   int32_t H, int32_t W, int32_t C, int32_t dtype, vsx_stream_t stream) {
  int vn = dtype == VSX_BF16 ? 8 : 4;
  VSX_CHECK(x && w && y && B > 0 && H > 0 && W > 0 && C > 0 && C % vn == 0, "vsx_dwconv7_fwd: bad arguments (C=%d)", C);
  return dtype == VSX_BF16 ? dw_launch<bf16_t>(x, w, bias, add, y, B, H, W, C, false, (hipStream_t)stream)
                           : dw_launch<float>(x, w, bias, add, y, B, H, W, C, false, (hipStream_t)stream);
}
/* data gradient: dx = conv(dy, flipped w) [+ add]  (add = the residual-branch gradient) */
extern "C" int32_t vsx_dwconv7_bwd_data(const void* dy, const float* w, const void* add, void* dx, int32_t B, int32_t H,
                                        int32_t W, int32_t C, int32_t dtype, vsx_stream_t stream) {
  int vn = dtype == VSX_BF16 ? 8 : 4;
  VSX_CHECK(dy && w && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % vn == 0, "vsx_dwconv7_bwd_data: bad arguments");
  return dtype == VSX_BF16 ? dw_launch<bf16_t>(dy, w, nullptr, add, dx, B, H, W, C, true, (hipStream_t)stream)
                           : dw_launch<float>(dy, w, nullptr, add, dx, B, H, W, C, true, (hipStream_t)stream);
}
/* weight gradient, accumulated (atomicAdd) into dw[49][C] and db[C] (fp32) */
extern "C" int32_t vsx_dwconv7_bwd_weight(const void* dy, const void* x, float* dw, float* db, int32_t B, int32_t H,
                                          int32_t W, int32_t C, int32_t dtype, vsx_stream_t stream) {
  int vn = dtype == VSX_BF16 ? 8 : 4;
  VSX_CHECK(dy && x && dw && B > 0 && H > 0 && W > 0 && C > 0 && C % vn == 0, "vsx_dwconv7_bwd_weight: bad arguments");
  int ncv = C / vn;
  int nrows = B * H;
  int ncvb = 1;
  while (ncvb < ncv && ncvb < 64) ncvb <<= 1;
  int ncvblocks = vsx_cdiv(ncv, ncvb);
  int nslot = 256 / ncvb;
  // aim for ~2048 blocks (x 7 ky); each chunk is a set of whole (b, y) rows
  int want_chunks = vsx_cdiv(2048, 7 * ncvblocks) * nslot;
  if (want_chunks > nrows) want_chunks = nrows;
  int rpc = vsx_cdiv(nrows, want_chunks);
  int nchunk = vsx_cdiv(nrows, rpc);
  dim3 grid(ncvblocks * vsx_cdiv(nchunk, nslot), 7);
  if (dtype == VSX_BF16)
    hipLaunchKernelGGL(dwconv7_wgrad_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                       (const bf16_t*)x, dw, db, B, H, W, C, rpc, ncvb, ncvblocks);
  else
    hipLaunchKernelGGL(dwconv7_wgrad_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dy,
                       (const float*)x, dw, db, B, H, W, C, rpc, ncvb, ncvblocks);
  VSX_LAUNCH_CHECK();
  return 0;
}
