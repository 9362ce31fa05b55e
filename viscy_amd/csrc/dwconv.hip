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
// thread = (channel vector, x-segment of SEG pixels, row slot) for the block's ky (blockIdx.y): per (b,y)
// row it loads SEG dy vectors and the SEG+6 input vectors of row y+ky-3 once and feeds 7 taps from
// registers (sliding window), looping over the rows dealt to its slot.  Block partials are combined
// in LDS and written to one workspace row per block; a second tiny kernel (reduce_rows) folds the
// workspace into dw / db — no same-address atomic storms (they serialise at ~0.2 us each).
template <typename T, int SEG>
__global__ __launch_bounds__(256) void dwconv7_wgrad_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            float* __restrict__ ws, int B, int H, int W, int C, int ncvb,
                                                            int nseg, int nrowslots) {
  constexpr int VN = VT<T>::N;
  extern __shared__ float red[];  // [8][VN][ncvb*nseg_b] → reduced to [8][VN][ncvb]
  const int ncv = C / VN;
  const int nrows = B * H;
  const int ky = blockIdx.y;
  // thread → (cvl, seg, rslot): cvl fastest (coalescing along channels)
  const int lanes_per_row = ncvb * nseg;          // threads covering one image row (all x-segments)
  const int cvl = threadIdx.x % ncvb;
  const int seg = (threadIdx.x / ncvb) % nseg;
  const int rs_in_block = threadIdx.x / lanes_per_row;
  const int rs_per_block = 256 / lanes_per_row;
  const int ncvblocks = (ncv + ncvb - 1) / ncvb;
  const int cvblock = blockIdx.x % ncvblocks, rblock = blockIdx.x / ncvblocks;
  const int cv = cvblock * ncvb + cvl;
  const int rslot = rblock * rs_per_block + rs_in_block;
  const int c0 = cv * VN;
  const int x0 = seg * SEG;
  float acc[7][VN], bsum[VN];
#pragma unroll
  for (int k = 0; k < 7; ++k)
#pragma unroll
    for (int j = 0; j < VN; ++j) acc[k][j] = 0.f;
#pragma unroll
  for (int j = 0; j < VN; ++j) bsum[j] = 0.f;
  const bool active = cv < ncv && rs_in_block < rs_per_block && rslot < nrowslots && x0 < W;
  if (active) {
    for (int rr = rslot; rr < nrows; rr += nrowslots) {
      const int b = rr / H, y = rr - b * H;
      const int yy = y + ky - 3;
      const bool row_ok = yy >= 0 && yy < H;
      if (!row_ok && ky != 3) continue;
      const T* dyr = dy + ((size_t)rr * W) * C + c0;
      const T* xr = x + (((size_t)b * H + (row_ok ? yy : 0)) * W) * C + c0;
      float d[SEG][VN];
#pragma unroll
      for (int i = 0; i < SEG; ++i) {
        if (x0 + i < W) {
          unpack<T>(ldvec<T>(dyr + (size_t)(x0 + i) * C), d[i]);
        } else {
#pragma unroll
          for (int j = 0; j < VN; ++j) d[i][j] = 0.f;
        }
      }
      if (ky == 3) {
#pragma unroll
        for (int i = 0; i < SEG; ++i)
#pragma unroll
          for (int j = 0; j < VN; ++j) bsum[j] += d[i][j];
      }
      if (row_ok) {
#pragma unroll
        for (int i = 0; i < SEG + 6; ++i) {
          const int xs = x0 + i - 3;
          if (xs < 0 || xs >= W) continue;
          float v[VN];
          unpack<T>(ldvec<T>(xr + (size_t)xs * C), v);
#pragma unroll
          for (int kx = 0; kx < 7; ++kx) {
            const int o = i - kx;  // output pixel (within the segment) this input feeds through tap kx
            if (o >= 0 && o < SEG) {
#pragma unroll
              for (int j = 0; j < VN; ++j) acc[kx][j] = fmaf(d[o][j], v[j], acc[kx][j]);
            }
          }
        }
      }
    }
  }
  // block reduction over (seg, rslot) → [8][VN][ncvb]
  const int nred = 8 * VN * ncvb;
  for (int i = threadIdx.x; i < nred; i += 256) red[i] = 0.f;
  __syncthreads();
  if (active) {
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
  // workspace row layout: [50][C] (taps 0..48, then bias at row 49); this block owns rows ky*7..ky*7+6 (+49 if ky==3)
  float* wrow = ws + (size_t)rblock * 50 * C;
  for (int i = threadIdx.x; i < nred; i += 256) {
    const int l = i % ncvb, kj = i / ncvb;
    const int j = kj % VN, kx = kj / VN;
    const int c = (cvblock * ncvb + l) * VN + j;
    if (c >= C) continue;
    if (kx < 7)
      wrow[(size_t)(ky * 7 + kx) * C + c] = red[i];
    else if (ky == 3)
      wrow[(size_t)49 * C + c] = red[i];
  }
}

// out[n] += sum_r ws[r][n]   (same structure as norm.hip's reduce_rows; kept local to this TU)
__global__ __launch_bounds__(256) void dw_reduce_rows_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                             float* __restrict__ db, int R, int C) {
  __shared__ float red[4][64];
  const int N = 50 * C;
  const int nl = threadIdx.x & 63, slot = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + nl;
  const int r0 = blockIdx.y * 64;
  const int r1 = r0 + 64 < R ? r0 + 64 : R;
  float a = 0.f;
  if (n < N) {
#pragma unroll 4
    for (int r = r0 + slot; r < r1; r += 4) a += ws[(size_t)r * N + n];
  }
  red[slot][nl] = a;
  __syncthreads();
  if (slot == 0 && n < N) {
    const float v = red[0][nl] + red[1][nl] + red[2][nl] + red[3][nl];
    if (n < 49 * C)
      atomicAdd(dw + n, v);
    else if (db)
      atomicAdd(db + (n - 49 * C), v);
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
/* weight gradient, accumulated into dw[49][C] and db[C] (fp32).  ws: caller-provided fp32 workspace of
 * ws_rows * 50 * C floats (one row of partials per block, folded by a second kernel). */
extern "C" int32_t vsx_dwconv7_bwd_weight(const void* dy, const void* x, float* dw, float* db, float* ws, int32_t ws_rows,
                                          int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype,
                                          vsx_stream_t stream) {
  int vn = dtype == VSX_BF16 ? 8 : 4;
  VSX_CHECK(dy && x && dw && ws && ws_rows > 0 && B > 0 && H > 0 && W > 0 && C > 0 && C % vn == 0,
            "vsx_dwconv7_bwd_weight: bad arguments");
  constexpr int SEG = 8;
  int ncv = C / vn;
  int nrows = B * H;
  int nseg = vsx_cdiv(W, SEG);
  int ncvb = 1;
  while (ncvb < ncv && ncvb * nseg < 256 && ncvb < 64) ncvb <<= 1;   // channel-vector lanes per row segment
  while (ncvb * nseg > 256 && ncvb > 1) ncvb >>= 1;
  VSX_CHECK(ncvb * nseg <= 256, "vsx_dwconv7_bwd_weight: image width %d too large for one block row", W);
  int ncvblocks = vsx_cdiv(ncv, ncvb);
  int rs_per_block = 256 / (ncvb * nseg);
  // row slots: enough threads to fill the chip (~256k threads over the 7 ky planes), >= 2 rows per slot
  long want_threads = 262144 / 7;
  int rblocks = vsx_cdiv(want_threads, 256L * ncvblocks);
  if (rblocks > vsx_cdiv(nrows, 2 * rs_per_block)) rblocks = vsx_cdiv(nrows, 2 * rs_per_block);
  if (rblocks > ws_rows) rblocks = ws_rows;
  if (rblocks < 1) rblocks = 1;
  int nrowslots = rblocks * rs_per_block;
  hipStream_t st = (hipStream_t)stream;
  hipMemsetAsync(ws, 0, (size_t)rblocks * 50 * C * sizeof(float), st);
  dim3 grid(ncvblocks * rblocks, 7);
  size_t sh = (size_t)8 * vn * ncvb * sizeof(float);
  if (dtype == VSX_BF16)
    hipLaunchKernelGGL((dwconv7_wgrad_kernel<bf16_t, SEG>), grid, dim3(256), sh, st, (const bf16_t*)dy, (const bf16_t*)x, ws,
                       B, H, W, C, ncvb, nseg, nrowslots);
  else
    hipLaunchKernelGGL((dwconv7_wgrad_kernel<float, SEG>), grid, dim3(256), sh, st, (const float*)dy, (const float*)x, ws, B,
                       H, W, C, ncvb, nseg, nrowslots);
  VSX_LAUNCH_CHECK();
  hipLaunchKernelGGL(dw_reduce_rows_kernel, dim3(vsx_cdiv(50 * C, 64), vsx_cdiv(rblocks, 64)), dim3(256), 0, st, ws, dw, db,
                     rblocks, C);
  VSX_LAUNCH_CHECK();
  return 0;
}
