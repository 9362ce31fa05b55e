// Depthwise 7x7 convolution, channels-last — forward, data gradient, weight gradient.
// VALU stencil (no MFMA: there is no channel contraction).  Lanes run along the contiguous channel
// axis (16-byte vectors); halo tiles are staged through LDS.
#include "vsx_common.h"
#include "../../include/vsx.h"

// ---------------------------------------------------------------------------------------------------
// forward / data-gradient: LDS-tiled stencil.  A block owns TH x TW output pixels x (NCV*VN) channels:
// the (TH+6) x (TW+6) input halo tile and the 49 x CB weights are staged in LDS with independent 16-byte
// loads (all in flight at once — a register-only version serialised its dependent global loads),
// then every thread slides a 7-tap window along a strip of K outputs reading 16-byte vectors from LDS
// (pixel pitch padded by 16 B so the ds_read_b128 of a wave are bank-conflict free).
// ---------------------------------------------------------------------------------------------------
template <typename T, int NCV, int TH, int TW, int K, bool FLIP>
__global__ __launch_bounds__(256) void dwconv7_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, const T* __restrict__ add,
                                                      T* __restrict__ y, int B, int H, int W, int C) {
  constexpr int VN = VT<T>::N;
  constexpr int CB = NCV * VN;
  constexpr int PITCH = CB * (int)sizeof(T) + 16;        // bytes per staged pixel
  constexpr int IH = TH + 6, IW = TW + 6;
  constexpr int SPR = TW / K;                             // strips per tile row
  static_assert(NCV * TH * SPR == 256, "thread mapping");
  typedef typename VT<T>::vec vec;
  __shared__ __attribute__((aligned(16))) char tile[IH * IW * PITCH];
  __shared__ __attribute__((aligned(16))) float wl[49 * CB];

  const int ncb = (C + CB - 1) / CB;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs; give each XCD one contiguous range of
  // (tile, channel-slab) ids so that the slabs of one pixel tile (which split 128-byte lines between them) and the
  // halo-sharing neighbour tiles meet in ONE L2 instead of being fetched through the fabric once per XCD
  int bid = blockIdx.x;
  if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
  const int cb = bid % ncb; bid /= ncb;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int c_base = cb * CB;
  const int y0 = ty * TH, x0 = tx * TW;

  // stage weights (tap-major [49][C] fp32) and the input halo tile
  for (int i = threadIdx.x; i < 49 * CB; i += 256) {
    const int t = i / CB, c = i - t * CB;
    const int tap = FLIP ? 48 - t : t;
    wl[i] = (c_base + c < C) ? w[(size_t)tap * C + c_base + c] : 0.f;
  }
  for (int i = threadIdx.x; i < IH * IW * NCV; i += 256) {
    const int cv = i % NCV;
    const int p = i / NCV;
    const int ix = p % IW, iy = p / IW;
    const int gy = y0 + iy - 3, gx = x0 + ix - 3;
    vec v = vzero<T>();
    if (gy >= 0 && gy < H && gx >= 0 && gx < W && c_base + cv * VN < C)
      v = ldvec<T>(x + (((size_t)b * H + gy) * W + gx) * C + c_base + cv * VN);
    *reinterpret_cast<vec*>(tile + p * PITCH + cv * 16) = v;
  }
  __syncthreads();

  const int cv = threadIdx.x % NCV;
  const int pt = threadIdx.x / NCV;
  const int sx = pt % SPR, sy = pt / SPR;   // strip (sx) of row sy
  const int c0 = c_base + cv * VN;
  float acc[K][VN];
#pragma unroll
  for (int o = 0; o < K; ++o)
#pragma unroll
    for (int j = 0; j < VN; ++j) acc[o][j] = (bias && c0 + j < C) ? bias[c0 + j] : 0.f;

#pragma unroll 1
  for (int ky = 0; ky < 7; ++ky) {
    float wk[7][VN];
#pragma unroll
    for (int kx = 0; kx < 7; ++kx)
#pragma unroll
      for (int j = 0; j < VN; j += 4) {
        float4 t = *reinterpret_cast<const float4*>(wl + (ky * 7 + kx) * CB + cv * VN + j);
        wk[kx][j] = t.x; wk[kx][j + 1] = t.y; wk[kx][j + 2] = t.z; wk[kx][j + 3] = t.w;
      }
    const char* rowp = tile + ((sy + ky) * IW + sx * K) * PITCH + cv * 16;
#pragma unroll
    for (int i = 0; i < K + 6; ++i) {
      float v[VN];
      unpack<T>(*reinterpret_cast<const vec*>(rowp + i * PITCH), v);
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const int o = i - kx;
        if (o >= 0 && o < K) {
#pragma unroll
          for (int j = 0; j < VN; ++j) acc[o][j] = fmaf(v[j], wk[kx][j], acc[o][j]);
        }
      }
    }
  }
  const int gy = y0 + sy;
  if (gy < H && c0 < C) {
#pragma unroll
    for (int o = 0; o < K; ++o) {
      const int gx = x0 + sx * K + o;
      if (gx < W) {
        const size_t off = (((size_t)b * H + gy) * W + gx) * C + c0;
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
}


// weight gradient: dw[ky*7+kx][c] += sum_{b,y,x} dy[b,y,x,c] * x[b,y+ky-3,x+kx-3,c];  db[c] += sum dy
// LDS-tiled like the forward: a block stages the (TH+6)x(TW+6) input halo tile and the TH x TW dy tile of
// NCV*VN channels, thread = (channel vector, ky, tile row) walks its dy row once with a 7-vector sliding
// window of the input row y+ky-3 held in registers (7 taps x VN accumulators).  A block loops over the
// tiles dealt to it, then combines its threads in LDS and writes ONE workspace row of partials; a second
// tiny kernel folds the workspace into dw / db (same-address global atomics serialise at ~0.2 us each).
template <typename T, int NCV, int TH, int TW>
__global__ __launch_bounds__(256) void dwconv7_wgrad_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            float* __restrict__ ws, int B, int H, int W, int C,
                                                            int ngroups) {
  constexpr int VN = VT<T>::N;
  constexpr int CB = NCV * VN;
  constexpr int PITCH = CB * (int)sizeof(T) + 16;
  constexpr int IH = TH + 6, IW = TW + 6;
  constexpr int XT_BYTES = IH * IW * PITCH, DT_BYTES = TH * TW * PITCH;
  static_assert(NCV * 7 * TH <= 256, "thread mapping");
  typedef typename VT<T>::vec vec;
  constexpr int CR = 2;  // rows combined per round of the final reduction (a full TH-row staging buffer would be 51 KB)
  constexpr int PART_BYTES = CR * 50 * CB * 4;
  constexpr int SMEM_BYTES = XT_BYTES + DT_BYTES > PART_BYTES ? XT_BYTES + DT_BYTES : PART_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
  char* xt = smem;
  char* dt = smem + XT_BYTES;

  const int ncb = (C + CB - 1) / CB;
  int bid = blockIdx.x;
  if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);  // XCD-aware (see dwconv7_kernel)
  const int cb = bid % ncb, group = bid / ncb;
  const int c_base = cb * CB;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int ntiles = B * tiles_y * tiles_x;

  const int cv = threadIdx.x % NCV;
  const int ky = (threadIdx.x / NCV) % 7;
  const int row = threadIdx.x / (NCV * 7);
  const bool worker = row < TH;
  float acc[7][VN], bsum[VN];
#pragma unroll
  for (int k = 0; k < 7; ++k)
#pragma unroll
    for (int j = 0; j < VN; ++j) acc[k][j] = 0.f;
#pragma unroll
  for (int j = 0; j < VN; ++j) bsum[j] = 0.f;

  for (int tile = group; tile < ntiles; tile += ngroups) {
    int t = tile;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    __syncthreads();  // previous tile fully consumed
    for (int i = threadIdx.x; i < IH * IW * NCV; i += 256) {
      const int c = i % NCV, p = i / NCV;
      const int ix = p % IW, iy = p / IW;
      const int gy = y0 + iy - 3, gx = x0 + ix - 3;
      vec v = vzero<T>();
      if (gy >= 0 && gy < H && gx >= 0 && gx < W && c_base + c * VN < C)
        v = ldvec<T>(x + (((size_t)b * H + gy) * W + gx) * C + c_base + c * VN);
      *reinterpret_cast<vec*>(xt + p * PITCH + c * 16) = v;
    }
    for (int i = threadIdx.x; i < TH * TW * NCV; i += 256) {
      const int c = i % NCV, p = i / NCV;
      const int ix = p % TW, iy = p / TW;
      const int gy = y0 + iy, gx = x0 + ix;
      vec v = vzero<T>();
      if (gy < H && gx < W && c_base + c * VN < C)
        v = ldvec<T>(dy + (((size_t)b * H + gy) * W + gx) * C + c_base + c * VN);
      *reinterpret_cast<vec*>(dt + p * PITCH + c * 16) = v;
    }
    __syncthreads();
    if (worker) {
      const char* xr = xt + ((row + ky) * IW) * PITCH + cv * 16;
      const char* dr = dt + (row * TW) * PITCH + cv * 16;
      float win[7][VN];
#pragma unroll
      for (int i = 0; i < 6; ++i) unpack<T>(*reinterpret_cast<const vec*>(xr + i * PITCH), win[i]);
      // the window index (o + kx) % 7 is static inside a 7-step body; the outer loop stays rolled so the
      // compiler cannot hoist every LDS read of the row into registers
#pragma unroll 1
      for (int ob = 0; ob < TW; ob += 7) {
#pragma unroll
        for (int u = 0; u < 7; ++u) {
          const int o = ob + u;
          if (o < TW) {
            unpack<T>(*reinterpret_cast<const vec*>(xr + (o + 6) * PITCH), win[(u + 6) % 7]);
            float d[VN];
            unpack<T>(*reinterpret_cast<const vec*>(dr + o * PITCH), d);
            if (ky == 3) {
#pragma unroll
              for (int j = 0; j < VN; ++j) bsum[j] += d[j];
            }
#pragma unroll
            for (int kx = 0; kx < 7; ++kx)
#pragma unroll
              for (int j = 0; j < VN; ++j) acc[kx][j] = fmaf(d[j], win[(u + kx) % 7][j], acc[kx][j]);
          }
        }
      }
    }
  }
  // block reduction over the TH row-threads without atomics: the workers of CR rows at a time park their 7 (+1) x VN
  // partials in LDS as part[row % CR][50][CB] (aliases the tiles); 50*CB thread-strided sums accumulate over the rounds
  float* part = reinterpret_cast<float*>(smem);
  constexpr int NV = (50 * CB + 255) / 256;
  float tot[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) tot[q] = 0.f;
  for (int r0 = 0; r0 < TH; r0 += CR) {
    __syncthreads();
    if (worker && row >= r0 && row < r0 + CR) {
      float* pr = part + (size_t)(row - r0) * 50 * CB;
#pragma unroll
      for (int kx = 0; kx < 7; ++kx)
#pragma unroll
        for (int j = 0; j < VN; j += 4)
          *reinterpret_cast<float4*>(pr + (ky * 7 + kx) * CB + cv * VN + j) =
              make_float4(acc[kx][j], acc[kx][j + 1], acc[kx][j + 2], acc[kx][j + 3]);
      if (ky == 3) {
#pragma unroll
        for (int j = 0; j < VN; j += 4)
          *reinterpret_cast<float4*>(pr + 49 * CB + cv * VN + j) = make_float4(bsum[j], bsum[j + 1], bsum[j + 2], bsum[j + 3]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const int i = threadIdx.x + q * 256;
      if (i < 50 * CB) {
#pragma unroll
        for (int r = 0; r < CR; ++r)
          if (r0 + r < TH) tot[q] += part[(size_t)r * 50 * CB + i];
      }
    }
  }
  float* wrow = ws + (size_t)group * 50 * C;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int i = threadIdx.x + q * 256;
    if (i < 50 * CB) {
      const int t = i / CB, c = c_base + (i - t * CB);
      if (c < C) wrow[(size_t)t * C + c] = tot[q];
    }
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

template <typename T, int NCV, int TH, int TW, int K>
static int dw_launch_cfg(const void* x, const float* w, const float* bias, const void* add, void* y, int B, int H, int W,
                         int C, bool flip, hipStream_t s) {
  constexpr int CB = NCV * VT<T>::N;
  long blocks = (long)B * vsx_cdiv(H, TH) * vsx_cdiv(W, TW) * vsx_cdiv(C, CB);
  dim3 grid((unsigned)blocks);
  if (flip)
    hipLaunchKernelGGL((dwconv7_kernel<T, NCV, TH, TW, K, true>), grid, dim3(256), 0, s, (const T*)x, w, bias,
                       (const T*)add, (T*)y, B, H, W, C);
  else
    hipLaunchKernelGGL((dwconv7_kernel<T, NCV, TH, TW, K, false>), grid, dim3(256), 0, s, (const T*)x, w, bias,
                       (const T*)add, (T*)y, B, H, W, C);
  VSX_LAUNCH_CHECK();
  return 0;
}

template <typename T>
static int dw_launch(const void* x, const float* w, const float* bias, const void* add, void* y, int B, int H, int W,
                     int C, bool flip, hipStream_t s);
int vsx_dwconv7_mfma_try(const void* x, const float* w, const float* bias, const void* add, void* y, int B, int H, int W, int C,
                         bool flip, hipStream_t s, int* taken);  // dwconv_mfma.hip
int vsx_dwconv7_wgrad_mfma_try(const void* dy, const void* x, float* ws, int ws_rows, int B, int H, int W, int C,
                               hipStream_t s, int* taken);  // dwconv_mfma.hip
template <>
int dw_launch<bf16_t>(const void* x, const float* w, const float* bias, const void* add, void* y, int B, int H, int W,
                      int C, bool flip, hipStream_t s) {
  int taken = 0;  // matrix-core Toeplitz path (flag dw_mfma, images >= 16 x 16): memory-bound instead of VALU-bound
  if (int rc = vsx_dwconv7_mfma_try(x, w, bias, add, y, B, H, W, C, flip, s, &taken)) return rc;
  if (taken) return 0;
  // measured alternatives (B = 512, tools/perf_ops.py dw): 8x16 px x 2 outputs / thread (occupancy 4) is 5-9 % slower,
  // 16x16 px x 4 outputs / thread within 2 %, two output rows per thread 3-18 % slower (240 registers, 2 workgroups per CU
  // instead of 3; removed in round 4) -> the kernel is VALU-bound (49 FMAs + bf16 unpacks per output), not LDS-bound
  if (W >= 24) return dw_launch_cfg<bf16_t, 4, 8, 32, 4>(x, w, bias, add, y, B, H, W, C, flip, s);   // 32 ch x 8x32 px
  return dw_launch_cfg<bf16_t, 8, 8, 8, 2>(x, w, bias, add, y, B, H, W, C, flip, s);                 // 64 ch x 8x8 px
}
template <>
int dw_launch<float>(const void* x, const float* w, const float* bias, const void* add, void* y, int B, int H, int W,
                     int C, bool flip, hipStream_t s) {
  if (W >= 24) return dw_launch_cfg<float, 8, 4, 32, 4>(x, w, bias, add, y, B, H, W, C, flip, s);    // 32 ch x 4x32 px
  return dw_launch_cfg<float, 8, 8, 8, 2>(x, w, bias, add, y, B, H, W, C, flip, s);                  // 32 ch x 8x8 px
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
template <typename T, int NCV, int TH, int TW>
static int dw_wgrad_cfg(const void* dy, const void* x, float* dw, float* db, float* ws, int ws_rows, int B, int H, int W,
                        int C, hipStream_t st) {
  constexpr int CB = NCV * VT<T>::N;
  const int ncb = vsx_cdiv(C, CB);
  const int ntiles = B * vsx_cdiv(H, TH) * vsx_cdiv(W, TW);
  int ngroups = ntiles < ws_rows ? ntiles : ws_rows;
  // keep >= ~1024 blocks in flight when there are enough tiles, but no more workspace rows than needed
  if ((long)ngroups * ncb > 4096) ngroups = (int)(4096 / ncb) > 0 ? (int)(4096 / ncb) : 1;
  hipLaunchKernelGGL((dwconv7_wgrad_kernel<T, NCV, TH, TW>), dim3(ngroups * ncb), dim3(256), 0, st, (const T*)dy,
                     (const T*)x, ws, B, H, W, C, ngroups);
  VSX_LAUNCH_CHECK();
  hipLaunchKernelGGL(dw_reduce_rows_kernel, dim3(vsx_cdiv(50 * C, 64), vsx_cdiv(ngroups, 64)), dim3(256), 0, st, ws, dw, db,
                     ngroups, C);
  VSX_LAUNCH_CHECK();
  return 0;
}

/* weight gradient, accumulated into dw[49][C] and db[C] (fp32).  ws: caller-provided fp32 workspace of
 * ws_rows * 50 * C floats (one row of partials per tile group, folded by a second kernel). */
extern "C" int32_t vsx_dwconv7_bwd_weight(const void* dy, const void* x, float* dw, float* db, float* ws, int32_t ws_rows,
                                          int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype,
                                          vsx_stream_t stream) {
  int vn = dtype == VSX_BF16 ? 8 : 4;
  VSX_CHECK(dy && x && dw && ws && ws_rows > 0 && B > 0 && H > 0 && W > 0 && C > 0 && C % vn == 0,
            "vsx_dwconv7_bwd_weight: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VSX_BF16) {
    int rows = 0;  // matrix-core path (flag dw_mfma bit 1): one workspace row per tile range, folded by the same second kernel
    if (int rc = vsx_dwconv7_wgrad_mfma_try(dy, x, ws, ws_rows, B, H, W, C, st, &rows)) return rc;
    if (rows > 0) {
      hipLaunchKernelGGL(dw_reduce_rows_kernel, dim3(vsx_cdiv(50 * C, 64), vsx_cdiv(rows, 64)), dim3(256), 0, st, ws, dw, db, rows,
                         C);
      VSX_LAUNCH_CHECK();
      return 0;
    }
    // 8 x 16-pixel tiles: 35 KB of LDS, 4 workgroups per CU (8 x 32: 63 KB, 2 workgroups, 13-22 % slower; removed in round 4)
    if (W >= 24) return dw_wgrad_cfg<bf16_t, 4, 8, 16>(dy, x, dw, db, ws, ws_rows, B, H, W, C, st);
    return dw_wgrad_cfg<bf16_t, 4, 8, 8>(dy, x, dw, db, ws, ws_rows, B, H, W, C, st);
  }
  if (W >= 24) return dw_wgrad_cfg<float, 4, 8, 32>(dy, x, dw, db, ws, ws_rows, B, H, W, C, st);
  return dw_wgrad_cfg<float, 4, 8, 8>(dy, x, dw, db, ws, ws_rows, B, H, W, C, st);
}
