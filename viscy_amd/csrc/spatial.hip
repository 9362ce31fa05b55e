// Pure data-movement kernels of the UNeXt2 path (SURVEY §2.1 K1 gather, K10, K12): stem patch
// gather, pixel-shuffle + skip concat, head pixel-shuffle / pad-pool / depth re-layout.
// All are HBM-bound permutations: one thread produces one 16-byte output vector, lanes run along
// the contiguous output channel axis so stores are coalesced; the strided 2-byte gathers of a
// pixel shuffle are absorbed by L1 (the 4 output pixels of a 2x2 quad share their input lines).
#include "vsx_common.h"
#include "../../include/vsx.h"

// ------------------------------------------------------------------ stem im2col (K1)
// P[(b, yo, xo), d*K + ((ci*kz + dz)*ky + dy)*kx + dx] = norm(x[b, ci, d*kz+dz, yo*ky+dy, xo*kx+dx])
// norm(v) = sub ? (v - sub[b]) / (div[b] + 1e-8) : v      (NormalizeSampled fused into the load)
template <typename T>
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* __restrict__ x, T* __restrict__ P,
                                                          const float* __restrict__ sub, const float* __restrict__ dv,
                                                          int B, int Cin, int Z, int H, int W, int kz, int ky, int kx, int ldp) {
  constexpr int VN = VT<T>::N;
  const int h = H / ky, w = W / kx, D = Z / kz;
  const int K = Cin * kz * ky * kx;
  const int KT = D * K;
  const int nch = ldp / VN;  // ldp >= KT: the columns [KT, ldp) of a row are zero (K padded to the GEMM's slab width)
  const long total = (long)B * h * w * nch;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int ch = (int)(gid % nch);
  long r = gid / nch;
  const int xo = (int)(r % w);
  r /= w;
  const int yo = (int)(r % h);
  const int b = (int)(r / h);
  float s = 0.f, inv = 1.f;
  if (sub) {
    s = sub[b];
    inv = 1.f / (dv[b] + 1e-8f);
  }
  float o[VN];
  if (kx == 4) {
    // the XY kernel of every built stem is 4 wide: 4 consecutive patch columns are 4 consecutive pixels of one input row —
    // one 16-byte load and one index decode per quad instead of four scalar loads with their own div / mod chains
    // (1.10 ms -> see DESIGN §3 at B = 512: the gather was instruction-bound at 1 TB/s)
#pragma unroll
    for (int q = 0; q < VN / 4; ++q) {
      const int col = ch * VN + q * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (col < KT) {
        const int d = col / K, k = col - d * K;
        int t = k >> 2;
        const int dy = t % ky;
        t /= ky;
        const int dz = t % kz;
        const int ci = t / kz;
        v = *reinterpret_cast<const float4*>(x + ((((size_t)b * Cin + ci) * Z + d * kz + dz) * H + yo * ky + dy) * W + xo * 4);
        if (sub) {
          v.x = (v.x - s) * inv; v.y = (v.y - s) * inv; v.z = (v.z - s) * inv; v.w = (v.w - s) * inv;
        }
      }
      o[q * 4] = v.x; o[q * 4 + 1] = v.y; o[q * 4 + 2] = v.z; o[q * 4 + 3] = v.w;
    }
    stvec<T>(P + ((size_t)(b * h + yo) * w + xo) * ldp + ch * VN, pack<T>(o));
    return;
  }
#pragma unroll
  for (int j = 0; j < VN; ++j) {
    int col = ch * VN + j;
    if (col >= KT) {
      o[j] = 0.f;
      continue;
    }
    int d = col / K, k = col - d * K;
    int dx = k % kx;
    int t = k / kx;
    int dy = t % ky;
    t /= ky;
    int dz = t % kz;
    int ci = t / kz;
    float v = x[((((size_t)b * Cin + ci) * Z + d * kz + dz) * H + yo * ky + dy) * W + xo * kx + dx];
    o[j] = sub ? (v - s) * inv : v;
  }
  stvec<T>(P + ((size_t)(b * h + yo) * w + xo) * ldp + ch * VN, pack<T>(o));
}

/* K1 gather half of UNeXt2Stem (viscy_models/components/stems.py:26-50): the Conv3d with
 * kernel = stride = (kz,ky,kx) is a GEMM over non-overlapping patches; this writes the patch
 * matrix, vsx_gemm_nt does the projection.  Optional per-sample (sub, div) fuses
 * NormalizeSampled (viscy_transforms/_normalize.py:72-80) into the load. */
static int stem_im2col_launch(const float* x, void* P, const float* sub, const float* div, int32_t B, int32_t Cin, int32_t Z,
                              int32_t H, int32_t W, int32_t kz, int32_t ky, int32_t kx, int32_t ldp, int32_t dtype,
                              vsx_stream_t stream) {
  int vn = dtype == VSX_BF16 ? 8 : 4;
  VSX_CHECK(x && P && B > 0 && Cin > 0, "vsx_stem_im2col: bad arguments");
  VSX_CHECK(Z % kz == 0 && H % ky == 0 && W % kx == 0, "vsx_stem_im2col: (%d,%d,%d) not divisible by kernel (%d,%d,%d)",
            Z, H, W, kz, ky, kx);
  const int KT = (Z / kz) * Cin * kz * ky * kx;
  VSX_CHECK(KT % vn == 0, "vsx_stem_im2col: patch size must be a multiple of %d", vn);
  VSX_CHECK(ldp >= KT && ldp % vn == 0, "vsx_stem_im2col: row length %d must be >= %d and a multiple of %d", ldp, KT, vn);
  VSX_CHECK((sub == nullptr) == (div == nullptr), "vsx_stem_im2col: sub/div must both be set or both NULL");
  long total = (long)B * (H / ky) * (W / kx) * (ldp / vn);
  dim3 grid(vsx_cdiv(total, 256));
  if (dtype == VSX_BF16)
    hipLaunchKernelGGL(stem_im2col_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)P, sub, div, B,
                       Cin, Z, H, W, kz, ky, kx, ldp);
  else
    hipLaunchKernelGGL(stem_im2col_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, x, (float*)P, sub, div, B,
                       Cin, Z, H, W, kz, ky, kx, ldp);
  VSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t vsx_stem_im2col(const float* x, void* P, const float* sub, const float* div, int32_t B, int32_t Cin,
                                   int32_t Z, int32_t H, int32_t W, int32_t kz, int32_t ky, int32_t kx, int32_t dtype,
                                   vsx_stream_t stream) {
  return stem_im2col_launch(x, P, sub, div, B, Cin, Z, H, W, kz, ky, kx, (Z / (kz > 0 ? kz : 1)) * Cin * kz * ky * kx, dtype, stream);
}

/* the same gather with rows of `ldp` >= patch-size elements, the tail zero-filled: K = 80 (the 5x4x4 single-channel stem)
 * becomes 96, a whole number of 32-deep MFMA slabs, and the projection runs on the lean GEMM instead of the generic one
 * (2.9 ms -> the launch's HBM time at B = 512) */
extern "C" int32_t vsx_stem_im2col_ld(const float* x, void* P, const float* sub, const float* div, int32_t B, int32_t Cin,
                                      int32_t Z, int32_t H, int32_t W, int32_t kz, int32_t ky, int32_t kx, int32_t ldp,
                                      int32_t dtype, vsx_stream_t stream) {
  return stem_im2col_launch(x, P, sub, div, B, Cin, Z, H, W, kz, ky, kx, ldp, dtype, stream);
}

// dst[r][k] = k < K ? src[r][k] : 0   (row length K -> Kp; weight-space helper of the padded stem GEMM)
template <typename T>
__global__ __launch_bounds__(256) void pad_cols_kernel(const T* __restrict__ src, T* __restrict__ dst, int R, int K, int Kp) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)R * Kp) return;
  const int k = (int)(gid % Kp), r = (int)(gid / Kp);
  dst[gid] = k < K ? src[(size_t)r * K + k] : from_f32<T>(0.f);
}
extern "C" int32_t vsx_pad_cols(const void* src, void* dst, int32_t R, int32_t K, int32_t Kp, int32_t dtype, vsx_stream_t stream) {
  VSX_CHECK(src && dst && R > 0 && K > 0 && Kp >= K, "vsx_pad_cols: bad arguments");
  dim3 grid(vsx_cdiv((long)R * Kp, 256));
  if (dtype == VSX_BF16)
    hipLaunchKernelGGL(pad_cols_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst, R, K, Kp);
  else
    hipLaunchKernelGGL(pad_cols_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, (float*)dst, R, K, Kp);
  VSX_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ dense 3x3 convolution as a GEMM: patch gather / scatter
// (decoder_upsample_pre_conv: MONAI SubpixelUpsample's Conv2d(C, C, 3, padding=1) in front of the pixel shuffle,
// blocks.py:138-146).  col[m][t*C + c] = x[b, y + t/3 - 1, x + t%3 - 1, c] (zero outside the image); the transpose
// gathers the nine shifted contributions of a pixel (no atomics): dx[m][c] = sum_t dcol[m - shift(t)][t*C + c].
template <typename T>
__global__ __launch_bounds__(256) void im2col3x3_kernel(const T* __restrict__ x, T* __restrict__ col, int B, int H, int W, int C) {
  constexpr int VN = VT<T>::N;
  const int nch = C / VN;
  const long total = (long)B * H * W * 9 * nch;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int cv = (int)(gid % nch);
  long r = gid / nch;
  const int t = (int)(r % 9);
  const long m = r / 9;
  const int xx = (int)(m % W);
  const int yy = (int)((m / W) % H);
  const int sy = yy + t / 3 - 1, sx = xx + t % 3 - 1;
  typename VT<T>::vec v = vzero<T>();
  if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = ldvec<T>(x + ((m - (long)yy * W - xx) + (long)sy * W + sx) * C + cv * VN);
  stvec<T>(col + (m * 9 + t) * C + cv * VN, v);
}

template <typename T>
__global__ __launch_bounds__(256) void col2im3x3_kernel(const T* __restrict__ dcol, T* __restrict__ dx, int B, int H, int W, int C) {
  constexpr int VN = VT<T>::N;
  const int nch = C / VN;
  const long total = (long)B * H * W * nch;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int cv = (int)(gid % nch);
  const long m = gid / nch;
  const int xx = (int)(m % W);
  const int yy = (int)((m / W) % H);
  float acc[VN];
#pragma unroll
  for (int j = 0; j < VN; ++j) acc[j] = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    // output pixel (oy, ox) read this pixel through tap t when oy + t/3 - 1 == yy, ox + t%3 - 1 == xx
    const int oy = yy - (t / 3 - 1), ox = xx - (t % 3 - 1);
    if (oy >= 0 && oy < H && ox >= 0 && ox < W) {
      float f[VN];
      unpack<T>(ldvec<T>(dcol + (((m - (long)yy * W - xx) + (long)oy * W + ox) * 9 + t) * C + cv * VN), f);
#pragma unroll
      for (int j = 0; j < VN; ++j) acc[j] += f[j];
    }
  }
  stvec<T>(dx + m * C + cv * VN, pack<T>(acc));
}

extern "C" int32_t vsx_im2col3x3(const void* x, void* col, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype,
                                 vsx_stream_t stream) {
  const int vn = dtype == VSX_BF16 ? 8 : 4;
  VSX_CHECK(x && col && B > 0 && H > 0 && W > 0 && C > 0 && C % vn == 0, "vsx_im2col3x3: bad arguments (C=%d)", C);
  dim3 grid(vsx_cdiv((long)B * H * W * 9 * (C / vn), 256));
  if (dtype == VSX_BF16)
    hipLaunchKernelGGL(im2col3x3_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)col, B, H, W, C);
  else
    hipLaunchKernelGGL(im2col3x3_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)col, B, H, W, C);
  VSX_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t vsx_col2im3x3(const void* dcol, void* dx, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype,
                                 vsx_stream_t stream) {
  const int vn = dtype == VSX_BF16 ? 8 : 4;
  VSX_CHECK(dcol && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % vn == 0, "vsx_col2im3x3: bad arguments (C=%d)", C);
  dim3 grid(vsx_cdiv((long)B * H * W * (C / vn), 256));
  if (dtype == VSX_BF16)
    hipLaunchKernelGGL(col2im3x3_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dcol, (bf16_t*)dx, B, H, W, C);
  else
    hipLaunchKernelGGL(col2im3x3_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dcol, (float*)dx, B, H, W, C);
  VSX_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ pixel shuffle x2 + concat (K10)
// out[b, Y, X, j] = j < c ? low[b, Y/2, X/2, 4j + 2(Y&1) + (X&1)] : skip[b, Y, X, j - c]
template <typename T>
__global__ __launch_bounds__(256) void ps_cat_fwd_kernel(const T* __restrict__ low, const T* __restrict__ skip,
                                                         T* __restrict__ out, int B, int h, int w, int c, int cs) {
  constexpr int VN = VT<T>::N;
  const int ct = c + cs;
  const int nch = ct / VN;
  const int H2 = 2 * h, W2 = 2 * w;
  const long total = (long)B * H2 * W2 * nch;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int ch = (int)(gid % nch);
  const long pix = gid / nch;
  const int X = (int)(pix % W2);
  const long r = pix / W2;
  const int Y = (int)(r % H2);
  const int b = (int)(r / H2);
  const int j0 = ch * VN;
  T* dst = out + (size_t)pix * ct + j0;
  if (j0 >= c && ((j0 - c) % VN == 0) && (cs % VN == 0)) {
    stvec<T>(dst, ldvec<T>(skip + (size_t)pix * cs + (j0 - c)));
    return;
  }
  const T* lp = low + (((size_t)b * h + (Y >> 1)) * w + (X >> 1)) * (4 * c) + 2 * (Y & 1) + (X & 1);
  typename VT<T>::vec tv;
  T* tmp = reinterpret_cast<T*>(&tv);
#pragma unroll
  for (int j = 0; j < VN; ++j) {
    int jj = j0 + j;
    tmp[j] = jj < c ? lp[4 * jj] : skip[(size_t)pix * cs + (jj - c)];
  }
  stvec<T>(dst, tv);
}

// backward: one thread per dcat vector; low-part channels scatter to dlow (each element written once),
// skip-part channels go to dskip.
template <typename T>
__global__ __launch_bounds__(256) void ps_cat_bwd_kernel(const T* __restrict__ dcat, T* __restrict__ dlow,
                                                         T* __restrict__ dskip, int B, int h, int w, int c, int cs) {
  constexpr int VN = VT<T>::N;
  const int ct = c + cs;
  const int nch = ct / VN;
  const int H2 = 2 * h, W2 = 2 * w;
  const long total = (long)B * H2 * W2 * nch;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int ch = (int)(gid % nch);
  const long pix = gid / nch;
  const int X = (int)(pix % W2);
  const long r = pix / W2;
  const int Y = (int)(r % H2);
  const int b = (int)(r / H2);
  const int j0 = ch * VN;
  typename VT<T>::vec v = ldvec<T>(dcat + (size_t)pix * ct + j0);
  if (j0 >= c && ((j0 - c) % VN == 0) && (cs % VN == 0)) {
    stvec<T>(dskip + (size_t)pix * cs + (j0 - c), v);
    return;
  }
  const T* tv = reinterpret_cast<const T*>(&v);
  T* lp = dlow + (((size_t)b * h + (Y >> 1)) * w + (X >> 1)) * (4 * c) + 2 * (Y & 1) + (X & 1);
#pragma unroll
  for (int j = 0; j < VN; ++j) {
    int jj = j0 + j;
    if (jj < c)
      lp[4 * jj] = tv[j];
    else
      dskip[(size_t)pix * cs + (jj - c)] = tv[j];
  }
}

// Vector form of the two kernels above for c % VN == 0 and cs % VN == 0 (every configuration of the path): the 4 sub-pixels of a
// low-resolution pixel take their VN channels j0 .. j0 + VN - 1 from the 4 * VN CONTIGUOUS low channels 4 j0 .. 4 j0 + 4 VN - 1, so
// one thread moves four whole vectors in and four out and (de)interleaves them in registers — the element-wise kernels issue
// VN two-byte accesses per vector (1.3 ms per step at 3.2 – 4.9 TB/s of 16-byte traffic's worth).
// items: [0, na) = (low pixel, channel group) of the shuffled part, [na, na + nb) = (output pixel, vector) of the skip part
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void ps_cat_vec_kernel(const T* __restrict__ a0, const T* __restrict__ a1, T* __restrict__ o0,
                                                         T* __restrict__ o1, int B, int h, int w, int c, int cs) {
  // forward:  a0 = low, a1 = skip, o0 = out (o1 unused);  backward: a0 = dcat, o0 = dlow, o1 = dskip (a1 unused)
  constexpr int VN = VT<T>::N;
  typedef typename VT<T>::vec vec;
  const int ct = c + cs, ng = c / VN, nsv = cs / VN;
  const int H2 = 2 * h, W2 = 2 * w;
  const long na = (long)B * h * w * ng, nb = (long)B * H2 * W2 * nsv;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= na + nb) return;
  if (gid >= na) {  // skip channels: a vector copy between the concatenated tensor and the skip tensor
    const long i = gid - na;
    const int k = (int)(i % nsv);
    const long pix = i / nsv;
    if (BWD) stvec<T>(o1 + (size_t)pix * cs + k * VN, ldvec<T>(a0 + (size_t)pix * ct + c + k * VN));
    else stvec<T>(o0 + (size_t)pix * ct + c + k * VN, ldvec<T>(a1 + (size_t)pix * cs + k * VN));
    return;
  }
  const int jg = (int)(gid % ng);
  const long lp = gid / ng;
  const int x = (int)(lp % w);
  const long r = lp / w;
  const int y = (int)(r % h);
  const int b = (int)(r / h);
  const size_t lowoff = (size_t)lp * (4 * c) + (size_t)jg * 4 * VN;
  const size_t pix00 = ((size_t)b * H2 + 2 * y) * W2 + 2 * x;  // sub-pixel s sits at pix00 + (s >> 1) * W2 + (s & 1)
  T lo[4 * VN], hi[4][VN];
  if (!BWD) {
#pragma unroll
    for (int v = 0; v < 4; ++v) *reinterpret_cast<vec*>(lo + v * VN) = ldvec<T>(a0 + lowoff + v * VN);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
      for (int j = 0; j < VN; ++j) hi[s4][j] = lo[4 * j + s4];
      stvec<T>(o0 + (pix00 + (s4 >> 1) * W2 + (s4 & 1)) * ct + jg * VN, *reinterpret_cast<const vec*>(hi[s4]));
    }
  } else {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
      *reinterpret_cast<vec*>(hi[s4]) = ldvec<T>(a0 + (pix00 + (s4 >> 1) * W2 + (s4 & 1)) * ct + jg * VN);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int j = 0; j < VN; ++j) lo[4 * j + s4] = hi[s4][j];
#pragma unroll
    for (int v = 0; v < 4; ++v) stvec<T>(o0 + lowoff + v * VN, *reinterpret_cast<const vec*>(lo + v * VN));
  }
}

/* K10: MONAI UpSample(mode="pixelshuffle", pre_conv=None) + torch.cat([up, skip], 1)
 * (viscy_models/components/blocks.py:138-146,170-171).  skip may be NULL (cs = 0). */
extern "C" int32_t vsx_pixel_shuffle_cat_fwd(const void* low, const void* skip, void* out, int32_t B, int32_t h,
                                             int32_t w, int32_t c, int32_t cs, int32_t dtype, vsx_stream_t stream) {
  int vn = dtype == VSX_BF16 ? 8 : 4;
  VSX_CHECK(low && out && B > 0 && h > 0 && w > 0 && c > 0 && cs >= 0, "vsx_pixel_shuffle_cat_fwd: bad arguments");
  VSX_CHECK((cs == 0) == (skip == nullptr), "vsx_pixel_shuffle_cat_fwd: skip pointer / cs mismatch");
  VSX_CHECK((c + cs) % vn == 0, "vsx_pixel_shuffle_cat_fwd: c+cs=%d must be a multiple of %d", c + cs, vn);
  if (c % vn == 0 && cs % vn == 0) {  // whole vectors on both sides: the register-(de)interleaving form
    const long items = (long)B * h * w * (c / vn) + (long)B * 4 * h * w * (cs / vn);
    dim3 g(vsx_cdiv(items, 256));
    if (dtype == VSX_BF16)
      hipLaunchKernelGGL((ps_cat_vec_kernel<bf16_t, false>), g, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)low,
                         (const bf16_t*)skip, (bf16_t*)out, (bf16_t*)nullptr, B, h, w, c, cs);
    else
      hipLaunchKernelGGL((ps_cat_vec_kernel<float, false>), g, dim3(256), 0, (hipStream_t)stream, (const float*)low,
                         (const float*)skip, (float*)out, (float*)nullptr, B, h, w, c, cs);
    VSX_LAUNCH_CHECK();
    return 0;
  }
  long total = (long)B * 4 * h * w * ((c + cs) / vn);
  dim3 grid(vsx_cdiv(total, 256));
  if (dtype == VSX_BF16)
    hipLaunchKernelGGL(ps_cat_fwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)low,
                       (const bf16_t*)skip, (bf16_t*)out, B, h, w, c, cs);
  else
    hipLaunchKernelGGL(ps_cat_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)low,
                       (const float*)skip, (float*)out, B, h, w, c, cs);
  VSX_LAUNCH_CHECK();
  return 0;
}
extern "C" int32_t vsx_pixel_shuffle_cat_bwd(const void* dcat, void* dlow, void* dskip, int32_t B, int32_t h, int32_t w,
                                             int32_t c, int32_t cs, int32_t dtype, vsx_stream_t stream) {
  int vn = dtype == VSX_BF16 ? 8 : 4;
  VSX_CHECK(dcat && dlow && B > 0 && h > 0 && w > 0 && c > 0 && cs >= 0, "vsx_pixel_shuffle_cat_bwd: bad arguments");
  VSX_CHECK((cs == 0) == (dskip == nullptr), "vsx_pixel_shuffle_cat_bwd: dskip pointer / cs mismatch");
  VSX_CHECK((c + cs) % vn == 0, "vsx_pixel_shuffle_cat_bwd: c+cs=%d must be a multiple of %d", c + cs, vn);
  if (c % vn == 0 && cs % vn == 0) {
    const long items = (long)B * h * w * (c / vn) + (long)B * 4 * h * w * (cs / vn);
    dim3 g(vsx_cdiv(items, 256));
    if (dtype == VSX_BF16)
      hipLaunchKernelGGL((ps_cat_vec_kernel<bf16_t, true>), g, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dcat,
                         (const bf16_t*)nullptr, (bf16_t*)dlow, (bf16_t*)dskip, B, h, w, c, cs);
    else
      hipLaunchKernelGGL((ps_cat_vec_kernel<float, true>), g, dim3(256), 0, (hipStream_t)stream, (const float*)dcat,
                         (const float*)nullptr, (float*)dlow, (float*)dskip, B, h, w, c, cs);
    VSX_LAUNCH_CHECK();
    return 0;
  }
  long total = (long)B * 4 * h * w * ((c + cs) / vn);
  dim3 grid(vsx_cdiv(total, 256));
  if (dtype == VSX_BF16)
    hipLaunchKernelGGL(ps_cat_bwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dcat,
                       (bf16_t*)dlow, (bf16_t*)dskip, B, h, w, c, cs);
  else
    hipLaunchKernelGGL(ps_cat_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dcat,
                       (float*)dlow, (float*)dskip, B, h, w, c, cs);
  VSX_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ head pixel shuffle (+ pad-pool) (K12)
// hin[b, Y, X, z*C3 + c3] = pool( v )(Y, X),  v(Y,X) = dec[b, Y/2, X/2, 4*(c3*D + z) + 2(Y&1) + (X&1)]
// pool(v)(Y,X) = 0.25 * (v(Y,X) + v(Y-1,X) + v(Y,X-1) + v(Y-1,X-1)), zero outside  (ConstantPad2d((1,0,1,0)) + AvgPool2d(2,1))
template <typename T>
__global__ __launch_bounds__(256) void head_shuffle_fwd_kernel(const T* __restrict__ dec, T* __restrict__ hin, int B,
                                                               int h, int w, int C3, int D, int pool) {
  constexpr int VN = VT<T>::N;
  const int Cm = C3 * D;
  const int nch = Cm / VN;
  const int H2 = 2 * h, W2 = 2 * w;
  const long total = (long)B * H2 * W2 * nch;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int ch = (int)(gid % nch);
  const long pix = gid / nch;
  const int X = (int)(pix % W2);
  const long r = pix / W2;
  const int Y = (int)(r % H2);
  const int b = (int)(r / H2);
  float o[VN];
#pragma unroll
  for (int j = 0; j < VN; ++j) {
    const int cp = ch * VN + j;
    const int z = cp / C3, c3 = cp - z * C3;
    const int src = 4 * (c3 * D + z);
    float acc = 0.f;
    const int ntap = pool ? 4 : 1;
    for (int t = 0; t < ntap; ++t) {
      const int yy = Y - (t >> 1), xx = X - (t & 1);
      if (yy < 0 || xx < 0) continue;
      acc += to_f32<T>(dec[(((size_t)b * h + (yy >> 1)) * w + (xx >> 1)) * (4 * Cm) + src + 2 * (yy & 1) + (xx & 1)]);
    }
    o[j] = pool ? 0.25f * acc : acc;
  }
  stvec<T>(hin + (size_t)pix * Cm + ch * VN, pack<T>(o));
}

template <typename T>
__global__ __launch_bounds__(256) void head_shuffle_bwd_kernel(const T* __restrict__ dhin, T* __restrict__ ddec, int B,
                                                               int h, int w, int C3, int D, int pool) {
  constexpr int VN = VT<T>::N;
  const int Cm = C3 * D;
  const int C4 = 4 * Cm;
  const int nch = C4 / VN;
  const int H2 = 2 * h, W2 = 2 * w;
  const long total = (long)B * h * w * nch;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int ch = (int)(gid % nch);
  const long pix = gid / nch;
  const int x = (int)(pix % w);
  const long r = pix / w;
  const int y = (int)(r % h);
  const int b = (int)(r / h);
  float o[VN];
#pragma unroll
  for (int j = 0; j < VN; ++j) {
    const int lc = ch * VN + j;
    const int chn = lc >> 2, sub = lc & 3;
    const int c3 = chn / D, z = chn - c3 * D;
    const int cp = z * C3 + c3;
    const int Y = 2 * y + (sub >> 1), X = 2 * x + (sub & 1);
    float acc = 0.f;
    const int ntap = pool ? 4 : 1;
    for (int t = 0; t < ntap; ++t) {
      const int yy = Y + (t >> 1), xx = X + (t & 1);
      if (yy >= H2 || xx >= W2) continue;
      acc += to_f32<T>(dhin[(((size_t)b * H2 + yy) * W2 + xx) * Cm + cp]);
    }
    o[j] = pool ? 0.25f * acc : acc;
  }
  stvec<T>(ddec + (size_t)pix * C4 + ch * VN, pack<T>(o));
}

// LDS-tiled versions of the two permutations above (one scalar 2-byte global load per tap per element made the
// direct versions run at 0.9 - 1.2 TB/s): a workgroup stages TS x TS decoder pixels (+ the one-pixel pool halo) with
// 16-byte row-contiguous loads, the in-pixel transpose (c3, z, dy, dx) <-> (dy, dx | z, c3) and the 2x2 pooling are LDS
// reads, stores are 16-byte row-contiguous again.
template <typename T, int TS>
__global__ __launch_bounds__(256) void head_shuffle_fwd_tiled_kernel(const T* __restrict__ dec, T* __restrict__ hin, int h,
                                                                     int w, int C3, int D, int pool) {
  constexpr int VN = VT<T>::N;
  constexpr int MAXC4 = 256;                       // 4 * C3 * D <= 256 (C3 * D <= 64)
  __shared__ T tile[(TS + 1) * (TS + 1) * MAXC4];
  __shared__ short srcb[64];
  const int Cm = C3 * D, C4 = 4 * Cm;
  const int b = blockIdx.z, y0 = blockIdx.y * TS, x0 = blockIdx.x * TS;
  for (int cp = threadIdx.x; cp < Cm; cp += 256) {
    const int z = cp / C3, c3 = cp - z * C3;
    srcb[cp] = (short)(4 * (c3 * D + z));
  }
  const int cpp = C4 / VN;
  for (int i = threadIdx.x; i < (TS + 1) * (TS + 1) * cpp; i += 256) {
    const int ch = i % cpp, pl = i / cpp;
    const int r = pl / (TS + 1), c = pl - r * (TS + 1);
    const int y = y0 - 1 + r, x = x0 - 1 + c;
    typename VT<T>::vec v = vzero<T>();
    if (y >= 0 && x >= 0 && y < h && x < w) v = ldvec<T>(dec + (((size_t)b * h + y) * w + x) * C4 + ch * VN);
    *reinterpret_cast<typename VT<T>::vec*>(tile + pl * MAXC4 + ch * VN) = v;
  }
  __syncthreads();
  const int nch = Cm / VN;
  const int H2 = 2 * h, W2 = 2 * w;
  for (int i = threadIdx.x; i < 4 * TS * TS * nch; i += 256) {
    const int ch = i % nch, pl = i / nch;
    const int Yl = pl / (2 * TS), Xl = pl - Yl * (2 * TS);
    const int Y = 2 * y0 + Yl, X = 2 * x0 + Xl;
    if (Y >= H2 || X >= W2) continue;
    float o[VN];
#pragma unroll
    for (int j = 0; j < VN; ++j) o[j] = 0.f;
    const int ntap = pool ? 4 : 1;
    for (int t = 0; t < ntap; ++t) {
      const int yy = Y - (t >> 1), xx = X - (t & 1);  // >= -1: the row / column of zeros staged above is the pad
      const int r = (yy >> 1) - y0 + 1, c = (xx >> 1) - x0 + 1;
      const T* src = tile + (r * (TS + 1) + c) * MAXC4 + 2 * (yy & 1) + (xx & 1);
#pragma unroll
      for (int j = 0; j < VN; ++j) o[j] += to_f32<T>(src[srcb[ch * VN + j]]);
    }
    if (pool) {
#pragma unroll
      for (int j = 0; j < VN; ++j) o[j] *= 0.25f;
    }
    stvec<T>(hin + (((size_t)b * H2 + Y) * W2 + X) * Cm + ch * VN, pack<T>(o));
  }
}

template <typename T, int TS>
__global__ __launch_bounds__(256) void head_shuffle_bwd_tiled_kernel(const T* __restrict__ dhin, T* __restrict__ ddec, int h,
                                                                     int w, int C3, int D, int pool) {
  constexpr int VN = VT<T>::N;
  constexpr int MAXCM = 64;
  constexpr int TO = 2 * TS + 1;                   // output-resolution tile + the pool halo (bottom / right)
  __shared__ T tile[TO * TO * MAXCM];
  __shared__ short cpof[64];                       // chn = c3 * D + z  ->  cp = z * C3 + c3
  const int Cm = C3 * D, C4 = 4 * Cm;
  const int H2 = 2 * h, W2 = 2 * w;
  const int b = blockIdx.z, y0 = blockIdx.y * TS, x0 = blockIdx.x * TS;
  for (int chn = threadIdx.x; chn < Cm; chn += 256) {
    const int c3 = chn / D, z = chn - c3 * D;
    cpof[chn] = (short)(z * C3 + c3);
  }
  const int nch = Cm / VN;
  for (int i = threadIdx.x; i < TO * TO * nch; i += 256) {
    const int ch = i % nch, pl = i / nch;
    const int r = pl / TO, c = pl - r * TO;
    const int Y = 2 * y0 + r, X = 2 * x0 + c;
    typename VT<T>::vec v = vzero<T>();
    if (Y < H2 && X < W2) v = ldvec<T>(dhin + (((size_t)b * H2 + Y) * W2 + X) * Cm + ch * VN);
    *reinterpret_cast<typename VT<T>::vec*>(tile + pl * MAXCM + ch * VN) = v;
  }
  __syncthreads();
  const int cpp = C4 / VN;
  for (int i = threadIdx.x; i < TS * TS * cpp; i += 256) {
    const int ch = i % cpp, pl = i / cpp;
    const int yl = pl / TS, xl = pl - yl * TS;
    const int y = y0 + yl, x = x0 + xl;
    if (y >= h || x >= w) continue;
    float o[VN];
#pragma unroll
    for (int j = 0; j < VN; ++j) {
      const int lc = ch * VN + j;
      const int cp = cpof[lc >> 2], sub = lc & 3;
      const int r = 2 * yl + (sub >> 1), c = 2 * xl + (sub & 1);
      float acc = to_f32<T>(tile[(r * TO + c) * MAXCM + cp]);
      if (pool) {  // the halo row / column beyond the image was staged as zeros
        acc += to_f32<T>(tile[(r * TO + c + 1) * MAXCM + cp]) + to_f32<T>(tile[((r + 1) * TO + c) * MAXCM + cp]) +
               to_f32<T>(tile[((r + 1) * TO + c + 1) * MAXCM + cp]);
        acc *= 0.25f;
      }
      o[j] = acc;
    }
    stvec<T>(ddec + (((size_t)b * h + y) * w + x) * C4 + ch * VN, pack<T>(o));
  }
}

// ------------------------------------------------------------------ round 6: the two permutations on column strips (bf16, pooled)
// The tiled kernels above do the in-pixel transpose with 2-byte LDS reads: 4 taps x Cm scalar reads per output pixel (896 per
// decoder pixel) — the LDS instruction rate, not HBM, sets their 3.0 / 4.0 TB/s.  Here a workgroup walks DOWN a strip of 64
// decoder columns; thread (xl, q) keeps the 16-byte chunk q (two shuffle groups cp' = 2q, 2q + 1 x the 2 x 2 sub-pixels) of its
// pixel for the whole walk, so of the four decoder pixels an output pixel's pool window touches
//     out(2y+1, 2x+1) = s00 + s01 + s10 + s11                 of (y, x)
//     out(2y,   2x+1) = (s00 + s01)(y, x) + (s10 + s11)(y-1, x)
//     out(2y+1, 2x  ) = (s00 + s10)(y, x) + (s01 + s11)(y, x-1)
//     out(2y,   2x  ) = s00(y, x) + s10(y-1, x) + s01(y, x-1) + s11(y-1, x-1)          (all x 0.25)
// the row above comes out of the thread's own registers (three partial sums per group carried from the previous row), the left
// neighbour's chunk is one 16-byte LDS read of the row the workgroup just staged, and its row-above part is carried too.  The
// eight results of a chunk go to four output pixels at channel z C3 + c3 (2-byte LDS scatter, 224 per decoder pixel instead of
// 896 gathers) and leave as whole 112-byte pixels: two output rows x 128 pixels per walk step, contiguous in HBM.  The next
// row's chunks are requested a step ahead.
constexpr int HS_TX = 64;  // decoder columns per strip
template <int K>           // chunks per thread and row = Cm / 8 (64 pixels x Cm / 2 chunks over 256 threads)
__global__ __launch_bounds__(256) void head_shuffle_fwd_strip_kernel(const bf16_t* __restrict__ dec, bf16_t* __restrict__ hin, int h,
                                                                    int w, int C3, int D, int rows_per_wg) {
  constexpr int NQ = 4 * K;            // chunks per decoder pixel
  constexpr int Cm = 8 * K, C4 = 32 * K;
  constexpr int PB = C4 * 2;           // bytes per decoder pixel
  constexpr int OB = Cm * 2;           // bytes per output pixel
  __shared__ __attribute__((aligned(16))) unsigned char rowb[(HS_TX + 1) * PB];        // staged decoder row, slot 0 = column x0 - 1
  __shared__ __attribute__((aligned(16))) unsigned char outb[2 * 2 * HS_TX * OB];      // two output rows of the strip
  const int tid = threadIdx.x;
  const int b = blockIdx.z, x0 = blockIdx.x * HS_TX;
  const int y0 = blockIdx.y * rows_per_wg, y1 = min(h, y0 + rows_per_wg);
  int xl[K], q[K];
  uint32_t cpo[K];  // byte offsets of the two groups' output channels (z C3 + c3) * 2, packed 16 | 16
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int i = tid + k * 256;
    xl[k] = i / NQ;
    q[k] = i - xl[k] * NQ;
    const int c0 = 2 * q[k], c1 = c0 + 1;
    const int o0 = ((c0 % D) * C3 + c0 / D) * 2, o1 = ((c1 % D) * C3 + c1 / D) * 2;
    cpo[k] = (uint32_t)o0 | ((uint32_t)o1 << 16);
  }
  const bool halo_thread = tid < NQ;  // also fetches chunk tid of column x0 - 1
  const size_t img = (size_t)b * h * w;
  auto request = [&](int y, vsx_u32x4 (&cur)[K], vsx_u32x4& hl) {
    const vsx_u32x4* src = reinterpret_cast<const vsx_u32x4*>(dec + (img + (size_t)y * w + x0) * C4);
#pragma unroll
    for (int k = 0; k < K; ++k) cur[k] = __builtin_nontemporal_load(src + tid + k * 256);
    hl = (vsx_u32x4){0u, 0u, 0u, 0u};
    if (halo_thread && x0 > 0) hl = __builtin_nontemporal_load(src - NQ + tid);
  };
  // carried from the row above, per chunk and group c: u1 = s10 + s11, u2 = s10 (own pixel), ul = s11 (left pixel)
  float u1[K][2], u2[K][2], ul[K][2];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int c = 0; c < 2; ++c) u1[k][c] = u2[k][c] = ul[k][c] = 0.f;
  vsx_u32x4 cur[K], hl;
  const int ys = y0 > 0 ? y0 - 1 : 0;  // the row above the strip only feeds the carries
  request(ys, cur, hl);
  for (int y = ys; y < y1; ++y) {
    // stage this row (every thread its own chunks; the halo column by the first NQ threads)
#pragma unroll
    for (int k = 0; k < K; ++k) *reinterpret_cast<vsx_u32x4*>(rowb + PB + (size_t)(tid + k * 256) * 16) = cur[k];
    if (halo_thread) *reinterpret_cast<vsx_u32x4*>(rowb + tid * 16) = hl;
    vsx_u32x4 me[K];
#pragma unroll
    for (int k = 0; k < K; ++k) me[k] = cur[k];
    __syncthreads();  // row staged; nobody still copies the previous output rows out (they finished before staging)
    if (y + 1 < y1) request(y + 1, cur, hl);
    const bool emit = y >= y0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const vsx_u32x4 lf = *reinterpret_cast<const vsx_u32x4*>(rowb + (size_t)xl[k] * PB + q[k] * 16);  // column x - 1, same chunk
      unsigned char* o11 = outb + ((size_t)(1 * 2 * HS_TX + 2 * xl[k] + 1)) * OB;
      unsigned char* o10 = o11 - OB;
      unsigned char* o01 = outb + ((size_t)(2 * xl[k] + 1)) * OB;
      unsigned char* o00 = o01 - OB;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const uint32_t w0 = me[k][2 * c], w1 = me[k][2 * c + 1], l0 = lf[2 * c], l1 = lf[2 * c + 1];
        const float s00 = bf16_bits_to_f32(w0 & 0xFFFFu), s01 = bf16_bits_to_f32(w0 >> 16);
        const float s10 = bf16_bits_to_f32(w1 & 0xFFFFu), s11 = bf16_bits_to_f32(w1 >> 16);
        const float l01 = bf16_bits_to_f32(l0 >> 16), l11 = bf16_bits_to_f32(l1 >> 16);
        const float top = s00 + s01, bot = s10 + s11;
        if (emit) {
          const float v11 = 0.25f * (top + bot);
          const float v01 = 0.25f * (top + u1[k][c]);
          const float v10 = 0.25f * ((s00 + s10) + (l01 + l11));
          const float v00 = 0.25f * ((s00 + u2[k][c]) + (l01 + ul[k][c]));
          const uint32_t off = c ? cpo[k] >> 16 : cpo[k] & 0xFFFFu;
          *reinterpret_cast<unsigned short*>(o11 + off) = (unsigned short)f32_to_bf16_bits(v11);
          *reinterpret_cast<unsigned short*>(o10 + off) = (unsigned short)f32_to_bf16_bits(v10);
          *reinterpret_cast<unsigned short*>(o01 + off) = (unsigned short)f32_to_bf16_bits(v01);
          *reinterpret_cast<unsigned short*>(o00 + off) = (unsigned short)f32_to_bf16_bits(v00);
        }
        u1[k][c] = bot;
        u2[k][c] = s10;
        ul[k][c] = l11;
      }
    }
    __syncthreads();  // output rows complete; every thread has read its left neighbours out of the staged row
    if (emit) {
      // rows 2y and 2y + 1 of the strip: 2 x 128 pixels x OB bytes, each row contiguous in HBM
      constexpr int CPO = 2 * HS_TX * OB / 16;  // chunks per output row of the strip
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        vsx_u32x4* dst = reinterpret_cast<vsx_u32x4*>(hin + (((size_t)b * 2 * h + 2 * y + r) * (2 * w) + 2 * x0) * Cm);
        for (int i = tid; i < CPO; i += 256) dst[i] = *reinterpret_cast<const vsx_u32x4*>(outb + (size_t)r * (2 * HS_TX * OB) + i * 16);
      }
    }
  }
}

// adjoint on the same strips: ddec[y, x, 4 cp' + 2 i + j] = 0.25 * sum_{a, b in {0, 1}} g[2y + i + a, 2x + j + b, z C3 + c3], g = dhin, zero
// beyond the image.  With the column-pair sums P_r0 = g[r, 2x] + g[r, 2x+1], P_r1 = g[r, 2x+1] + g[r, 2x+2] of the three output
// rows r = 2y, 2y+1, 2y+2:  ds00 = P_00 + P_10, ds01 = P_01 + P_11, ds10 = P_10 + P_20, ds11 = P_11 + P_21 (x 0.25) — row 2y + 2
// is row 2(y + 1) of the next step, so a step stages two output rows (+ the halo column), gathers 12 two-byte values per chunk
// and carries two sums per group; the chunk leaves straight from the registers (16 bytes per thread, contiguous over the strip).
template <int K>
__global__ __launch_bounds__(256) void head_shuffle_bwd_strip_kernel(const bf16_t* __restrict__ dhin, bf16_t* __restrict__ ddec, int h,
                                                                    int w, int C3, int D, int rows_per_wg) {
  constexpr int NQ = 4 * K;
  constexpr int Cm = 8 * K, C4 = 32 * K;
  constexpr int OB = Cm * 2;                      // bytes per output-resolution pixel
  constexpr int NPX = 2 * HS_TX + 1;              // staged pixels per output row (last = halo column 2 x0 + 128)
  constexpr int CPR = 2 * HS_TX * OB / 16;        // 16-byte chunks of an output row inside the strip (896 at Cm = 56)
  constexpr int KR = CPR / 256;                   // whole rounds of the 256 threads per row
  __shared__ __attribute__((aligned(16))) unsigned char gb[2 * NPX * OB];
  const int tid = threadIdx.x;
  const int b = blockIdx.z, x0 = blockIdx.x * HS_TX;
  const int y0 = blockIdx.y * rows_per_wg, y1 = min(h, y0 + rows_per_wg);
  const int H2 = 2 * h, W2 = 2 * w;
  int xl[K];
  uint32_t cpo[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int i = tid + k * 256;
    xl[k] = i / NQ;
    const int qq = i - xl[k] * NQ;
    const int c0 = 2 * qq, c1 = c0 + 1;
    const int o0 = ((c0 % D) * C3 + c0 / D) * 2, o1 = ((c1 % D) * C3 + c1 / D) * 2;
    cpo[k] = (uint32_t)o0 | ((uint32_t)o1 << 16);
  }
  constexpr int REM = CPR - KR * 256;             // leftover chunks of a row (threads tid < REM)
  constexpr int HC = OB / 16;                     // chunks of the halo pixel (threads tid < HC)
  // registers of one staged row pair: [row][KR + 1 (remainder) + 1 (halo)]
  auto request = [&](int y, vsx_u32x4 (&rg)[2][KR + 2]) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int Y = 2 * y + 1 + r;
      const bool okr = Y >= 0 && Y < H2;
      const vsx_u32x4* src = reinterpret_cast<const vsx_u32x4*>(dhin + (((size_t)b * H2 + (okr ? Y : 0)) * W2 + 2 * x0) * Cm);
#pragma unroll
      for (int k = 0; k < KR; ++k) rg[r][k] = okr ? __builtin_nontemporal_load(src + tid + k * 256) : (vsx_u32x4){0u, 0u, 0u, 0u};
      rg[r][KR] = (okr && REM > 0 && tid < REM) ? __builtin_nontemporal_load(src + tid + KR * 256) : (vsx_u32x4){0u, 0u, 0u, 0u};
      rg[r][KR + 1] = (okr && tid < HC && 2 * x0 + 2 * HS_TX < W2) ? __builtin_nontemporal_load(src + CPR + tid) : (vsx_u32x4){0u, 0u, 0u, 0u};
    }
  };
  float p0[K][2], p1[K][2];  // P_20 / P_21 of the previous step = P_00 / P_01 of this one
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int c = 0; c < 2; ++c) p0[k][c] = p1[k][c] = 0.f;
  vsx_u32x4 rg[2][KR + 2];
  request(y0 - 1, rg);
  const size_t img = (size_t)b * h * w;
  for (int y = y0 - 1; y < y1; ++y) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      unsigned char* dst = gb + (size_t)r * NPX * OB;
#pragma unroll
      for (int k = 0; k < KR; ++k) *reinterpret_cast<vsx_u32x4*>(dst + (size_t)(tid + k * 256) * 16) = rg[r][k];
      if (REM > 0 && tid < REM) *reinterpret_cast<vsx_u32x4*>(dst + (size_t)(tid + KR * 256) * 16) = rg[r][KR];
      if (tid < HC) *reinterpret_cast<vsx_u32x4*>(dst + (size_t)(CPR + tid) * 16) = rg[r][KR + 1];
    }
    __syncthreads();
    if (y + 1 < y1) request(y + 1, rg);
    const bool emit = y >= y0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      uint32_t ow[4];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const uint32_t off = c ? cpo[k] >> 16 : cpo[k] & 0xFFFFu;
        const unsigned char* r1 = gb + (size_t)(2 * xl[k]) * OB + off;   // row 2y + 1
        const unsigned char* r2 = r1 + (size_t)NPX * OB;                 // row 2y + 2
        const float a0 = bf16_bits_to_f32(*reinterpret_cast<const unsigned short*>(r1));
        const float a1 = bf16_bits_to_f32(*reinterpret_cast<const unsigned short*>(r1 + OB));
        const float a2 = bf16_bits_to_f32(*reinterpret_cast<const unsigned short*>(r1 + 2 * OB));
        const float b0 = bf16_bits_to_f32(*reinterpret_cast<const unsigned short*>(r2));
        const float b1 = bf16_bits_to_f32(*reinterpret_cast<const unsigned short*>(r2 + OB));
        const float b2 = bf16_bits_to_f32(*reinterpret_cast<const unsigned short*>(r2 + 2 * OB));
        const float P10 = a0 + a1, P11 = a1 + a2, P20 = b0 + b1, P21 = b1 + b2;
        const float d00 = 0.25f * (p0[k][c] + P10), d01 = 0.25f * (p1[k][c] + P11);
        const float d10 = 0.25f * (P10 + P20), d11 = 0.25f * (P11 + P21);
        ow[2 * c] = f32x2_to_bf16x2_bits(d00, d01);
        ow[2 * c + 1] = f32x2_to_bf16x2_bits(d10, d11);
        p0[k][c] = P20;
        p1[k][c] = P21;
      }
      if (emit) {
        vsx_u32x4* dst = reinterpret_cast<vsx_u32x4*>(ddec + (img + (size_t)y * w + x0) * C4);
        dst[tid + k * 256] = (vsx_u32x4){ow[0], ow[1], ow[2], ow[3]};
      }
    }
    __syncthreads();  // gathers done before the next pair of rows is staged
  }
}

/* K12: PixelToVoxelHead.upsample + reshape (viscy_models/components/heads.py:607-615,632-637).
 * dec: [B, h, w, 4*C3*D] → hin: [B, 2h, 2w, D*C3] with the depth axis outermost inside a pixel
 * (channel = z*C3 + c3), so the 3x3x3 head convolution reads contiguous channel slices per z. */
extern "C" int32_t vsx_head_shuffle_fwd(const void* dec, void* hin, int32_t B, int32_t h, int32_t w, int32_t C3,
                                        int32_t D, int32_t pool, int32_t dtype, vsx_stream_t stream) {
  int vn = dtype == VSX_BF16 ? 8 : 4;
  VSX_CHECK(dec && hin && B > 0 && h > 0 && w > 0 && C3 > 0 && D > 0, "vsx_head_shuffle_fwd: bad arguments");
  VSX_CHECK((C3 * D) % vn == 0, "vsx_head_shuffle_fwd: C3*D=%d must be a multiple of %d", C3 * D, vn);
  long total = (long)B * 4 * h * w * (C3 * D / vn);
  dim3 grid(vsx_cdiv(total, 256));
  if (dtype == VSX_BF16 && pool && (g_vsx_head_rows & 8) && C3 * D == 56 && w % HS_TX == 0 && B <= 65535) {
    // column strips (round 6); rows per workgroup: two workgroups per CU in one round when the batch allows it
    const long strips = (long)B * (w / HS_TX);
    long rpw = (long)h * strips / 512;
    rpw = rpw < 8 ? 8 : (rpw > h ? h : rpw);
    hipLaunchKernelGGL((head_shuffle_fwd_strip_kernel<7>), dim3(w / HS_TX, vsx_cdiv(h, rpw), B), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dec, (bf16_t*)hin, h, w, C3, D, (int)rpw);
    VSX_LAUNCH_CHECK();
    return 0;
  }
  if (C3 * D <= 64 && B <= 65535) {  // LDS-tiled permutation
    if (dtype == VSX_BF16)
      hipLaunchKernelGGL((head_shuffle_fwd_tiled_kernel<bf16_t, 8>), dim3(vsx_cdiv(w, 8), vsx_cdiv(h, 8), B), dim3(256), 0,
                         (hipStream_t)stream, (const bf16_t*)dec, (bf16_t*)hin, h, w, C3, D, pool);
    else
      hipLaunchKernelGGL((head_shuffle_fwd_tiled_kernel<float, 4>), dim3(vsx_cdiv(w, 4), vsx_cdiv(h, 4), B), dim3(256), 0,
                         (hipStream_t)stream, (const float*)dec, (float*)hin, h, w, C3, D, pool);
    VSX_LAUNCH_CHECK();
    return 0;
  }
  if (dtype == VSX_BF16)
    hipLaunchKernelGGL(head_shuffle_fwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dec,
                       (bf16_t*)hin, B, h, w, C3, D, pool);
  else
    hipLaunchKernelGGL(head_shuffle_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dec,
                       (float*)hin, B, h, w, C3, D, pool);
  VSX_LAUNCH_CHECK();
  return 0;
}
extern "C" int32_t vsx_head_shuffle_bwd(const void* dhin, void* ddec, int32_t B, int32_t h, int32_t w, int32_t C3,
                                        int32_t D, int32_t pool, int32_t dtype, vsx_stream_t stream) {
  int vn = dtype == VSX_BF16 ? 8 : 4;
  VSX_CHECK(dhin && ddec && B > 0 && h > 0 && w > 0 && C3 > 0 && D > 0, "vsx_head_shuffle_bwd: bad arguments");
  VSX_CHECK((4 * C3 * D) % vn == 0, "vsx_head_shuffle_bwd: 4*C3*D must be a multiple of %d", vn);
  long total = (long)B * h * w * (4 * C3 * D / vn);
  dim3 grid(vsx_cdiv(total, 256));
  if (dtype == VSX_BF16 && pool && (g_vsx_head_rows & 16) && C3 * D == 56 && w % HS_TX == 0 && B <= 65535) {
    const long strips = (long)B * (w / HS_TX);
    long rpw = (long)h * strips / 512;
    rpw = rpw < 8 ? 8 : (rpw > h ? h : rpw);
    hipLaunchKernelGGL((head_shuffle_bwd_strip_kernel<7>), dim3(w / HS_TX, vsx_cdiv(h, rpw), B), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dhin, (bf16_t*)ddec, h, w, C3, D, (int)rpw);
    VSX_LAUNCH_CHECK();
    return 0;
  }
  if (C3 * D <= 64 && B <= 65535) {  // LDS-tiled permutation
    if (dtype == VSX_BF16)
      hipLaunchKernelGGL((head_shuffle_bwd_tiled_kernel<bf16_t, 8>), dim3(vsx_cdiv(w, 8), vsx_cdiv(h, 8), B), dim3(256), 0,
                         (hipStream_t)stream, (const bf16_t*)dhin, (bf16_t*)ddec, h, w, C3, D, pool);
    else
      hipLaunchKernelGGL((head_shuffle_bwd_tiled_kernel<float, 4>), dim3(vsx_cdiv(w, 4), vsx_cdiv(h, 4), B), dim3(256), 0,
                         (hipStream_t)stream, (const float*)dhin, (float*)ddec, h, w, C3, D, pool);
    VSX_LAUNCH_CHECK();
    return 0;
  }
  if (dtype == VSX_BF16)
    hipLaunchKernelGGL(head_shuffle_bwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dhin,
                       (bf16_t*)ddec, B, h, w, C3, D, pool);
  else
    hipLaunchKernelGGL(head_shuffle_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dhin,
                       (float*)ddec, B, h, w, C3, D, pool);
  VSX_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ PixelToVoxelShuffleHead (FCMAE head, heads.py:656-685)
// out[b, co, z, Y, X] = pool(v)(Y, X),  v(Y, X) = feat[b, Y/s, X/s, ((co*D + z)*s + Y%s)*s + X%s]   (nn.PixelShuffle(s))
// pool = MONAI SubpixelUpsample pad-pool: ConstantPad2d((s-1, 0, s-1, 0)) + AvgPool2d(s, stride 1)
//      = mean of v over the s x s block ending at (Y, X), zeros outside.  feat: [B*h*w, Cout*D*s*s] (T), out: fp32 NCDHW.
template <typename T>
__global__ __launch_bounds__(256) void voxel_shuffle_fwd_kernel(const T* __restrict__ feat, float* __restrict__ out, int B, int h,
                                                                int w, int Cout, int D, int s, int pool) {
  const int H = h * s, W = w * s;
  const int Cd = Cout * D * s * s;
  const long total = (long)B * Cout * D * H * W;
  const float inv = pool ? 1.f / (float)(s * s) : 1.f;
  const int nt = pool ? s : 1;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int X = (int)(i % W);
    long r = i / W;
    const int Y = (int)(r % H); r /= H;
    const int z = (int)(r % D); r /= D;
    const int co = (int)(r % Cout);
    const int b = (int)(r / Cout);
    const int cb = (co * D + z) * s * s;
    float acc = 0.f;
    for (int ty = 0; ty < nt; ++ty) {
      const int yy = Y - ty;
      if (yy < 0) break;
      for (int tx = 0; tx < nt; ++tx) {
        const int xx = X - tx;
        if (xx < 0) break;
        acc += to_f32<T>(feat[(((size_t)b * h + yy / s) * w + xx / s) * Cd + cb + (yy % s) * s + (xx % s)]);
      }
    }
    out[i] = acc * inv;
  }
}
// dfeat[b, y, x, (ch*s + dy)*s + dx] = inv * sum_{ty, tx < s} dout[b, ch, s*y + dy + ty, s*x + dx + tx]   (inside the image)
template <typename T>
__global__ __launch_bounds__(256) void voxel_shuffle_bwd_kernel(const float* __restrict__ dout, T* __restrict__ dfeat, int B,
                                                                int h, int w, int Cout, int D, int s, int pool) {
  constexpr int VN = VT<T>::N;
  const int H = h * s, W = w * s;
  const int Cd = Cout * D * s * s;
  const int nch = Cd / VN;
  const long total = (long)B * h * w * nch;
  const float inv = pool ? 1.f / (float)(s * s) : 1.f;
  const int nt = pool ? s : 1;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ch = (int)(i % nch);
    const long pix = i / nch;
    const int x = (int)(pix % w);
    const long r = pix / w;
    const int y = (int)(r % h);
    const int b = (int)(r / h);
    float o[VN];
#pragma unroll
    for (int j = 0; j < VN; ++j) {
      const int c = ch * VN + j;
      const int dx = c % s, dy = (c / s) % s, cz = c / (s * s);  // cz = co*D + z
      const float* plane = dout + ((size_t)b * Cout * D + cz) * H * W;
      const int Y = y * s + dy, X = x * s + dx;
      float acc = 0.f;
      for (int ty = 0; ty < nt && Y + ty < H; ++ty)
        for (int tx = 0; tx < nt && X + tx < W; ++tx) acc += plane[(size_t)(Y + ty) * W + X + tx];
      o[j] = acc * inv;
    }
    stvec<T>(dfeat + (size_t)pix * Cd + ch * VN, pack<T>(o));
  }
}

/* FCMAE head: viscy_models.components.heads.PixelToVoxelShuffleHead (heads.py:656-685) = MONAI UpSample(pixelshuffle,
 * scale s, pre_conv None, apply_pad_pool) + reshape to (B, Cout, D, s*h, s*w).  feat [B*h*w, Cout*D*s*s] dtype -> out fp32. */
extern "C" int32_t vsx_voxel_shuffle_fwd(const void* feat, float* out, int32_t B, int32_t h, int32_t w, int32_t Cout, int32_t D,
                                         int32_t s, int32_t pool, int32_t dtype, vsx_stream_t stream) {
  VSX_CHECK(feat && out && B > 0 && h > 0 && w > 0 && Cout > 0 && D > 0 && s > 0, "vsx_voxel_shuffle_fwd: bad arguments");
  long total = (long)B * Cout * D * h * s * w * s;
  int g = vsx_cdiv(total, 256);
  if (g > 65536) g = 65536;
  if (dtype == VSX_BF16)
    hipLaunchKernelGGL(voxel_shuffle_fwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)feat, out, B, h,
                       w, Cout, D, s, pool);
  else
    hipLaunchKernelGGL(voxel_shuffle_fwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)feat, out, B, h, w,
                       Cout, D, s, pool);
  VSX_LAUNCH_CHECK();
  return 0;
}
extern "C" int32_t vsx_voxel_shuffle_bwd(const float* dout, void* dfeat, int32_t B, int32_t h, int32_t w, int32_t Cout, int32_t D,
                                         int32_t s, int32_t pool, int32_t dtype, vsx_stream_t stream) {
  int vn = dtype == VSX_BF16 ? 8 : 4;
  VSX_CHECK(dout && dfeat && B > 0 && h > 0 && w > 0 && Cout > 0 && D > 0 && s > 0, "vsx_voxel_shuffle_bwd: bad arguments");
  VSX_CHECK((Cout * D * s * s) % vn == 0, "vsx_voxel_shuffle_bwd: Cout*D*s*s=%d must be a multiple of %d", Cout * D * s * s, vn);
  long total = (long)B * h * w * (Cout * D * s * s / vn);
  int g = vsx_cdiv(total, 256);
  if (g > 65536) g = 65536;
  if (dtype == VSX_BF16)
    hipLaunchKernelGGL(voxel_shuffle_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, dout, (bf16_t*)dfeat, B, h, w,
                       Cout, D, s, pool);
  else
    hipLaunchKernelGGL(voxel_shuffle_bwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, dout, (float*)dfeat, B, h, w, Cout,
                       D, s, pool);
  VSX_LAUNCH_CHECK();
  return 0;
}
