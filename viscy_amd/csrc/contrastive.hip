// DynaCLR contrastive path: the small tail behind the ConvNeXt trunk and the NT-Xent loss.
//   global average pool      timm NormMlpClassifierHead.global_pool (viscy_models/contrastive/encoder.py:93-99,125-128)
//   BatchNorm1d (+ ReLU)     projection MLP, encoder.py:115-121
//   NT-Xent / NT-Xent-HCL    viscy_models/contrastive/loss.py:20-186 on pytorch-metric-learning's pair semantics
// Every tensor here is tiny next to the trunk's activations ([B, 768] features, a [2B, 2B] similarity matrix): the kernels
// are written for exactness (fp32, two-pass statistics, deterministic row-owned reductions — no atomics), not for a roofline.
#include "vsx_common.h"
#include "../../include/vsx.h"

// ------------------------------------------------------------------ global average pool over the hw rows of each sample
template <typename T>
__global__ __launch_bounds__(256) void avgpool_rows_fwd_kernel(const T* __restrict__ x, float* __restrict__ out, int hw, int C) {
  const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const T* xb = x + (size_t)b * hw * C + c;
  float s = 0.f;
  for (int r = 0; r < hw; ++r) s += to_f32<T>(xb[(size_t)r * C]);
  out[(size_t)b * C + c] = s / (float)hw;
}
template <typename T>
__global__ __launch_bounds__(256) void avgpool_rows_bwd_kernel(const float* __restrict__ dout, T* __restrict__ dx, int hw, int C) {
  const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const T v = from_f32<T>(dout[(size_t)b * C + c] / (float)hw);
  T* db = dx + (size_t)b * hw * C + c;
  for (int r = 0; r < hw; ++r) db[(size_t)r * C] = v;
}

extern "C" int32_t vsx_avgpool_rows_fwd(const void* x, float* out, int32_t B, int32_t hw, int32_t C, int32_t dtype,
                                        vsx_stream_t stream) {
  VSX_CHECK(x && out && B > 0 && hw > 0 && C > 0 && B <= 65535, "vsx_avgpool_rows_fwd: bad arguments");
  dim3 g(vsx_cdiv(C, 256), B);
  if (dtype == VSX_BF16)
    hipLaunchKernelGGL(avgpool_rows_fwd_kernel<bf16_t>, g, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, out, hw, C);
  else
    hipLaunchKernelGGL(avgpool_rows_fwd_kernel<float>, g, dim3(256), 0, (hipStream_t)stream, (const float*)x, out, hw, C);
  VSX_LAUNCH_CHECK();
  return 0;
}
extern "C" int32_t vsx_avgpool_rows_bwd(const float* dout, void* dx, int32_t B, int32_t hw, int32_t C, int32_t dtype,
                                        vsx_stream_t stream) {
  VSX_CHECK(dout && dx && B > 0 && hw > 0 && C > 0 && B <= 65535, "vsx_avgpool_rows_bwd: bad arguments");
  dim3 g(vsx_cdiv(C, 256), B);
  if (dtype == VSX_BF16)
    hipLaunchKernelGGL(avgpool_rows_bwd_kernel<bf16_t>, g, dim3(256), 0, (hipStream_t)stream, dout, (bf16_t*)dx, hw, C);
  else
    hipLaunchKernelGGL(avgpool_rows_bwd_kernel<float>, g, dim3(256), 0, (hipStream_t)stream, dout, (float*)dx, hw, C);
  VSX_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ BatchNorm1d over the batch axis of [B, F] (+ fused ReLU)
// one thread owns a feature: lanes run along F (coalesced rows), the loop runs over the batch.
__global__ __launch_bounds__(256) void bn1d_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* __restrict__ rmean,
                                                       float* __restrict__ rvar, float* __restrict__ y,
                                                       float* __restrict__ smean, float* __restrict__ srstd, int B, int F,
                                                       float eps, float momentum, int training, int relu) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= F) return;
  float mean, var;
  if (training) {
    float s = 0.f;
    for (int r = 0; r < B; ++r) s += x[(size_t)r * F + f];
    mean = s / (float)B;
    float q = 0.f;
    for (int r = 0; r < B; ++r) {
      const float d = x[(size_t)r * F + f] - mean;
      q = fmaf(d, d, q);
    }
    var = q / (float)B;
    // running statistics (torch: unbiased variance, momentum 0.1)
    rmean[f] = (1.f - momentum) * rmean[f] + momentum * mean;
    rvar[f] = (1.f - momentum) * rvar[f] + momentum * (B > 1 ? q / (float)(B - 1) : var);
  } else {
    mean = rmean[f];
    var = rvar[f];
  }
  const float rstd = 1.0f / sqrtf(var + eps);
  smean[f] = mean;
  srstd[f] = rstd;
  const float g = w[f] * rstd, sh = b[f] - mean * g;
  for (int r = 0; r < B; ++r) {
    float v = fmaf(x[(size_t)r * F + f], g, sh);
    if (relu) v = fmaxf(v, 0.f);
    y[(size_t)r * F + f] = v;
  }
}
__global__ __launch_bounds__(256) void bn1d_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                       const float* __restrict__ y, const float* __restrict__ w,
                                                       const float* __restrict__ smean, const float* __restrict__ srstd,
                                                       float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db,
                                                       int B, int F, int training, int relu) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= F) return;
  const float mean = smean[f], rstd = srstd[f];
  float sb = 0.f, sg = 0.f;
  for (int r = 0; r < B; ++r) {
    const size_t i = (size_t)r * F + f;
    const float d = (relu && y[i] <= 0.f) ? 0.f : dy[i];
    sb += d;
    sg = fmaf(d, (x[i] - mean) * rstd, sg);
  }
  dw[f] += sg;
  db[f] += sb;
  const float g = w[f] * rstd, mb = sb / (float)B, mg = sg / (float)B;
  for (int r = 0; r < B; ++r) {
    const size_t i = (size_t)r * F + f;
    const float d = (relu && y[i] <= 0.f) ? 0.f : dy[i];
    dx[i] = training ? g * (d - mb - (x[i] - mean) * rstd * mg) : g * d;
  }
}

extern "C" int32_t vsx_bn1d_fwd(const float* x, const float* w, const float* b, float* running_mean, float* running_var, float* y,
                                float* save_mean, float* save_rstd, int32_t B, int32_t F, float eps, float momentum,
                                int32_t training, int32_t relu, vsx_stream_t stream) {
  VSX_CHECK(x && w && b && running_mean && running_var && y && save_mean && save_rstd && B > 0 && F > 0, "vsx_bn1d_fwd: bad arguments");
  VSX_CHECK(!training || B > 1, "vsx_bn1d_fwd: Expected more than 1 value per channel when training, got input size [%d, %d]", B, F);
  hipLaunchKernelGGL(bn1d_fwd_kernel, dim3(vsx_cdiv(F, 256)), dim3(256), 0, (hipStream_t)stream, x, w, b, running_mean, running_var,
                     y, save_mean, save_rstd, B, F, eps, momentum, training, relu);
  VSX_LAUNCH_CHECK();
  return 0;
}
extern "C" int32_t vsx_bn1d_bwd(const float* dy, const float* x, const float* y, const float* w, const float* save_mean,
                                const float* save_rstd, float* dx, float* dw, float* db, int32_t B, int32_t F, int32_t training,
                                int32_t relu, vsx_stream_t stream) {
  VSX_CHECK(dy && x && y && w && save_mean && save_rstd && dx && dw && db && B > 0 && F > 0, "vsx_bn1d_bwd: bad arguments");
  hipLaunchKernelGGL(bn1d_bwd_kernel, dim3(vsx_cdiv(F, 256)), dim3(256), 0, (hipStream_t)stream, dy, x, y, w, save_mean, save_rstd,
                     dx, dw, db, B, F, training, relu);
  VSX_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ NT-Xent (+ hard-negative concentration)
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// En = E / max(|E|, 1e-12) (F.normalize), inv[i] = 1 / max(|E_i|, 1e-12);   one block per row
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ E, float* __restrict__ En,
                                                          float* __restrict__ inv, int D) {
  __shared__ float red[4];
  const float* e = E + (size_t)blockIdx.x * D;
  float q = 0.f;
  for (int k = threadIdx.x; k < D; k += 256) q = fmaf(e[k], e[k], q);
  q = block_sum(q, red);
  const float iv = 1.0f / fmaxf(sqrtf(q), 1e-12f);
  if (threadIdx.x == 0) inv[blockIdx.x] = iv;
  for (int k = threadIdx.x; k < D; k += 256) En[(size_t)blockIdx.x * D + k] = e[k] * iv;
}

// S[i, j] = <En_i, En_j>;   one block per row i, En_i staged in LDS
__global__ __launch_bounds__(256) void sim_rows_kernel(const float* __restrict__ En, float* __restrict__ S, int N, int D) {
  extern __shared__ float ei[];
  const int i = blockIdx.x;
  for (int k = threadIdx.x; k < D; k += 256) ei[k] = En[(size_t)i * D + k];
  __syncthreads();
  for (int j = threadIdx.x; j < N; j += 256) {
    const float* ej = En + (size_t)j * D;
    float s = 0.f;
    for (int k = 0; k < D; ++k) s = fmaf(ei[k], ej[k], s);
    S[(size_t)i * N + j] = s;
  }
}

// one block per anchor a.  P(a) = {p != a : label_p == label_a},  Nn(a) = {n : label_n != label_a}
//   w_n = K exp(beta s_an) / sum_n exp(beta s_an), K = |Nn(a)|                       (== 1 for beta == 0)
//   loss(a,p) = -log( e^{s_ap/T - m} / (sum_n w_n e^{s_an/T - m} + e^{s_ap/T - m}) + tiny ),  m = max(s_ap/T, max_n s_an/T)
// rowsum[a] = sum_p loss(a,p), rowcnt[a] = |P(a)|, dS[a,:] = d rowsum[a] / d S[a,:]  (unscaled by the pair count)
__global__ __launch_bounds__(256) void ntxent_rows_kernel(const float* __restrict__ S, const int* __restrict__ labels,
                                                          float* __restrict__ rowsum, float* __restrict__ rowcnt,
                                                          float* __restrict__ dS, int N, float invT, float beta) {
  __shared__ float red[4];
  const int a = blockIdx.x;
  const float* s = S + (size_t)a * N;
  float* d = dS + (size_t)a * N;
  const int la = labels[a];
  // negatives: count, max, HCL normaliser
  float cnt = 0.f, mx = -3.0e38f, z = 0.f;
  for (int j = threadIdx.x; j < N; j += 256) {
    if (labels[j] != la) {
      cnt += 1.f;
      mx = fmaxf(mx, s[j] * invT);
      z += __expf(beta * s[j]);
    }
  }
  const float K = block_sum(cnt, red);
  const float mneg = block_max(mx, red);
  const float Z = fmaxf(block_sum(z, red), 1e-8f);
  for (int j = threadIdx.x; j < N; j += 256) d[j] = 0.f;
  __syncthreads();
  float lsum = 0.f, lcnt = 0.f;
  if (K == 0.f) {  // an anchor without negatives: its positive pairs count with zero loss (log(1 + tiny)), as in pml
    for (int p = 0; p < N; ++p)
      if (p != a && labels[p] == la) lcnt += 1.f;
  } else {
    for (int p = 0; p < N; ++p) {  // positives of a (block-uniform loop)
      if (p == a || labels[p] != la) continue;
      const float sp = s[p] * invT;
      const float m = fmaxf(sp, mneg);
      float part = 0.f;
      for (int j = threadIdx.x; j < N; j += 256)
        if (labels[j] != la) {
          const float w = beta != 0.f ? K * __expf(beta * s[j]) / Z : 1.f;
          part += w * __expf(s[j] * invT - m);
        }
      const float dneg = block_sum(part, red);
      const float num = __expf(sp - m);
      const float den = dneg + num;
      lsum += -logf(num / den + 1.17549435e-38f);
      lcnt += 1.f;
      // gradient of this pair's loss w.r.t. row a of S
      for (int j = threadIdx.x; j < N; j += 256) {
        if (j == p) {
          d[j] += -invT * (1.f - num / den);
        } else if (labels[j] != la) {
          const float w = beta != 0.f ? K * __expf(beta * s[j]) / Z : 1.f;
          const float e = __expf(s[j] * invT - m);
          float gneg = w * e * (invT + beta);
          if (beta != 0.f) gneg -= beta * w * dneg / K;
          d[j] += gneg / den;
        }
      }
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) {
    rowsum[a] = lsum;
    rowcnt[a] = lcnt;
  }
}

// loss = sum(rowsum) / sum(rowcnt);  acc = {loss, pair count}
__global__ __launch_bounds__(256) void ntxent_finalize_kernel(const float* __restrict__ rowsum, const float* __restrict__ rowcnt,
                                                              float* __restrict__ acc, int N) {
  __shared__ float red[4];
  float a = 0.f, c = 0.f;
  for (int j = threadIdx.x; j < N; j += 256) {
    a += rowsum[j];
    c += rowcnt[j];
  }
  a = block_sum(a, red);
  c = block_sum(c, red);
  if (threadIdx.x == 0) {
    acc[0] = c > 0.f ? a / c : 0.f;
    acc[1] = c;
  }
}

// dE_i = gout / pairs * inv_i * (G_i - En_i <En_i, G_i>),  G_i = sum_j (dS[i,j] + dS[j,i]) En_j;   one block per row
__global__ __launch_bounds__(256) void ntxent_bwd_rows_kernel(const float* __restrict__ dS, const float* __restrict__ En,
                                                              const float* __restrict__ inv, const float* __restrict__ acc,
                                                              const float* __restrict__ gout, float* __restrict__ dE, int N,
                                                              int D) {
  extern __shared__ float coef[];  // N
  __shared__ float red[4];
  const int i = blockIdx.x;
  for (int j = threadIdx.x; j < N; j += 256) coef[j] = dS[(size_t)i * N + j] + dS[(size_t)j * N + i];
  __syncthreads();
  const float scale = acc[1] > 0.f ? gout[0] / acc[1] : 0.f;
  float dot = 0.f;
  // thread k owns feature k (D <= a few hundred): two sweeps, the second after the projection term is known
  for (int k = threadIdx.x; k < D; k += 256) {
    float g = 0.f;
    for (int j = 0; j < N; ++j) g = fmaf(coef[j], En[(size_t)j * D + k], g);
    dE[(size_t)i * D + k] = g;  // parked
    dot = fmaf(g, En[(size_t)i * D + k], dot);
  }
  dot = block_sum(dot, red);
  const float iv = inv[i] * scale;
  for (int k = threadIdx.x; k < D; k += 256) dE[(size_t)i * D + k] = iv * (dE[(size_t)i * D + k] - En[(size_t)i * D + k] * dot);
}

extern "C" int32_t vsx_ntxent_fwd(const float* E, const int32_t* labels, float* En, float* inv, float* S, float* dS, float* rows,
                                  float* acc, int32_t N, int32_t D, float temperature, float beta, vsx_stream_t stream) {
  VSX_CHECK(E && labels && En && inv && S && dS && rows && acc && N > 1 && D > 0, "vsx_ntxent_fwd: bad arguments");
  VSX_CHECK(temperature > 0.f, "vsx_ntxent_fwd: temperature must be positive");
  VSX_CHECK(D * sizeof(float) <= 48 * 1024, "vsx_ntxent_fwd: embedding dimension %d too large", D);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3(N), dim3(256), 0, s, E, En, inv, D);
  hipLaunchKernelGGL(sim_rows_kernel, dim3(N), dim3(256), D * sizeof(float), s, En, S, N, D);
  hipLaunchKernelGGL(ntxent_rows_kernel, dim3(N), dim3(256), 0, s, S, labels, rows, rows + N, dS, N, 1.f / temperature, beta);
  hipLaunchKernelGGL(ntxent_finalize_kernel, dim3(1), dim3(256), 0, s, rows, rows + N, acc, N);
  VSX_LAUNCH_CHECK();
  return 0;
}
extern "C" int32_t vsx_ntxent_bwd(const float* dS, const float* En, const float* inv, const float* acc, const float* gout,
                                  float* dE, int32_t N, int32_t D, vsx_stream_t stream) {
  VSX_CHECK(dS && En && inv && acc && gout && dE && N > 1 && D > 0, "vsx_ntxent_bwd: bad arguments");
  VSX_CHECK(N * sizeof(float) <= 48 * 1024, "vsx_ntxent_bwd: %d embeddings exceed the row kernel's LDS budget (12288)", N);
  hipLaunchKernelGGL(ntxent_bwd_rows_kernel, dim3(N), dim3(256), N * sizeof(float), (hipStream_t)stream, dS, En, inv, acc, gout, dE,
                     N, D);
  VSX_LAUNCH_CHECK();
  return 0;
}
