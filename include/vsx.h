/* vsx.h — C-ABI of libvsx.so: hand-written gfx950 (MI355X / CDNA4) kernels for the VisCy
 * UNeXt2 virtual-staining hot path (forward, backward, MixedLoss, normalisation, AdamW).
 *
 * The reference (mehta-lab/VisCy) has no FFI: every device op on this path is an ATen/cuDNN
 * call reached through torch.nn / timm / MONAI.  Each entry point
 * below names the reference op(s) it replaces (paths relative to /root/reference/packages).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to a contiguous buffer owned by the caller
 *     (PyTorch caching allocator: tensor.data_ptr()); the library never allocates, frees,
 *     synchronises or changes device; kernels are enqueued on `stream`
 *     (torch.cuda.current_stream().cuda_stream) and are hipGraph-capturable.
 *   - `dtype`: VSX_F32 (0) = fp32 storage + exact-fp32 MFMA (parity mode),
 *              VSX_BF16 (1) = bf16 storage + bf16 MFMA, fp32 accumulation / statistics.
 *   - activations inside the trunk are channels-last: [B, H, W, C] ("pixel rows x channels").
 *   - return 0 on success; non-zero → vsx_last_error() (thread-local) has the message.
 *   - re-entrant, no global mutable state besides read-only tables and vsx_set_flag knobs.
 */
#ifndef VSX_H
#define VSX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vsx_stream_t; /* hipStream_t */

#define VSX_F32 0
#define VSX_BF16 1

int32_t vsx_version(void);
const char* vsx_last_error(void);
/* debug / A-B knobs: "tn_tr" (1 = ds_read_b64_tr_b16 fragments in the wgrad GEMM, default 1) */
int32_t vsx_set_flag(const char* name, int32_t value);
int32_t vsx_get_flag(const char* name);

/* ---------------------------------------------------------------------------------------------
 * Operand "gather" description shared by the two GEMM kernels.  Row m of the logical
 * [M, K] operand is a pixel (b, y, x) of a gh x gw grid; K = ntaps * cs.
 *   VSX_A_ROWS   : plain row-major, K contiguous per row (ntaps = 1)
 *   VSX_A_PATCH2 : 2x2 stride-2 patch of a [B, 2gh, 2gw, cs] tensor  (k = (ky, kx, c))
 *   VSX_A_CONV3  : 3x3 zero-padded neighbourhood of a [B, gh, gw, *] tensor, cs channels
 *                  starting at coff                                   (k = (ky, kx, c))
 * --------------------------------------------------------------------------------------------- */
#define VSX_A_ROWS 0
#define VSX_A_PATCH2 1
#define VSX_A_CONV3 2

#define VSX_PRO_NONE 0
#define VSX_PRO_GRN 1 /* a = g * s[b, k] + beta[k]   (GRN applied on the fly to the stored activation g = gelu(h)) */

#define VSX_EPI_NONE 0
#define VSX_EPI_BIAS 1         /* c = acc + bias[n] */
#define VSX_EPI_BIAS_GELU_SQ 2 /* c = h = acc + bias; c2 = g = gelu(h); red0[b, n] += g^2     (fc1 + GRN pass A) */
#define VSX_EPI_BIAS_RES 3     /* c = acc + bias[n] + res[m, n]   (bias may be NULL)      (fc2 + residual)   */
#define VSX_EPI_DZ 4           /* c = dz = acc; red0[b, n] += dz * aux[m, n] (aux = g); red1[b, n] += dz  (fc2 dgrad) */
#define VSX_EPI_BIAS_STATS 5   /* c = acc + bias; red0[b, n] += c; red1[b, n] += c^2      (head conv + IN)   */

typedef struct VsxGemm {
  /* C[M, N] = pro(A)[M, K] * B[N, K]^T  (vsx_gemm_nt)   |   W[N, K] += X[M, N]^T * pro(A)[M, K]  (vsx_gemm_tn) */
  const void* A; /* NT: left operand; TN: the "A-like" (gathered) operand Y[M, K] */
  const void* B; /* NT: weights [N, K], K contiguous, row stride ldb; TN: X[M, N], row stride ldb */
  void* C;       /* NT: output (dtype); TN: fp32 accumulation target [N, ldc] (atomicAdd) */
  int32_t M, N, K;
  int32_t lda, ldb, ldc;
  int32_t a_mode, gh, gw, cs;
  int32_t nz;           /* z-batch count (gridDim.z), 1 if unused */
  int32_t a_coff[8];    /* per-z channel offset into A rows */
  int32_t b_off[8];     /* per-z offset: NT → element offset into B; TN → column offset into X rows */
  int32_t c_coff[8];    /* per-z: NT → column offset into C rows; TN → element offset into W */
  int32_t c_mode;       /* NT only: VSX_A_ROWS or VSX_A_PATCH2 (scatter C rows back to 2x2 patches, n = (ky,kx,c)) */
  int32_t c_cs;         /* channels per tap for c_mode PATCH2 */
  int32_t pro;
  const float* grn_s;   /* [nb, K] */
  const float* grn_b;   /* [K] */
  int32_t hw;           /* rows per batch sample (b = m / hw) */
  int32_t epi;
  const float* bias;    /* [N] */
  const void* res;      /* [M, ldr] dtype */
  int32_t ldr;
  const void* aux;      /* [M, ldx] dtype (EPI_DZ: stored activation g) */
  int32_t ldx;
  float* red0;
  float* red1;
  float* colsum;        /* TN only: [N] += sum_m X[m, n] (bias gradient), may be NULL */
  void* C2;             /* NT, EPI_BIAS_GELU_SQ: second output g = gelu(h), same layout as C */
} VsxGemm;

/* K5/K8/K9/K11/K13 (pointwise / patch / 3x3 convolutions as MFMA GEMMs) — replaces
 * nn.Linear / 1x1 nn.Conv2d inside timm GlobalResponseNormMlp, the 2x2-s2 downsample conv,
 * the decoder 1x1 projection (viscy_models/components/blocks.py:54-74) and the head Conv3d
 * (viscy_models/components/heads.py:617-625). */
int32_t vsx_gemm_nt(const VsxGemm* p, int32_t dtype, vsx_stream_t stream);
/* weight-gradient GEMM (contraction over pixels) for the same layers */
int32_t vsx_gemm_tn(const VsxGemm* p, int32_t dtype, vsx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VSX_H */
