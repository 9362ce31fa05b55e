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
 *   - re-entrant; the only mutable global state is the set of kernel-selection flags behind vsx_set_flag (plain ints read at
 *     launch time: set them before launching from several threads, not while launches are in flight) and the thread-local
 *     error string.
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
/* kernel template the last vsx_gemm_nt / vsx_gemm_tn call on this thread dispatched to: "gemm_nt_fast", "gemm_nt_generic",
 * "gemm_nt2" (both kernels of gemm_nt2.hip), "gemm_tn_fast", "gemm_tn_generic" ("" before the first call).  Measurement aid:
 * bench.py groups its live launch timings by kernel family with it. */
const char* vsx_last_kernel(void);
/* debug / A-B knobs: "tn_tr" (1 = ds_read_b64_tr_b16 fragments in the wgrad GEMM, default 1) */
int32_t vsx_set_flag(const char* name, int32_t value);
int32_t vsx_get_flag(const char* name);
/* vsx_set_flag("det_reduce", 1): the forward's per-sample sums (GRN sum of squares of the fused GRN-MLP passes and of gemm_nt's
 * GELU epilogue on the 256-row-tile kernel, InstanceNorm sum / sum of squares of vsx_head_conv_fwd) are formed in a fixed order
 * — per-workgroup partials in this caller-owned scratch (`floats` fp32 values, thread-local, used by the NEXT launches of this
 * thread; every launch checks the size and names what it needs), then one ordered pass — instead of by fp32 atomics, whose order
 * differs from run to run (timm GlobalResponseNorm / nn.InstanceNorm3d reduce deterministically on the CPU: reference behaviour
 * restated at viscy_models/unet/fcmae.py:174-221, components/heads.py:617-627).  NULL / 0 releases the scratch. */
int32_t vsx_det_workspace(float* ws, int64_t floats);

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
#define VSX_EPI_LN_BWD 6       /* c = rstd[m] * (d - mean_n(d) - xh * mean_n(d * xh)),  d = acc rounded to the storage type:
                                * the backward of an affine-free LayerNorm over the N columns applied to the GEMM result (fc1 data
                                * gradient + block LayerNorm backward in one launch).  aux = xh [M, ldx] (the normalised rows),
                                * grn_s = rstd [M] (fp32).  bf16, plain row operands, N <= 256, M % 256 == 0, K % 32 == 0 only
                                * (vsx_gemm_nt_ln_bwd_supported).  With grn_b = mean [M] non-NULL: aux holds the UN-normalised rows y
                                * (xh = bf16((y - mean) * rstd) is re-formed in the epilogue) and A is expected row-scaled by rstd
                                * (vsx_mlp_bwd_dh_ln), so c = d - mean_n(d) - xh * mean_n(d * xh) without the leading factor. */

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
  void* C2;             /* NT, EPI_BIAS_GELU_SQ: second output g = gelu(h), same layout as C (C itself may be NULL then:
                           inference keeps only the activation) */
  int64_t b_bstride;    /* NT: element stride between PER-SAMPLE weight matrices B[b] (b = m / hw); 0 = one shared B.
                         * Needs hw % 128 == 0, plain row operands, N > 64, K % 32 == 0 (the lean instantiation). Used to
                         * fold the GRN scale into fc2: a·W2^T with a = g·s[b] + beta  ==  g·(W2·diag(s[b]))^T + W2·beta
                         * TN: element stride between PER-SAMPLE OUTPUT matrices C[b] (and colsum rows [b][N]): one product
                         * X_b^T·Y_b per sample of hw rows, plain stores (bf16, no prologue, hw % 64 == 0).  The block backward
                         * derives the fc2 weight gradient and the GRN statistics from these (vsx_grn_q_reduce) */
  const float* rscale;  /* NT, EPI_BIAS_RES: per-sample scale of the branch, c = (acc + bias) * rscale[m / hw] + res — stochastic
                         * depth (timm DropPath: 0 or 1 / keep_prob per sample), NULL = 1.  Needs hw > 0. */
} VsxGemm;

/* K5/K8/K9/K11/K13 (pointwise / patch / 3x3 convolutions as MFMA GEMMs) — replaces
 * nn.Linear / 1x1 nn.Conv2d inside timm GlobalResponseNormMlp, the 2x2-s2 downsample conv,
 * the decoder 1x1 projection (viscy_models/components/blocks.py:54-74) and the head Conv3d
 * (viscy_models/components/heads.py:617-625). */
int32_t vsx_gemm_nt(const VsxGemm* p, int32_t dtype, vsx_stream_t stream);
/* 1 if vsx_gemm_nt takes VSX_EPI_LN_BWD for this shape (otherwise: a plain vsx_gemm_nt followed by vsx_ln_bwd) */
int32_t vsx_gemm_nt_ln_bwd_supported(int64_t M, int32_t N, int32_t K, int32_t dtype);
/* weight-gradient GEMM (contraction over pixels) for the same layers */
int32_t vsx_gemm_tn(const VsxGemm* p, int32_t dtype, vsx_stream_t stream);


/* =============================================================================================
 * Remaining entry points (one per fused kernel of the path).  `ws` arguments are caller-owned fp32
 * scratch for two-stage reductions (per-block partials + a fold kernel): same-address global atomics
 * serialise at ~0.2 us each on MI355X, so no kernel issues more than a few dozen per address.
 * ============================================================================================= */

/* K3 depthwise 7x7 conv — timm ConvNeXtBlock.conv_dw = nn.Conv2d(C, C, 7, padding=3, groups=C) (block math restated in-repo at
 * viscy_models/unet/fcmae.py:174-221).  x,y: [B,H,W,C] channels-last; w: tap-major fp32 [49][C]; y = conv(x) + bias [+ add]. */
int32_t vsx_dwconv7_fwd(const void* x, const float* w, const float* bias, const void* add, void* y,
    int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, vsx_stream_t stream);

/* data gradient of K3: dx = conv(dy, flipped w) [+ add]; add = gradient arriving through the residual branch. */
int32_t vsx_dwconv7_bwd_data(const void* dy, const float* w, const void* add, void* dx, int32_t B, int32_t H,
    int32_t W, int32_t C, int32_t dtype, vsx_stream_t stream);

/* weight/bias gradient of K3 accumulated into dw[49][C], db[C] (fp32).  ws: caller-owned workspace of ws_rows*50*C floats. */
int32_t vsx_dwconv7_bwd_weight(const void* dy, const void* x, float* dw, float* db, float* ws,
    int32_t ws_rows, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, vsx_stream_t stream);

/* K12 direct: the head's 3x3x3 Conv3d (MONAI Convolution conv, viscy_models/components/heads.py:607-616) as LDS-tiled MFMA
 * kernels for the production shape — bf16, C3 = 8 -> Cmid = 32, 5 output planes, H2 and W2 multiples of 16
 * (vsx_head_conv_supported tells; everything else runs through vsx_gemm_nt / vsx_gemm_tn with VSX_A_CONV3).
 * hin [B*H2*W2, 7*8], U / dU [B*H2*W2, 5*32], Wc [32][27*8] (k = ((dy*3+dx)*3+dz)*8 + c, the prepared conv weight),
 * ssum / ssq [B,32] += InstanceNorm statistics of the stored U, dW fp32 [32][216] += , db fp32 [32] += (may be NULL),
 * Wp: 45*2*64*8 bf16 packed by vsx_head_conv_dgrad_prep from Wc. */
int32_t vsx_head_conv_supported(int32_t H2, int32_t W2, int32_t c3, int32_t cmid, int32_t zo, int32_t dtype);
int32_t vsx_head_conv_fwd(const void* hin, const void* Wc, const float* bias, void* U, float* ssum, float* ssq,
    int32_t B, int32_t H2, int32_t W2, int32_t c3, int32_t cmid, int32_t zo, int32_t dtype, vsx_stream_t stream);
int32_t vsx_head_conv_wgrad(const void* hin, const void* dU, float* dW, float* db, int32_t B, int32_t H2, int32_t W2,
    int32_t c3, int32_t cmid, int32_t zo, int32_t dtype, vsx_stream_t stream);
int32_t vsx_head_conv_dgrad_prep(const void* Wc, void* Wp, int32_t dtype, vsx_stream_t stream);
int32_t vsx_head_conv_dgrad(const void* dU, const void* Wp, void* dhin, int32_t B, int32_t H2, int32_t W2, int32_t c3,
    int32_t cmid, int32_t zo, int32_t dtype, vsx_stream_t stream);

/* FCMAE head: viscy_models.components.heads.PixelToVoxelShuffleHead (heads.py:656-685) = MONAI UpSample(pixelshuffle, scale s,
 * pre_conv None, apply_pad_pool) + reshape: feat [B*h*w, Cout*D*s*s] (dtype) <-> out / dout fp32 (B, Cout, D, s*h, s*w). */
int32_t vsx_voxel_shuffle_fwd(const void* feat, float* out, int32_t B, int32_t h, int32_t w, int32_t Cout, int32_t D,
    int32_t s, int32_t pool, int32_t dtype, vsx_stream_t stream);
int32_t vsx_voxel_shuffle_bwd(const float* dout, void* dfeat, int32_t B, int32_t h, int32_t w, int32_t Cout, int32_t D,
    int32_t s, int32_t pool, int32_t dtype, vsx_stream_t stream);

/* out[m, :] = x[m, :] * scale[m / hw]  (rows of C elements, dtype): the gradient of a stochastic-depth branch
 * (timm DropPath in the ConvNeXt blocks, `drop_path_rate` / `encoder_drop_path_rate` of the reference models). */
int32_t vsx_scale_rows_samples(const void* x, const float* scale, void* out, int64_t M, int32_t C, int32_t hw, int32_t dtype,
    vsx_stream_t stream);

/* FCMAE masked pre-training.  masked_patchify / masked_unpatchify / `x *= unmasked`
 * (viscy_models/unet/fcmae.py:95-141,216-226) on channels-last row matrices, as one row permutation:
 *   dst[r,:] = map[r] >= 0 ? src[map[r],:] (+ add[r,:] when add != NULL) : 0        r < n_out; rows of C elements (dtype)
 * gather: map = dense row per kept token; scatter: map = compact row per dense row or -1 (zero fill); mask: map[r] = r or -1. */
int32_t vsx_rows_select(const void* src, const int32_t* map, const void* add, void* dst, int64_t n_out, int32_t C,
    int32_t dtype, vsx_stream_t stream);

/* cytoland.engine.MaskedMSELoss.forward (applications/cytoland/src/cytoland/engine.py:104-125):
 *   loss = sum(mask * mean_z (pred - orig)^2) / sum(mask);  pred / orig fp32 (B,C,Z,H,W), mask uint8 (B,1,H,W), H*W % 4 == 0.
 * acc: 2 floats of scratch kept for the backward {sum mask*(p-o)^2, sum(mask)}; dpred = gout * d loss / d pred. */
int32_t vsx_masked_mse_fwd(const float* pred, const float* orig, const uint8_t* mask, float* acc, float* loss, int32_t B,
    int32_t C, int32_t Z, int64_t HW, vsx_stream_t stream);
int32_t vsx_masked_mse_bwd(const float* pred, const float* orig, const uint8_t* mask, const float* acc, const float* gout,
    float* dpred, int32_t B, int32_t C, int32_t Z, int64_t HW, vsx_stream_t stream);

/* ConvNeXt-V1 layer scale (timm ConvNeXtBlock.gamma, ls_init_value 1e-6; the `convnext_tiny` trunk of
 * viscy_models.contrastive.ContrastiveEncoder, encoder.py:93-99) folded into the block's second pointwise layer:
 * Ws = diag(gamma) W [R,K], bs = gamma * b; unfold ADDS dW += diag(gamma) dWs, db += gamma dbs,
 * dgamma += rowsum(dWs * W) + dbs * b.  fp32. */
int32_t vsx_layer_scale_fold(const float* W, const float* b, const float* gamma, float* Ws, float* bs, int32_t R, int32_t K,
    vsx_stream_t stream);
int32_t vsx_layer_scale_unfold(const float* dWs, const float* dbs, const float* W, const float* b, const float* gamma,
    float* dW, float* db, float* dgamma, int32_t R, int32_t K, vsx_stream_t stream);

/* DynaCLR contrastive path: tail of viscy_models.contrastive.ContrastiveEncoder
 * (packages/viscy-models/src/viscy_models/contrastive/encoder.py:93-154) behind the ConvNeXt trunk.
 * Global average pool of a channels-last feature map x [B*hw, C] (dtype) -> out [B, C] fp32 (timm head.global_pool) and
 * its transpose. */
int32_t vsx_avgpool_rows_fwd(const void* x, float* out, int32_t B, int32_t hw, int32_t C, int32_t dtype, vsx_stream_t stream);
int32_t vsx_avgpool_rows_bwd(const float* dout, void* dx, int32_t B, int32_t hw, int32_t C, int32_t dtype, vsx_stream_t stream);
/* nn.BatchNorm1d on [B, F] fp32 (projection MLP, encoder.py:115-121), optional fused ReLU.  training != 0: batch
 * statistics, running_{mean,var} updated in place (momentum, unbiased variance); else the running statistics normalise.
 * save_{mean,rstd} [F] feed the backward, which ADDS into dw / db. */
int32_t vsx_bn1d_fwd(const float* x, const float* w, const float* b, float* running_mean, float* running_var, float* y,
    float* save_mean, float* save_rstd, int32_t B, int32_t F, float eps, float momentum, int32_t training, int32_t relu,
    vsx_stream_t stream);
int32_t vsx_bn1d_bwd(const float* dy, const float* x, const float* y, const float* w, const float* save_mean,
    const float* save_rstd, float* dx, float* dw, float* db, int32_t B, int32_t F, int32_t training, int32_t relu,
    vsx_stream_t stream);
/* viscy_models.contrastive.loss.NTXentLoss / NTXentHCL (loss.py:20-186; pytorch-metric-learning pair semantics):
 * E [N, D] fp32 embeddings, labels [N]; equal labels = positive pairs, different labels = negatives; cosine similarity;
 * beta = 0: NT-Xent, beta > 0: hard-negative re-weighting.  Scratch owned by the caller: En [N,D], inv [N], S [N,N],
 * dS [N,N], rows [2N]; acc [2] = {loss, number of positive pairs}.  vsx_ntxent_bwd: dE = gout * d loss / d E. */
int32_t vsx_ntxent_fwd(const float* E, const int32_t* labels, float* En, float* inv, float* S, float* dS, float* rows, float* acc,
    int32_t N, int32_t D, float temperature, float beta, vsx_stream_t stream);
int32_t vsx_ntxent_bwd(const float* dS, const float* En, const float* inv, const float* acc, const float* gout, float* dE,
    int32_t N, int32_t D, vsx_stream_t stream);

/* K13 (norm+act) + K14: MONAI Convolution ADN (InstanceNorm3d eps 1e-5 → PReLU) → nn.Conv3d(mid, 4*out, 1) → transpose /
 * nn.PixelShuffle(2) / transpose (viscy_models/components/heads.py:617-625,638-641).  U: [B,H2,W2,Z,Cmid] conv output;
 * ssum/ssq: [B,Cmid] from the conv GEMM epilogue (VSX_EPI_BIAS_STATS); out: (B, Cout, Z, 2*H2, 2*W2) fp32. */
int32_t vsx_head_out_fwd(const void* U, const float* ssum, const float* ssq, const float* w2,
    const float* b2, const float* alpha, float* out, int32_t B, int32_t H2, int32_t W2, int32_t Z, int32_t Cmid,
    int32_t Cout, float eps, int32_t dtype, vsx_stream_t stream);

/* backward pass 1 of the head tail: writes act = PReLU(IN(U)) and dv (gathered output gradient) for the 1x1x1 weight-gradient GEMM;
 * accumulates S1 = Σ dn, S2 = Σ dn·n̂ per (b, channel) and the PReLU slope gradient. */
int32_t vsx_head_out_bwd1(const void* U, const float* ssum, const float* ssq, const float* w2,
    const float* alpha, const float* dout, void* act, void* dv, float* S1, float* S2, float* dalpha, int32_t B,
    int32_t H2, int32_t W2, int32_t Z, int32_t Cmid, int32_t Cout, float eps, int32_t dtype,
    vsx_stream_t stream);

/* backward pass 1 with the weight gradient of the 1x1x1 convolution folded in (bf16 only; replaces vsx_head_out_bwd1 +
 * the vsx_gemm_tn over act/dv = autograd of nn.Conv3d(mid, 4*out, 1), heads.py:617-625): act is never materialised, the
 * contraction over the voxels runs on MFMA inside the kernel.  dW2 [4*Cout, Cmid] and db2 [4*Cout] are ACCUMULATED into;
 * scratch: B * (4*Cout*Cmid + 4*Cout) floats (zeroed here).  dv, S1, S2, dalpha as vsx_head_out_bwd1. */
int32_t vsx_head_out_bwd1_wgrad(const void* U, const float* ssum, const float* ssq, const float* w2,
    const float* alpha, const float* dout, void* dv, float* S1, float* S2, float* dalpha, float* dW2, float* db2,
    float* scratch, int32_t B, int32_t H2, int32_t W2, int32_t Z, int32_t Cmid, int32_t Cout, float eps, int32_t dtype,
    vsx_stream_t stream);

/* backward pass 2: dU = rstd * (dn - S1/cnt - n̂ * S2/cnt)  (InstanceNorm3d backward). */
int32_t vsx_head_out_bwd2(const void* U, const float* ssum, const float* ssq, const float* w2,
    const float* alpha, const void* dv, const float* S1, const float* S2, void* dU, int32_t B, int32_t H2,
    int32_t W2, int32_t Z, int32_t Cmid, int32_t Cout, float eps, int32_t dtype, vsx_stream_t stream);

/* avg_pool3d(·,(1,2,2)) of preds/target (viscy_utils/evaluation/metrics.py:340-341), target.max() of the INPUT planes (metrics.py:298) and the
 * L1 / L2 sums of MixedLoss (viscy_utils/losses/mixed_loss.py:58-63) in one pass.  tmax must be pre-set to -inf. */
int32_t vsx_loss_pool(const float* P, const float* T, float* Po, float* To, float* tmax, float* l1sum,
    float* l2sum, int32_t planes, int32_t H, int32_t W, vsx_stream_t stream);

/* one scale of ssim_25d (metrics.py:174-305): per-sample sums of the SSIM and contrast-sensitivity maps, bf16 rounding points as in
 * _compute_ssim_and_cs_bf16 (metrics.py:243-255). */
int32_t vsx_ssim_scale_fwd(const float* P, const float* T, const float* tmax, float* sum_ssim, float* sum_cs,
    int32_t B, int32_t C, int32_t D, int32_t H, int32_t W, vsx_stream_t stream);

/* backward of one scale (+ pooled-gradient chain from the next scale, + L1/L2 gradient at scale 0); gout: device scalar (upstream gradient) or NULL = 1. */
int32_t vsx_ssim_scale_bwd(const float* P, const float* T, const float* tmax, const float* coef, float* dmu,
    const float* dPnext, float* dP, int32_t B, int32_t C, int32_t D, int32_t H, int32_t W, float l1c, float l2c,
    const float* gout, int32_t has_ssim, vsx_stream_t stream);

/* ms_ssim_25d combination with clamp(min=1e-4) (metrics.py:326-349) + MixedLoss weights (mixed_loss.py:56-69); writes loss, MS-SSIM and the per-(scale, sample) map-pixel gradients. */
/* Training variant of the two calls above with one pass over the stack less: vsx_ssim_scale_fwd_dmu = the sums of
 * vsx_ssim_scale_fwd plus the UNSCALED gradient field dmu [3][B*C][H-10][W-10] of this scale (last != 0: the SSIM map, else
 * the contrast map — what ms_ssim_25d uses of it, metrics.py:326-349), kept by the caller; vsx_ssim_scale_bwd_in = the
 * second half of vsx_ssim_scale_bwd, applying this scale's per-sample coef [B][2] (vsx_loss_finalize) to the stored field. */
int32_t vsx_ssim_scale_fwd_dmu(const float* P, const float* T, const float* tmax, float* sum_ssim, float* sum_cs, float* dmu,
                               int32_t B, int32_t C, int32_t D, int32_t H, int32_t W, int32_t last, vsx_stream_t stream);
int32_t vsx_ssim_scale_bwd_in(const float* P, const float* T, const float* dmu, const float* coef, const float* dPnext,
                              float* dP, int32_t B, int32_t C, int32_t D, int32_t H, int32_t W, float l1c, float l2c,
                              const float* gout, int32_t has_ssim, int32_t last, vsx_stream_t stream);
/* One-pass training forward of a scale: vsx_ssim_scale_fwd_dmu + the vsx_loss_pool pass of the NEXT scale (pooled stacks Po / To,
 * their data range tmax_next, and with l1sum != NULL this scale's L1 / L2 sums) from the same reads of P / T; tmax[0] must be
 * final (vsx_loss_tmax for scale 0).  tmax, tmax_next, l1sum, l2sum are SLOTTED here: VSX_LOSS_SLOTS partial values,
 * VSX_LOSS_SLOT_STRIDE floats apart (range = their max, sums = their sum; vsx_loss_finalize takes sum_slots = VSX_LOSS_SLOTS).
 * Reference: metrics.py:272-305 + 340-341, mixed_loss.py:58-63. */
#define VSX_LOSS_SLOTS 64        /* partial accumulators per scalar of the one-pass forward (same-address atomics serialise) */
#define VSX_LOSS_SLOT_STRIDE 32  /* floats between two slots (one 128-byte line each) */
int32_t vsx_loss_tmax(const float* T, int64_t n, float* tmax, vsx_stream_t stream);
int32_t vsx_ssim_scale_fwd_fused(const float* P, const float* T, const float* tmax, float* sum_ssim, float* sum_cs, float* dmu,
                                 float* Po, float* To, float* tmax_next, float* l1sum, float* l2sum, int32_t B, int32_t C,
                                 int32_t D, int32_t H, int32_t W, int32_t last, vsx_stream_t stream);
int32_t vsx_loss_finalize(const float* sum_ssim, const float* sum_cs, const float* l1sum, const float* l2sum,
    const float* npix, float nelem, int32_t B, int32_t nscale, int32_t sum_slots, float a1, float a2, float a3,
    const float* gout, float* loss, float* coef, float* ms_out, vsx_stream_t stream);  /* sum_slots: 0 / 1 = l1sum, l2sum are scalars (vsx_loss_pool); VSX_LOSS_SLOTS = slotted (vsx_ssim_scale_fwd_fused) */

/* K2/K4: timm LayerNorm2d / nn.LayerNorm over channels, eps 1e-6 (reached via timm ConvNeXtStage / ConvNeXtBlock at viscy_models/unet/unext2.py:79,
 * viscy_models/components/blocks.py:60-69).  gamma == NULL → no affine (the block LN's affine is folded into fc1). */
int32_t vsx_ln_fwd(const void* x, void* y, float* mean, float* rstd, const float* gamma, const float* beta,
    int32_t rows, int32_t C, float eps, int32_t dtype, vsx_stream_t stream);

/* LayerNorm backward: x̂ = mean ? (x-mean)*rstd : x; dx = rstd*(g - mean(g) - x̂*mean(g*x̂)) [+ add], g = dy*gamma; dgamma/dbeta accumulated. */
int32_t vsx_ln_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
    const void* add, void* dx, float* dgamma, float* dbeta, int32_t rows, int32_t C, int32_t dtype,
    vsx_stream_t stream);

/* K7: timm GlobalResponseNorm statistics: s[b,n] = 1 + gamma[n]*g/(mean_n g + eps), g = sqrt(colsq[b,n]); colsq comes from the fc1 GEMM epilogue. */
int32_t vsx_grn_scale(const float* colsq, const float* gamma, float* s, int32_t nb, int32_t N, float eps,
    vsx_stream_t stream);

/* backward of the GRN statistics path: P[b,n] = Σ_hw dz*g, Sb[b,n] = Σ_hw dz → t[b,n] (factor of g in dG), dgamma, dbeta
 * (both ADDED to).  rowst: nb*N floats of scratch (per-sample dgamma contributions, column-reduced by a second launch).
 * dgamma == NULL (then Sb, dbeta NULL too): only t and rowst are produced, the caller reduces rowst / its Sb over the samples
 * itself (VSX_WTASK_REDUCE_ROWS jobs of vsx_weight_tasks). */
int32_t vsx_grn_bwd_stats(const float* colsq, const float* P, const float* Sb, const float* gamma, float* t,
    float* dgamma, float* dbeta, float* rowst, int32_t nb, int32_t N, float eps, vsx_stream_t stream);

/* K6/K7 backward, elementwise pass: dh = (dz*s + gelu(h)*t) * gelu'(h) written over dz; colsum[n] += Σ_m dh.  ws: ws_rows*N floats. */
int32_t vsx_grn_gelu_bwd(void* dz, const void* h, const float* s, const float* t, float* colsum, float* ws,
    int32_t ws_rows, int32_t M, int32_t N, int32_t hw, int32_t dtype, vsx_stream_t stream);

/* K26: torch.optim.AdamW step (viscy_utils/optimizers.py:50) on flat fp32 buffers; hyper = device array
 * {lr, beta1, beta2, eps, weight_decay, 1-beta1^t, 1-beta2^t, grad_scale} so the launch is hipGraph-replayable. */
int32_t vsx_adamw(float* p, const float* g, float* m, float* v, const float* hyper, int64_t n,
    vsx_stream_t stream);

/* Fused GRN-MLP of a ConvNeXt-V2 block (timm ConvNeXtBlock as called from viscy_models/unet/unext2.py:79; math restated at
 * viscy_models/unet/fcmae.py:174-221), bf16: the 4C-wide hidden activation never leaves the CU (csrc/mlp.hip).
 *   vsx_mlp_supported : 1 when a fused instantiation exists for (C, pixels per sample hw, rows M, dtype)
 *   vsx_mlp_image_bytes / vsx_mlp_pack : fragment-major LDS image of the prepared weights W1' [4C, C] (LayerNorm affine folded)
 *                       and W2 [C, 4C] (both bf16, row-major as vsx_prep_weight writes them)
 *   vsx_mlp_fwd mode 0: colsq[b, 4C] += sum_hw gelu(fc1(xh))^2   (GRN statistics; nothing else is stored)
 *               mode 1: out = res + rscale[b] * (fc2(gelu(fc1(xh)) * s[b] + beta) + b2) */
int32_t vsx_mlp_supported(int32_t C, int32_t hw, int64_t M, int32_t dtype);   /* the inference pair (modes 0 and 1) */
int32_t vsx_mlp_mode_supported(int32_t C, int32_t hw, int64_t M, int32_t mode, int32_t dtype);  /* one pass (mode 0..7) */
int64_t vsx_mlp_image_bytes(int32_t C);
int32_t vsx_mlp_pack(const void* W1, const void* W2, void* img, int32_t C, vsx_stream_t stream);
/* vsx_mlp_fwd / vsx_mlp_fc1 with the block LayerNorm (eps, no affine) applied in the kernel's prologue: y = the UN-normalised
 * rows (output of the depthwise convolution).  Modes 0 / 1: the normalised rows never exist in memory; vsx_mlp_fc1_ln also
 * writes them (xh_out [M, C]) and rstd_out [M] for the backward — or, with xh_out = NULL and mean_out [M] non-NULL, only the
 * two row statistics (the backward then re-normalises y: vsx_mlp_bwd_dh_ln).  Replaces vsx_ln_fwd + vsx_mlp_fwd / vsx_mlp_fc1. */
int32_t vsx_mlp_fwd_ln(const void* y, float eps, const void* wimg, const float* b1, const float* grn_s, const float* grn_b,
    const float* b2, const void* res, const float* rscale, void* out, float* colsq, const float* gtab, int64_t M, int32_t C,
    int32_t hw, int32_t mode, int32_t dtype, vsx_stream_t stream);
int32_t vsx_mlp_fc1_ln(const void* y, float eps, void* xh_out, float* rstd_out, float* mean_out, const void* wimg, const float* b1,
    float* colsq, const float* gtab, void* h, void* g, int64_t M, int32_t C, int32_t hw, int32_t dtype, vsx_stream_t stream);
int32_t vsx_mlp_fwd(const void* xh, const void* wimg, const float* b1, const float* grn_s, const float* grn_b, const float* b2,
    const void* res, const float* rscale, void* out, float* colsq, const float* gelu_table, int64_t M, int32_t C, int32_t hw,
    int32_t mode, int32_t dtype, vsx_stream_t stream);
/* training fc1 on the same kernel: h = bf16(xh . W1'^T + b1) and g = bf16(gelu(h)) stored ([M, 4C] each), colsq[b, 4C] +=
 * sum_hw g^2 — the outputs of vsx_gemm_nt with VSX_EPI_BIAS_GELU_SQ */
int32_t vsx_mlp_fc1(const void* xh, const void* wimg, const float* b1, float* colsq, const float* gelu_table, void* h, void* g,
    int64_t M, int32_t C, int32_t hw, int32_t dtype, vsx_stream_t stream);
/* GRN statistics and fc2 weight gradient from the per-sample products Q[b] = dout_b^T . g_b ([C, 4C] fp32 each, vsx_gemm_tn
 * with b_bstride) and the per-sample column sums cs[b][C] of dout — dz = dout . W2 is linear in dout, so
 *   P[b, j] = sum_hw dz*g = sum_c W2[c, j] * Q[b, c, j]          S[b, j] = sum_hw dz = sum_c W2[c, j] * cs[b, c]
 *   dW2[c, j] += sum_b s[b, j] * Q[b, c, j] + beta[j] * sum_b cs[b, c]        db2[c] += sum_b cs[b, c]
 * (W2 = the bf16 GEMM operand [C, 4C]): neither dz nor a pass over the 4C-wide activations is needed for the statistics. */
int32_t vsx_grn_q_reduce(const float* Q, const float* cs, const void* W2, const float* s, const float* beta, float* P, float* S,
    float* dW2, float* db2, float* ws, int64_t ws_floats, int32_t nb, int32_t C, int32_t dtype, vsx_stream_t stream);
/* floats of caller-owned scratch vsx_grn_q_reduce needs (Q is read ONCE: the per-sample-group partials of dW2 pass through it) */
int64_t vsx_grn_q_reduce_ws_floats(int32_t nb, int32_t C);
/* Block backward without a stored dz (csrc/mlp.hip MODE 3 / 4; wimg = vsx_mlp_pack with W2^T [4C, C] in the place of W1'):
 *   vsx_mlp_bwd_stats: P[b, 4C] += sum_hw dz * g, S[b, 4C] += sum_hw dz with dz = bf16(dout . W2) recomputed tile by tile —
 *                      what vsx_gemm_nt(VSX_EPI_DZ) accumulates, minus its 4C-wide output
 *   vsx_mlp_bwd_dh   : dh = (dz * s[b] + gelu(h) * t[b]) * gelu'(h) stored [M, 4C] (dz recomputed), colsum[4C] += sum of dh over
 *                      all rows through the caller-owned workspace ws [ws_rows >= M / vsx_mlp_rows_per_workgroup, 4C] —
 *                      vsx_gemm_nt(VSX_EPI_DZ) + vsx_grn_gelu_bwd with ONE 4C-wide write instead of two */
int32_t vsx_mlp_bwd_stats(const void* dout, const void* wimg, const void* g, float* P, float* S, int64_t M, int32_t C, int32_t hw,
    int32_t dtype, vsx_stream_t stream);
int32_t vsx_mlp_bwd_dh(const void* dout, const void* wimg, const void* h, const float* s, const float* t, void* dh, float* ws,
                       int64_t ws_rows, float* colsum, const float* gelu_table, int64_t M, int32_t C, int32_t hw, int32_t dtype,
                       vsx_stream_t stream);
int32_t vsx_mlp_rows_per_workgroup(int32_t C, int32_t hw, int64_t M);
/* vsx_mlp_bwd_dh WITHOUT a stored pre-activation (csrc/mlp.hip MODE 5; round 4): h = bf16(xh . W1'^T + b1) is recomputed on chip
 * from the normalised rows xh [M, C] — wimg_fwd = vsx_mlp_pack(W1', W2) and b1 are the operands vsx_mlp_fc1 ran with, so the
 * recomputed h is bit-identical to the one it would have stored — beside dz = dout . W2 (wimg_bwd as for vsx_mlp_bwd_dh).  With
 * vsx_mlp_fc1 / vsx_mlp_fc1_ln called with h = NULL (MODE 6: only g is stored) the 4C-wide h never exists in memory: one 4C-wide
 * write less in the forward, one 4C-wide read less in the backward, per block.  Available where
 * vsx_mlp_mode_supported(.., 5, ..) / (.., 6, ..) (C <= 224, `mlp_fused` bit 6).  Reference math: timm GlobalResponseNormMlp as
 * restated in viscy_models/unet/fcmae.py:174-221. */
int32_t vsx_mlp_bwd_dh_re(const void* dout, const void* xh, const void* wimg_bwd, const void* wimg_fwd, const float* b1, const float* s,
                          const float* t, void* dh, float* ws, int64_t ws_rows, float* colsum, const float* gelu_table, int64_t M,
                          int32_t C, int32_t hw, int32_t dtype, vsx_stream_t stream);
/* vsx_mlp_bwd_dh_re for a block whose forward stored NO normalised rows (csrc/mlp.hip MODE 7; vsx_mlp_fc1_ln with xh_out = NULL):
 * y [M, C] = the LayerNorm input (depthwise output), mean / rstd [M] = its row statistics; x^ = bf16((y - mean) * rstd) is re-formed
 * on chip as the forward formed it.  The pass writes dh' = dh * rstd (row-scaled) — what the consumers want once they read y, too:
 *   fc1 weight gradient  dh^T . x^ = dh'^T . y - u (x) 1  with u[j] = sum_r dh'[r, j] * mean[r]   (plain vsx_gemm_tn on y; then
 *                        vsx_unprep_grad(.., rowsub = u))
 *   LayerNorm backward   dy = dx' - mean_c(dx') - x^ * mean_c(dx' * x^), dx' = dh' . W1'   (vsx_gemm_nt VSX_EPI_LN_BWD with
 *                        aux = y, grn_s = rstd, grn_b = mean: the trailing "* rstd" is already in dx')
 * colsum2 [2, 4C] += { sum_r dh[r, j] (the fc1 bias gradient), u[j] }; ws [ws_rows >= M / vsx_mlp_rows_per_workgroup, 2 * 4C].
 * Available where vsx_mlp_mode_supported(.., 7, ..) (C <= 224, `mlp_fused` bits 5, 6 and 7).  Reference math: the LayerNorm
 * -> Linear pair of timm's ConvNeXtBlock (norm, mlp.fc1) as restated in viscy_models/unet/fcmae.py:174-221. */
int32_t vsx_mlp_bwd_dh_ln(const void* dout, const void* y, const float* mean, const float* rstd, const void* wimg_bwd,
                          const void* wimg_fwd, const float* b1, const float* s, const float* t, void* dh, float* ws, int64_t ws_rows,
                          float* colsum2, const float* gelu_table, int64_t M, int32_t C, int32_t hw, int32_t dtype,
                          vsx_stream_t stream);
/* GELU of a bf16 value through a table (csrc/mlp.hip): vsx_mlp_gelu_table fills `tab` (vsx_mlp_gelu_table_len() = 2 N floats)
 * with r(a) = a * Phi(-a) (first N) and d(a) = Phi(a) + a * phi(a) - 1/2 (second N) for every bf16 magnitude a in [2^-24, 16):
 * gelu(h) = max(h, 0) - r(|h|), gelu'(h) = 1/2 + sign(h) * d(|h|).  The forward passes read the first half, the dh passes both. */
int32_t vsx_mlp_gelu_table_len(void);
int32_t vsx_mlp_gelu_table(float* tab, vsx_stream_t stream);

/* Device-side half of the optimiser schedule (viscy_utils/optimizers.py:50-61: AdamW + MONAI WarmupCosineSchedule stepped per
 * batch): reads cfg = {base_lr, beta1, beta2, eps, weight_decay, grad_scale, schedule (0 constant | 1 warm-up cosine),
 * warmup_steps, t_total, warmup_multiplier, cycles} and the int32 step counter, writes the 8 scalars vsx_adamw reads for
 * THIS step and increments the counter.  No host memory is touched: a captured step replays correctly however far the
 * host runs ahead. */
int32_t vsx_adamw_advance(const double* cfg, int32_t* step, float* hyper, vsx_stream_t stream);

/* p[0 .. n) = value (16-byte aligned p): `optimizer.zero_grad()` on the flat gradient buffer, the zero-filled reduction
 * targets of a pass, start values of running maxima — ordinary kernel nodes in the captured step. */
int32_t vsx_fill_f32(float* p, int64_t n, float value, vsx_stream_t stream);

/* fp32 parameter viewed as [R, Cs, Tn] (out, in, taps) → GEMM operand dst [R, Tn*Cs] and/or dstT [Tn*Cs, R] in `dtype`, optionally scaled per
 * input channel by gamma (LayerNorm fold).  tapmode 1 = head Conv3d tap order (kz,ky,kx) → (ky,kx,kz). */
int32_t vsx_prep_weight(const float* src, void* dst, void* dstT, const float* gamma, int32_t R, int32_t Cs,
    int32_t Tn, int32_t tapmode, int32_t dtype, vsx_stream_t stream);

/* inverse of vsx_prep_weight for gradients: dparam[r][c][t] += g'[r][k]*gamma[c] + u[r]*beta[c]; dgamma[c] += Σ g'*W, with
 * g'[r][k] = g[r][k] - rowsub[r] (rowsub may be NULL: the rank-1 term of a weight gradient that was contracted with un-centred
 * rows, see vsx_mlp_bwd_dh_ln). */
int32_t vsx_unprep_grad(const float* g, float* dparam, const float* gamma, const float* W, float* dgamma,
    const float* u, const float* beta, const float* rowsub, int32_t R, int32_t Cs, int32_t Tn, int32_t tapmode,
    vsx_stream_t stream);

/* fold of the GRN affine into the fc2 weights (timm GlobalResponseNorm inside GlobalResponseNormMlp, between act and fc2):
 * out[b][r][k] = dtype(W[r][k] * s[b][k]) for b < B; W fp32 [R][K], s fp32 [B][K].  Pairs with VsxGemm.b_bstride = R*K. */
int32_t vsx_scale_weight_samples(const float* W, const float* s, void* out, int32_t B, int32_t R, int32_t K,
    int32_t dtype, vsx_stream_t stream);

/* out[r] = (b ? b[r] : 0) + Σ_c W[r][c]*v[c]   (fold LayerNorm beta into the fc1 bias). */
int32_t vsx_matvec(const float* W, const float* v, const float* b, float* out, int32_t R, int32_t C,
    vsx_stream_t stream);

/* out[c] += Σ_r W[r][c]*u[r]   (gradient of the folded LayerNorm beta). */
int32_t vsx_matvec_t_add(const float* W, const float* u, float* out, int32_t R, int32_t C,
    vsx_stream_t stream);

/* dst[j][i] (+)= src[i][j]  (depthwise weights [C][49] <-> [49][C]). */
int32_t vsx_transpose_f32(const float* src, float* dst, int32_t A, int32_t Bn, int32_t accumulate,
    vsx_stream_t stream);

/* A list of independent weight-space jobs in one launch per VSX_WTASK_MAX tasks (the step refreshes ~200 prepared operands
 * per weight update; as single launches each leaves the chip idle for 4 - 7 us).  kind selects which single-op entry point
 * the task stands for; fields as documented at vsx_weight_tasks in csrc/optim.hip:
 *   PREP: p0 src, p1 dst, p2 dstT, p3 gamma, i0 R, i1 Cs, i2 Tn, i3 tapmode, dtype;  TRANSPOSE: p0 src, p1 dst, i0 A, i1 Bn,
 *   i2 accumulate;  MATVEC: p0 W, p3 v, p2 b, p1 out, i0 R, i1 C;  MLP_PACK: p0 W1, p3 W2, p1 img, i0 C.
 * No task may read or accumulate into what another task of the same call writes. */
#define VSX_WTASK_PREP 0
#define VSX_WTASK_TRANSPOSE 1
#define VSX_WTASK_MATVEC 2
#define VSX_WTASK_MLP_PACK 3
#define VSX_WTASK_UNPREP 4    /* vsx_unprep_grad: p0 g, p1 dparam, p3 gamma, p4 W, p2 dgamma, p5 u, p6 beta, p7 rowsub, i0 R, i1 Cs, i2 Tn, i3 tapmode */
#define VSX_WTASK_MATVEC_T 5  /* vsx_matvec_t_add: p0 W, p3 u, p1 out, i0 R, i1 C */
#define VSX_WTASK_REDUCE_ROWS 6  /* p1 out[n] += sum_r p0 ws[r][n], i0 rows, i1 columns */
#define VSX_WTASK_MAX 40
typedef struct VsxWTask {
  int32_t kind, dtype, i0, i1, i2, i3;
  const void* p0;
  void* p1;
  void* p2;
  const void* p3;
  const void* p4;
  const void* p5;
  const void* p6;
  const void* p7;
} VsxWTask;
int32_t vsx_weight_tasks(const VsxWTask* tasks, int32_t n, vsx_stream_t stream);

/* head Conv3d data-gradient weights: [Zout+2][C3][27*Cmid], zero where the depth tap falls outside [0,2]. */
int32_t vsx_prep_head_dgrad(const float* W, void* dst, int32_t Cmid, int32_t C3, int32_t Zout, int32_t dtype,
    vsx_stream_t stream);

/* K1 gather half of UNeXt2Stem (viscy_models/components/stems.py:26-50): the Conv3d with kernel = stride = (kz,ky,kx) is a GEMM over
 * non-overlapping patches; writes the patch matrix [B*h*w, D'*K].  Optional per-sample (sub, div) fuses NormalizeSampled
 * (viscy_transforms/_normalize.py:72-80): (x - sub)/(div + 1e-8). */
int32_t vsx_stem_im2col(const float* x, void* P, const float* sub, const float* div, int32_t B, int32_t Cin,
    int32_t Z, int32_t H, int32_t W, int32_t kz, int32_t ky, int32_t kx, int32_t dtype, vsx_stream_t stream);
/* the same gather with rows of ldp >= patch-size elements (tail zero-filled), and the matching weight-space helper
 * dst[r][k] = k < K ? src[r][k] : 0: K = 80 of the 5x4x4 stem becomes 96 = a whole number of 32-deep MFMA slabs */
int32_t vsx_stem_im2col_ld(const float* x, void* P, const float* sub, const float* div, int32_t B, int32_t Cin, int32_t Z,
    int32_t H, int32_t W, int32_t kz, int32_t ky, int32_t kx, int32_t ldp, int32_t dtype, vsx_stream_t stream);
int32_t vsx_pad_cols(const void* src, void* dst, int32_t R, int32_t K, int32_t Kp, int32_t dtype, vsx_stream_t stream);

/* Dense 3x3 convolution (padding 1) of a channels-last map as a GEMM — MONAI SubpixelUpsample's pre-convolution, selected by
 * UNeXt2(decoder_upsample_pre_conv=True) (/root/reference/packages/viscy-models/src/viscy_models/components/blocks.py:138-146).
 * vsx_im2col3x3: col[m][t*C + c] = x[b, y + t/3 - 1, x + t%3 - 1, c], zero outside the image (t = 3*ky + kx: the K order of
 * vsx_prep_weight(conv.weight, Cout, C, 9)).  vsx_col2im3x3: its transpose, dx[m][c] = sum_t dcol[m - shift(t)][t*C + c]. */
int32_t vsx_im2col3x3(const void* x, void* col, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, vsx_stream_t stream);
int32_t vsx_col2im3x3(const void* dcol, void* dx, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, vsx_stream_t stream);

/* K10: MONAI UpSample(mode="pixelshuffle", pre_conv=None) + torch.cat([up, skip], 1) (viscy_models/components/blocks.py:138-146,170-171). */
int32_t vsx_pixel_shuffle_cat_fwd(const void* low, const void* skip, void* out, int32_t B, int32_t h,
    int32_t w, int32_t c, int32_t cs, int32_t dtype, vsx_stream_t stream);

/* backward of K10: scatter dcat into dlow (inverse shuffle) and dskip. */
int32_t vsx_pixel_shuffle_cat_bwd(const void* dcat, void* dlow, void* dskip, int32_t B, int32_t h, int32_t w,
    int32_t c, int32_t cs, int32_t dtype, vsx_stream_t stream);

/* K12: PixelToVoxelHead.upsample (pixel shuffle x2 + optional ConstantPad2d((1,0,1,0)) + AvgPool2d(2,1)) + reshape (heads.py:607-615,632-637);
 * output channel = z*C3 + c3 so the 3x3x3 conv reads contiguous channel slices per depth. */
int32_t vsx_head_shuffle_fwd(const void* dec, void* hin, int32_t B, int32_t h, int32_t w, int32_t C3,
    int32_t D, int32_t pool, int32_t dtype, vsx_stream_t stream);

/* backward of K12. */
int32_t vsx_head_shuffle_bwd(const void* dhin, void* ddec, int32_t B, int32_t h, int32_t w, int32_t C3,
    int32_t D, int32_t pool, int32_t dtype, vsx_stream_t stream);

/* K17 NormalizeSampled.__call__ (viscy_transforms/_normalize.py:72-80) on a (B, ...) fp32 batch with (B,) statistics. */
int32_t vsx_normalize(const float* x, float* y, const float* sub, const float* div, int32_t B, int64_t per_sample,
    vsx_stream_t stream);

/* MinMaxSampled.__call__ (viscy_transforms/_normalize.py:124-134). */
int32_t vsx_minmax_norm(const float* x, float* y, const float* lo, const float* hi, int32_t B, int64_t per_sample,
    vsx_stream_t stream);

/* per-sample min / max needed by MONAI AdjustContrast (K19); mn / mx must be pre-set to +inf / -inf. */
int32_t vsx_sample_minmax(const float* x, float* mn, float* mx, int32_t B, int64_t per_sample, vsx_stream_t stream);

/* K19-K21 fused: BatchedRandAdjustContrast (_adjust_contrast.py:54-86) -> BatchedRandScaleIntensity
 * (_scale_intensity.py:59-77) -> BatchedRandGaussianNoise (_noise.py:158-204) with injected per-sample parameters. */
int32_t vsx_intensity_aug(const float* x, float* y, const float* mn, const float* mx, const float* gamma,
    const float* factor, const float* noise, const float* nstd, float nmean, int32_t invert, int32_t B, int64_t per_sample,
    vsx_stream_t stream);
/* invert: bit 0 = the gamma curve runs on -x (MONAI AdjustContrast(invert_image=True), called at _adjust_contrast.py:76-80),
 * bit 1 = the result is negated back; 0 or 3 in normal use, 1 when the retain_stats affine follows on the host side.
 * vsx_sample_moments: per-sample sum and sum of squares in double (sums[B][2], pre-set to 0) — the mean / std that
 * AdjustContrast(retain_stats=True) measures before and restores after the curve. */
int32_t vsx_sample_moments(const float* x, double* sums, int32_t B, int64_t per_sample, vsx_stream_t stream);

/* K25 _blend_in (viscy_utils/callbacks/prediction_writer.py:74-111): Z-feathered running average, fz[Z] = factors. */
int32_t vsx_blend_in(const float* oldp, const float* newp, float* out, const float* fz, int32_t Z, int64_t plane,
    int64_t total, vsx_stream_t stream);

/* K23 viscy_transforms.BatchedRandWeightedCropd (_crop.py:263-386): window weights = clamp(sum over (C, Z), 0) summed over
 * every (cy, cx) window (stride 1) -> wpool [B, (Y-cy+1)*(X-cx+1)]; tmp = caller scratch of B*Y*X + B*Y*(X-cx+1) floats. */
int32_t vsx_crop_weights(const float* w, float* wpool, float* tmp, int32_t B, int32_t CZ, int32_t Y, int32_t X,
    int32_t cy, int32_t cx, vsx_stream_t stream);
/* inverse-CDF draw: idx[b] = smallest i with sum_{j<=i} wpool[b,j] > u[b] * sum_j wpool[b,j]  (uniform when the sum is 0) */
int32_t vsx_sample_index(const float* wpool, const float* u, int32_t* idx, int32_t B, int64_t n, vsx_stream_t stream);
/* y[b,c,z,yy,xx] = x[b,c, starts[b][0]+z, starts[b][1]+yy, starts[b][2]+xx]   (starts int32 [B,3] on the device) */
int32_t vsx_crop3d(const float* x, float* y, const int32_t* starts, int32_t B, int32_t C, int32_t Z, int32_t Y, int32_t X,
    int32_t cz, int32_t cy, int32_t cx, vsx_stream_t stream);

/* K18 kornia warp_affine3d as used by BatchedRandAffined (viscy_transforms/_affine.py:33-47,358-393): trilinear (or
 * nearest) resampling; Minv[B][3][4] maps output-voxel to input-voxel coordinates (x, y, z order).
 * The kernel applies Minv as given; the caller folds kornia's align_corners convention into it (with align_corners=False — kornia's
 * RandomAffine3D default, which the reference forwards — output voxel i reads Da^-1 (R Da (i + 1/2) + t) - 1/2, a = (size - 1) / size
 * per axis: viscy_amd.transforms.kornia_sampling_matrix).
 * mode: bit 0 = nearest-neighbour; bits 1-2 = padding_mode of _affine.py:102-108 as torch's grid_sample treats it: 0 "zeros",
 * 1 "border" (coordinates clamped to the volume), 2 "reflection" with align_corners=True (mirrored about 0 and n-1),
 * 3 "reflection" with align_corners=False (mirrored about -1/2 and n-1/2, then clamped). */
int32_t vsx_warp_affine3d(const float* x, float* y, const float* Minv, int32_t B, int32_t C, int32_t D, int32_t H,
    int32_t W, int32_t mode, vsx_stream_t stream);
/* the same warp restricted to the output window [z0,z0+Do) x [y0,y0+Ho) x [x0,x0+Wo) of the (D,H,W) frame: fuses the
 * BatchedCenterSpatialCrop that follows the affine in the recipes (_crop.py:164-187); y: (B,C,Do,Ho,Wo). */
int32_t vsx_warp_affine3d_roi(const float* x, float* y, const float* Minv, int32_t B, int32_t C, int32_t D, int32_t H,
    int32_t W, int32_t z0, int32_t y0, int32_t x0, int32_t Do, int32_t Ho, int32_t Wo, int32_t mode, vsx_stream_t stream);

/* K22 one pass of the separable Gaussian of BatchedRandGaussianSmooth (viscy_transforms/_gaussian_smooth.py:141-167):
 * per-sample 1-D taps along the axis with element stride `stride` and length L, zero border. */
int32_t vsx_conv1d_axis(const float* x, float* y, const float* taps, int32_t k, int32_t B, int64_t per_sample,
    int64_t stride, int32_t L, vsx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VSX_H */
