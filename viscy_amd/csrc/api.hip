// libvsx.so — error plumbing, version and debug knobs of the C-ABI (include/vsx.h).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "vsx_common.h"
#include "../../include/vsx.h"

static thread_local char g_err[512] = "";
thread_local const char* g_vsx_last_kernel = "";  // kernel template the last GEMM entry point on this thread dispatched (vsx_last_kernel)
int g_vsx_tn_tr = 1;
int g_vsx_nt_wide = 1;
int g_vsx_nt_fast = 3;  // bit 0: the lean NT / TN instantiations; bit 1 (round 6): K tails and the 2 x 2 patch gather on the lean NT kernel
int g_vsx_tn_wide = 1;
int g_vsx_ggb_blocks = 2048;  // grn_gelu_bwd: target workgroup count (tuning knob, see norm.hip)
int g_vsx_tn_rect = 11;  // rectangular TN tiles: bit 0 = when N or K is 224..256 wide, bit 1 = 256x128 when 256 divides N (no prologue), bit 2 = 128x256 when 256 divides K (slower: off), bit 3 (round 5) = where the block backward takes its GRN statistics by recomputation (C = 384), the fc2 weight gradient delivers them instead: per-sample products scaled / contracted in the accumulators (gemm_tn_fast_kernel PRO == 2; read by viscy_amd.ops.tn_grn_stats_ok); bits 4 / 5 / 6 (round 6, off): eight-wave workgroups on 256 x 256 tiles for the plain weight gradients with K in [192, 256] (isolated -8 .. -15 %, step +-0), 256 x 192 tiles for K = 384 / 768 (+7 %), 256 x 256 tiles for the per-sample products (+4 .. +10 %)
int g_vsx_dw_mfma = 15;  // depthwise conv on the matrix cores (dwconv_mfma.hip): bit 0 forward / data gradient (banded Toeplitz tiles), bit 3 the same with its tiles fetched by LDS-DMA two tiles ahead (round 4; whole 32-channel slabs only), bit 1 weight gradient (row contraction, transpose reads), bit 2 16-column weight-gradient tiles at every width (the 32-column variant spills: 471 vs 285 us at 64x64x96, B = 512); 0: VALU stencils
int g_vsx_ln_fblk = 32768;  // LayerNorm forward: cap on workgroups per launch (each sweeps rows / cap windows).  Measured at B = 512 (64x64x96 / x224): 2048 -> 209 / 438 us, 8192 -> 172 / 351, 32768 -> 162 / 331 (a grid-stride sweep by few workgroups streams at 5.0 TB/s where one vector per thread reaches 6.8: tools/micro/write_rate.hip)
int g_vsx_ln_bblk = 8192;   // LayerNorm backward WITHOUT affine gradients (the block LayerNorms): cap on workgroups (with dgamma: 512, same-address atomics).  512 -> 319 / 651 us, 2048 -> 254 / 586, 8192 -> 240 / 541 (16x16x384: 83 -> 61), 32768 -> 225 / 516 but 78 at 16x16x384
int g_vsx_ln_ablk = 512;  // LayerNorm backward WITH affine gradients: workgroup count = same-address atomics per dgamma / dbeta element; 0 = sized from the bytes of the pass (norm.hip ln_launch: isolated launches 336 -> 262 us at C = 96, 171 -> 132 at C = 192; bench step +-0: 91.47 vs 91.38 ms, profiles/r06_ln_affine_cap.txt), so the fixed 512 of rounds 1 - 5 stays
int g_vsx_ln_pack = 1;  // LayerNorm backward: rows of 24 / 48 16-byte vectors (C = 192 / 384 in bf16) on 8- / 16-lane groups with three vectors per lane instead of 32 / 64 lanes a quarter idle (norm.hip ln_dispatch, round 6)
int g_vsx_nt_stream = 3;  // (round 4: a non-temporal LDS-DMA of the A panel in the second-generation NT kernel measured 14.9 -> 17.1 ms for the class: not kept) lean NT kernel: bit 0 = non-temporal stores of the wide outputs (fc1 h / g, fc2 data gradient dz): +0.6..1.9 % on the step; bit 1 = non-temporal load of the stored activation in the dZ epilogue (its last reader): +0.7 % (same-box A/B).  ON since round 3: the streaming stores are compiler builtins now (round 1 used inline asm, see vsx_common.h stvec_stream), soak / determinism / poison tests run with them
int g_vsx_grn_stream = 2;  // grn_gelu_bwd: bit 0 = non-temporal store of dz (no effect), bit 1 = non-temporal load of h, its last reader (-3 % on the kernel)
int g_vsx_ggb_contig = 1;  // grn_gelu_bwd: contiguous row range per workgroup instead of grid-strided rows
int g_vsx_tn_fill = 1;  // TN split counts chosen to fill their last round of workgroups (csrc/gemm.hip fill_splits)
int g_vsx_tn_want3 = 256;  // TN split target of the eight-wave 256 x 256 / 256 x 384 tiles (tn_rect bit 4): workgroups per launch, one per CU
int g_vsx_tn_want2 = 512;  // TN split target of the rectangular (256 x 128 / 128 x 256) tiles: workgroups per launch
int g_vsx_tn_p2_rounds = 1;  // weight gradient with GRN statistics (gemm_tn_fast_kernel PRO == 2): rounds of 512 workgroups the split count aims at (0 = power-of-two splits, round-5 first version)
int g_vsx_tn_want = 768;  // TN split target: workgroups per launch (tiles x splits)
int g_vsx_ln_stream = 3;  // non-temporal loads of operands with no later reader: bit 0 = ln_bwd (dy, x: -2 % on the kernel), bit 1 = ln_fwd (x: -4.5 %)
int g_vsx_tn_stream = 3;  // lean TN kernel: non-temporal loads of an operand that the launch reads exactly once (its dimension fits one tile): bit 0 = X [M, N], bit 1 = Y [M, K] (round 4)
int g_vsx_tn_contig = 1;  // lean TN kernel: contiguous step range per split
int g_vsx_mlp_fused = 111;  // fused GRN-MLP kernels (csrc/mlp.hip): bit 0 = inference forward (statistics + output passes, hidden activation on chip), bit 1 = training fc1 (statistics pass that also stores h and g), bit 3 = block backward without a stored dz (statistics from the per-sample weight-gradient products or MODE 3, then MODE 4 writes dh once) — on the C = 96 / 192 / 224 blocks; bit 2 = the training passes also on the C = 384 blocks (same step time, 7.7 GB less traffic per step); (bit 4 was the inference pair on the C = 384 blocks, slower than the unfused GEMMs there: removed in round 4); bit 5 = the block LayerNorm in the prologue of the fused passes (vsx_mlp_fwd_ln / vsx_mlp_fc1_ln: no separate LayerNorm pass); bit 6 = the pre-activation h is never stored on the C <= 224 blocks: the training fc1 writes g only (MODE 6) and the dh pass recomputes h from the C-wide normalised rows (MODE 5, vsx_mlp_bwd_dh_re); bit 7 (with 5 and 6) = the normalised rows x^ are not stored either: the forward keeps the depthwise output y + the row mean / rstd, the dh pass re-normalises y and writes dh * rstd (MODE 7, vsx_mlp_bwd_dh_ln), the fc1 weight gradient is a plain TN GEMM on y with a rank-1 correction, the LayerNorm backward in the data-gradient GEMM re-forms x^ from y.  Round 6 ships 111 (bit 7 off: x^ stored again, MODE 5): 92.28 -> 91.72 ms on the bench step, 90.42 -> 89.95 ms on the gate shape, four alternating pairs each on one box (and -0.25 / -0.7 ms on two other boxes), for 5.2 GB / step more C-wide writes: at round 6's kernels the re-normalisation in the dh pass and the rank-1 correction cost more than the bytes they save
int g_vsx_mlp_sf32 = 1 | 4 | 64;  // fused GRN-MLP kernels: bit m = MODE m runs the build without packed-fp32 VALU instructions (csrc/mlp.hip, round 5: a v_pk_*_f32 next to MFMAs costs ~15 cycles; the forward passes gain 6 - 20 %, the dh passes are VALU-bound and keep the packed build)
int g_vsx_det_reduce = 0;  // 1: the forward's per-sample sums — GRN sum g^2 of the fused GRN-MLP passes and of gemm_nt2's GELU epilogue, InstanceNorm sum / sum^2 of the direct head convolution — are formed in a FIXED order (per-workgroup partials in a caller-owned workspace, vsx_det_workspace, then one ordered pass) instead of by fp32 atomics: the bf16 forward is then bit-identical from run to run (with atomics: 7e-3 of the output maximum at 2048^2).  Cost: one small launch per pass, tools/det_fwd.py
thread_local float* g_vsx_det_ws = nullptr;
thread_local long g_vsx_det_ws_floats = 0;
int g_vsx_nt2 = 17;  // second-generation NT kernel (gemm_nt2.hip: 256 x 128 tiles, LDS-DMA operand path, wave-private epilogue): bit 0 = on for the launches it supports, bit 1 = also below 512 tiles
int g_vsx_head_rows = 63;  // PixelToVoxelHead tail on row tiles with the 1x1x1 contraction on the matrix cores (head.hip, round 6; bf16, 64 | W2, Z <= 8): bit 0 = forward, bit 1 = backward pass 2, bit 2 = backward pass 1 with the folded weight gradient; bits 3 / 4 = the pixel shuffle + pad-pool in front of the head convolution and its adjoint on column strips (spatial.hip head_shuffle_{fwd,bwd}_strip_kernel; bf16, pooled, C3 * D = 56, 64 | w); bit 5 = the direct head convolution forward as a persistent kernel that requests the next halo tile ahead (headconv.hip)
int g_vsx_head_bps = 0;  // head backward pass 1: workgroups per sample (0 = max(32, 8192 / B)); a small value makes every workgroup walk several tiles (tests)
int g_vsx_loss_fused = 1;  // MixedLoss training forward: one pass per scale (SSIM sums + gradient field + next scale's pooling / data range + L1 / L2 sums: vsx_ssim_scale_fwd_fused) instead of a pooling pass and an SSIM pass; read by viscy_amd/losses.py

void vsx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

/* det_reduce: scratch for the per-workgroup partial sums of the NEXT launches on this thread (caller-owned, `floats` fp32 values;
 * each launch that needs it checks the size and says how much it wants).  NULL / 0 takes it away again. */
extern "C" int32_t vsx_det_workspace(float* ws, int64_t floats) {
  g_vsx_det_ws = ws;
  g_vsx_det_ws_floats = ws ? (long)floats : 0;
  return 0;
}
extern "C" int32_t vsx_version(void) { return 1; }
extern "C" const char* vsx_last_error(void) { return g_err; }
extern "C" const char* vsx_last_kernel(void) { return g_vsx_last_kernel; }
extern "C" int32_t vsx_set_flag(const char* name, int32_t value) {
  if (name && !strcmp(name, "tn_tr")) { g_vsx_tn_tr = value; return 0; }
  if (name && !strcmp(name, "nt_wide")) { g_vsx_nt_wide = value; return 0; }
  if (name && !strcmp(name, "nt_fast")) { g_vsx_nt_fast = value; return 0; }
  if (name && !strcmp(name, "tn_wide")) { g_vsx_tn_wide = value; return 0; }
  if (name && !strcmp(name, "nt2")) { g_vsx_nt2 = value; return 0; }
  if (name && !strcmp(name, "nt_stream")) { g_vsx_nt_stream = value; return 0; }
  if (name && !strcmp(name, "grn_stream")) { g_vsx_grn_stream = value; return 0; }
  if (name && !strcmp(name, "ggb_contig")) { g_vsx_ggb_contig = value; return 0; }
  if (name && !strcmp(name, "tn_want")) { g_vsx_tn_want = value; return 0; }
  if (name && !strcmp(name, "tn_p2_rounds")) { g_vsx_tn_p2_rounds = value; return 0; }
  if (name && !strcmp(name, "tn_want2")) { g_vsx_tn_want2 = value; return 0; }
  if (name && !strcmp(name, "tn_want3")) { g_vsx_tn_want3 = value; return 0; }
  if (name && !strcmp(name, "tn_fill")) { g_vsx_tn_fill = value; return 0; }
  if (name && !strcmp(name, "tn_contig")) { g_vsx_tn_contig = value; return 0; }
  if (name && !strcmp(name, "tn_stream")) { g_vsx_tn_stream = value; return 0; }
  if (name && !strcmp(name, "ln_stream")) { g_vsx_ln_stream = value; return 0; }
  if (name && !strcmp(name, "ggb_blocks")) { g_vsx_ggb_blocks = value; return 0; }
  if (name && !strcmp(name, "tn_rect")) { g_vsx_tn_rect = value; return 0; }
  if (name && !strcmp(name, "dw_mfma")) { g_vsx_dw_mfma = value; return 0; }
  if (name && !strcmp(name, "ln_fblk") && value > 0) { g_vsx_ln_fblk = value; return 0; }
  if (name && !strcmp(name, "ln_bblk") && value > 0) { g_vsx_ln_bblk = value; return 0; }
  if (name && !strcmp(name, "ln_ablk") && value >= 0) { g_vsx_ln_ablk = value; return 0; }
  if (name && !strcmp(name, "ln_pack")) { g_vsx_ln_pack = value; return 0; }
  if (name && !strcmp(name, "mlp_fused")) { g_vsx_mlp_fused = value; return 0; }
  if (name && !strcmp(name, "loss_fused")) { g_vsx_loss_fused = value; return 0; }
  if (name && !strcmp(name, "mlp_sf32")) { g_vsx_mlp_sf32 = value; return 0; }
  if (name && !strcmp(name, "det_reduce")) { g_vsx_det_reduce = value; return 0; }
  if (name && !strcmp(name, "head_rows")) { g_vsx_head_rows = value; return 0; }
  if (name && !strcmp(name, "head_bps") && value >= 0) { g_vsx_head_bps = value; return 0; }
  vsx_set_error("vsx_set_flag: unknown flag '%s'", name ? name : "(null)");
  return 1;
}
extern "C" int32_t vsx_get_flag(const char* name) {
  if (name && !strcmp(name, "tn_tr")) return g_vsx_tn_tr;
  if (name && !strcmp(name, "nt_wide")) return g_vsx_nt_wide;
  if (name && !strcmp(name, "nt_fast")) return g_vsx_nt_fast;
  if (name && !strcmp(name, "tn_wide")) return g_vsx_tn_wide;
  if (name && !strcmp(name, "nt2")) return g_vsx_nt2;
  if (name && !strcmp(name, "nt_stream")) return g_vsx_nt_stream;
  if (name && !strcmp(name, "grn_stream")) return g_vsx_grn_stream;
  if (name && !strcmp(name, "ggb_contig")) return g_vsx_ggb_contig;
  if (name && !strcmp(name, "tn_want")) return g_vsx_tn_want;
  if (name && !strcmp(name, "tn_p2_rounds")) return g_vsx_tn_p2_rounds;
  if (name && !strcmp(name, "tn_want2")) return g_vsx_tn_want2;
  if (name && !strcmp(name, "tn_want3")) return g_vsx_tn_want3;
  if (name && !strcmp(name, "tn_fill")) return g_vsx_tn_fill;
  if (name && !strcmp(name, "tn_contig")) return g_vsx_tn_contig;
  if (name && !strcmp(name, "tn_stream")) return g_vsx_tn_stream;
  if (name && !strcmp(name, "ln_stream")) return g_vsx_ln_stream;
  if (name && !strcmp(name, "ggb_blocks")) return g_vsx_ggb_blocks;
  if (name && !strcmp(name, "tn_rect")) return g_vsx_tn_rect;
  if (name && !strcmp(name, "dw_mfma")) return g_vsx_dw_mfma;
  if (name && !strcmp(name, "ln_fblk")) return g_vsx_ln_fblk;
  if (name && !strcmp(name, "ln_bblk")) return g_vsx_ln_bblk;
  if (name && !strcmp(name, "ln_ablk")) return g_vsx_ln_ablk;
  if (name && !strcmp(name, "ln_pack")) return g_vsx_ln_pack;
  if (name && !strcmp(name, "mlp_fused")) return g_vsx_mlp_fused;
  if (name && !strcmp(name, "loss_fused")) return g_vsx_loss_fused;
  if (name && !strcmp(name, "mlp_sf32")) return g_vsx_mlp_sf32;
  if (name && !strcmp(name, "det_reduce")) return g_vsx_det_reduce;
  if (name && !strcmp(name, "head_rows")) return g_vsx_head_rows;
  if (name && !strcmp(name, "head_bps")) return g_vsx_head_bps;
  return -1;
}
