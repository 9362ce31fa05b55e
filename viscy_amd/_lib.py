"""ctypes binding of libvsx.so (C-ABI declared in include/vsx.h).

There is NO fallback: if the shared library is missing or a tensor is not on a HIP device the
wrappers raise — a silent eager-PyTorch path would void every parity / performance claim.
"""

from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VSX_LIB", os.path.join(_HERE, "libvsx.so"))  # VSX_LIB: A/B-test another build

VSX_F32, VSX_BF16 = 0, 1
A_ROWS, A_PATCH2, A_CONV3 = 0, 1, 2
PRO_NONE, PRO_GRN = 0, 1
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU_SQ, EPI_BIAS_RES, EPI_DZ, EPI_BIAS_STATS, EPI_LN_BWD = 0, 1, 2, 3, 4, 5, 6
LOSS_SLOTS, LOSS_SLOT_STRIDE = 64, 32  # VSX_LOSS_SLOTS / VSX_LOSS_SLOT_STRIDE of include/vsx.h

_I32, _F32, _P, _I64 = C.c_int32, C.c_float, C.c_void_p, C.c_int64


class VsxGemm(C.Structure):
    _fields_ = [
        ("A", _P), ("B", _P), ("C", _P),
        ("M", _I32), ("N", _I32), ("K", _I32),
        ("lda", _I32), ("ldb", _I32), ("ldc", _I32),
        ("a_mode", _I32), ("gh", _I32), ("gw", _I32), ("cs", _I32),
        ("nz", _I32),
        ("a_coff", _I32 * 8), ("b_off", _I32 * 8), ("c_coff", _I32 * 8),
        ("c_mode", _I32), ("c_cs", _I32),
        ("pro", _I32),
        ("grn_s", _P), ("grn_b", _P),
        ("hw", _I32),
        ("epi", _I32),
        ("bias", _P), ("res", _P), ("ldr", _I32),
        ("aux", _P), ("ldx", _I32),
        ("red0", _P), ("red1", _P), ("colsum", _P), ("C2", _P),
        ("b_bstride", _I64),
        ("rscale", _P),
    ]


class VsxWTask(C.Structure):
    """one job of vsx_weight_tasks (include/vsx.h)"""
    _fields_ = [("kind", _I32), ("dtype", _I32), ("i0", _I32), ("i1", _I32), ("i2", _I32), ("i3", _I32),
                ("p0", _P), ("p1", _P), ("p2", _P), ("p3", _P), ("p4", _P), ("p5", _P), ("p6", _P), ("p7", _P)]


WTASK_PREP, WTASK_TRANSPOSE, WTASK_MATVEC, WTASK_MLP_PACK, WTASK_UNPREP, WTASK_MATVEC_T, WTASK_REDUCE_ROWS = 0, 1, 2, 3, 4, 5, 6

_SIGS = {
    "vsx_version": (_I32, []),
    "vsx_weight_tasks": (_I32, [_P, _I32, _P]),
    "vsx_last_error": (C.c_char_p, []),
    "vsx_last_kernel": (C.c_char_p, []),
    "vsx_set_flag": (_I32, [C.c_char_p, _I32]),
    "vsx_get_flag": (_I32, [C.c_char_p]),
    "vsx_det_workspace": (_I32, [_P, _I64]),
    "vsx_gemm_nt": (_I32, [C.POINTER(VsxGemm), _I32, _P]),
    "vsx_gemm_tn": (_I32, [C.POINTER(VsxGemm), _I32, _P]),
    "vsx_gemm_nt_ln_bwd_supported": (_I32, [_I64, _I32, _I32, _I32]),
    "vsx_ln_fwd": (_I32, [_P, _P, _P, _P, _P, _P, _I32, _I32, _F32, _I32, _P]),
    "vsx_ln_bwd": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P]),
    "vsx_grn_scale": (_I32, [_P, _P, _P, _I32, _I32, _F32, _P]),
    "vsx_grn_bwd_stats": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _F32, _P]),
    "vsx_grn_gelu_bwd": (_I32, [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "vsx_dwconv7_fwd": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "vsx_dwconv7_bwd_data": (_I32, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "vsx_dwconv7_bwd_weight": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "vsx_stem_im2col": (_I32, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "vsx_stem_im2col_ld": (_I32, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "vsx_pad_cols": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _P]),
    "vsx_im2col3x3": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "vsx_col2im3x3": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "vsx_pixel_shuffle_cat_fwd": (_I32, [_P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "vsx_pixel_shuffle_cat_bwd": (_I32, [_P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "vsx_head_shuffle_fwd": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "vsx_head_shuffle_bwd": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "vsx_head_out_fwd": (_I32, [_P] * 7 + [_I32] * 6 + [_F32, _I32, _P]),
    "vsx_head_out_bwd1": (_I32, [_P] * 11 + [_I32] * 6 + [_F32, _I32, _P]),
    "vsx_head_out_bwd1_wgrad": (_I32, [_P] * 13 + [_I32] * 6 + [_F32, _I32, _P]),
    "vsx_head_out_bwd2": (_I32, [_P] * 9 + [_I32] * 6 + [_F32, _I32, _P]),
    "vsx_loss_pool": (_I32, [_P] * 7 + [_I32] * 3 + [_P]),
    "vsx_ssim_scale_fwd": (_I32, [_P] * 5 + [_I32] * 5 + [_P]),
    "vsx_ssim_scale_bwd": (_I32, [_P] * 7 + [_I32] * 5 + [_F32, _F32, _P, _I32, _P]),
    "vsx_loss_finalize": (_I32, [_P] * 5 + [_F32, _I32, _I32, _I32, _F32, _F32, _F32, _P, _P, _P, _P, _P]),
    "vsx_ssim_scale_fwd_dmu": (_I32, [_P] * 6 + [_I32] * 6 + [_P]),
    "vsx_ssim_scale_bwd_in": (_I32, [_P] * 6 + [_I32] * 5 + [_F32, _F32, _P, _I32, _I32, _P]),
    "vsx_loss_tmax": (_I32, [_P, _I64, _P, _P]),
    "vsx_ssim_scale_fwd_fused": (_I32, [_P] * 11 + [_I32] * 6 + [_P]),
    "vsx_adamw": (_I32, [_P, _P, _P, _P, _P, _I64, _P]),
    "vsx_adamw_advance": (_I32, [_P, _P, _P, _P]),
    "vsx_fill_f32": (_I32, [_P, _I64, _F32, _P]),
    "vsx_mlp_supported": (_I32, [_I32, _I32, _I64, _I32]),
    "vsx_mlp_mode_supported": (_I32, [_I32, _I32, _I64, _I32, _I32]),
    "vsx_mlp_image_bytes": (_I64, [_I32]),
    "vsx_mlp_pack": (_I32, [_P, _P, _P, _I32, _P]),
    "vsx_mlp_fwd": (_I32, [_P] * 11 + [_I64, _I32, _I32, _I32, _I32, _P]),
    "vsx_mlp_fc1": (_I32, [_P] * 7 + [_I64, _I32, _I32, _I32, _P]),
    "vsx_mlp_fwd_ln": (_I32, [_P, _F32] + [_P] * 10 + [_I64, _I32, _I32, _I32, _I32, _P]),
    "vsx_mlp_fc1_ln": (_I32, [_P, _F32] + [_P] * 9 + [_I64, _I32, _I32, _I32, _P]),
    "vsx_grn_q_reduce": (_I32, [_P] * 10 + [_I64, _I32, _I32, _I32, _P]),
    "vsx_grn_q_reduce_ws_floats": (_I64, [_I32, _I32]),
    "vsx_mlp_bwd_stats": (_I32, [_P] * 5 + [_I64, _I32, _I32, _I32, _P]),
    "vsx_mlp_bwd_dh": (_I32, [_P] * 7 + [_I64, _P, _P, _I64, _I32, _I32, _I32, _P]),
    "vsx_mlp_rows_per_workgroup": (_I32, [_I32, _I32, _I64]),
    "vsx_mlp_bwd_dh_re": (_I32, [_P] * 9 + [_I64, _P, _P, _I64, _I32, _I32, _I32, _P]),
    "vsx_mlp_bwd_dh_ln": (_I32, [_P] * 11 + [_I64, _P, _P, _I64, _I32, _I32, _I32, _P]),
    "vsx_mlp_gelu_table_len": (_I32, []),
    "vsx_mlp_gelu_table": (_I32, [_P, _P]),
    "vsx_prep_weight": (_I32, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "vsx_unprep_grad": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "vsx_matvec": (_I32, [_P, _P, _P, _P, _I32, _I32, _P]),
    "vsx_matvec_t_add": (_I32, [_P, _P, _P, _I32, _I32, _P]),
    "vsx_transpose_f32": (_I32, [_P, _P, _I32, _I32, _I32, _P]),
    "vsx_prep_head_dgrad": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _P]),
    "vsx_normalize": (_I32, [_P, _P, _P, _P, _I32, _I64, _P]),
    "vsx_minmax_norm": (_I32, [_P, _P, _P, _P, _I32, _I64, _P]),
    "vsx_sample_minmax": (_I32, [_P, _P, _P, _I32, _I64, _P]),
    "vsx_intensity_aug": (_I32, [_P] * 8 + [_F32, _I32, _I32, _I64, _P]),
    "vsx_sample_moments": (_I32, [_P, _P, _I32, _I64, _P]),
    "vsx_blend_in": (_I32, [_P, _P, _P, _P, _I32, _I64, _I64, _P]),
    "vsx_scale_weight_samples": (_I32, [_P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "vsx_voxel_shuffle_fwd": (_I32, [_P, _P] + [_I32] * 8 + [_P]),
    "vsx_voxel_shuffle_bwd": (_I32, [_P, _P] + [_I32] * 8 + [_P]),
    "vsx_layer_scale_fold": (_I32, [_P] * 5 + [_I32, _I32, _P]),
    "vsx_layer_scale_unfold": (_I32, [_P] * 8 + [_I32, _I32, _P]),
    "vsx_avgpool_rows_fwd": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _P]),
    "vsx_avgpool_rows_bwd": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _P]),
    "vsx_bn1d_fwd": (_I32, [_P] * 8 + [_I32, _I32, _F32, _F32, _I32, _I32, _P]),
    "vsx_bn1d_bwd": (_I32, [_P] * 9 + [_I32] * 4 + [_P]),
    "vsx_ntxent_fwd": (_I32, [_P] * 8 + [_I32, _I32, _F32, _F32, _P]),
    "vsx_ntxent_bwd": (_I32, [_P] * 6 + [_I32, _I32, _P]),
    "vsx_scale_rows_samples": (_I32, [_P, _P, _P, _I64, _I32, _I32, _I32, _P]),
    "vsx_rows_select": (_I32, [_P, _P, _P, _P, _I64, _I32, _I32, _P]),
    "vsx_masked_mse_fwd": (_I32, [_P] * 5 + [_I32] * 3 + [_I64, _P]),
    "vsx_masked_mse_bwd": (_I32, [_P] * 6 + [_I32] * 3 + [_I64, _P]),
    "vsx_head_conv_supported": (_I32, [_I32] * 6),
    "vsx_head_conv_fwd": (_I32, [_P] * 6 + [_I32] * 7 + [_P]),
    "vsx_head_conv_wgrad": (_I32, [_P] * 4 + [_I32] * 7 + [_P]),
    "vsx_head_conv_dgrad_prep": (_I32, [_P, _P, _I32, _P]),
    "vsx_head_conv_dgrad": (_I32, [_P] * 3 + [_I32] * 7 + [_P]),
    "vsx_crop_weights": (_I32, [_P, _P, _P] + [_I32] * 6 + [_P]),
    "vsx_sample_index": (_I32, [_P, _P, _P, _I32, _I64, _P]),
    "vsx_crop3d": (_I32, [_P, _P, _P] + [_I32] * 8 + [_P]),
    "vsx_warp_affine3d": (_I32, [_P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "vsx_warp_affine3d_roi": (_I32, [_P, _P, _P] + [_I32] * 12 + [_P]),
    "vsx_conv1d_axis": (_I32, [_P, _P, _P, _I32, _I32, _I64, _I64, _I32, _P]),
}

_lib = None


def exported_symbols() -> list[str]:
    return sorted(_SIGS)


def lib() -> C.CDLL:
    """Load libvsx.so (once). Raises if it has not been built: `python -m viscy_amd.build`."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"viscy_amd: {LIB_PATH} not found — the HIP kernels are mandatory (no CPU / eager fallback). "
                "Build them with `python -m viscy_amd.build` (hipcc, --offload-arch=gfx950)."
            )
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)  # AttributeError here = header / library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = l
        # VSX_FLAGS="nt2=0,tn_rect=0": tuning knobs of vsx_set_flag for A/B runs of unmodified programs (bench.py, tests)
        for kv in filter(None, os.environ.get("VSX_FLAGS", "").split(",")):
            name, _, val = kv.partition("=")
            if l.vsx_set_flag(name.strip().encode(), int(val)) != 0:
                raise RuntimeError(f"VSX_FLAGS: {l.vsx_last_error().decode(errors='replace')}")
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().vsx_last_error().decode(errors="replace")
        raise RuntimeError(f"libvsx {what} failed (rc={rc}): {msg}")


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return VSX_F32
    if dt == torch.bfloat16:
        return VSX_BF16
    raise TypeError(f"viscy_amd kernels compute in float32 or bfloat16, got {dt}")


def ptr(t: torch.Tensor | None) -> int | None:
    """Device pointer of a contiguous HIP tensor (None → NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("viscy_amd: tensor is not on a HIP device — the MI355X kernels have no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError("viscy_amd: tensor must be contiguous")
    return t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream
