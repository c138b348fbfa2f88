"""ctypes binding of libarcflow_hip.so (include/arcflow_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call fails this
module raises.  Build it with ``python -m arcflow_amd.build`` (or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from .build import lib_path

AFX_DT_BF16 = 1
AFX_DT_F32 = 2
AFX_DT_FP8 = 3

EXPORTS = [
    'afx_last_error', 'afx_version', 'afx_create', 'afx_destroy', 'afx_bind_weight', 'afx_finalize',
    'afx_workspace_bytes', 'afx_set_workspace', 'afx_mmdit_forward', 'afx_profile_enable', 'afx_profile_read', 'afx_set_checkpoint_buffer', 'afx_arcflow_step',
    'afx_arcflow_velocity', 'afx_linear_bf16', 'afx_attention_ws_bytes', 'afx_attention_bf16', 'afx_attention_to_mx8',
    'afx_norm_modulate_bf16', 'afx_qk_norm_rope_bf16', 'afx_gemv_bf16',
    'afx_attention_fwd_lse_bf16', 'afx_attention_bwd_ws_bytes', 'afx_attention_bwd_bf16',
    'afx_ln_modulate_backward', 'afx_qk_norm_rope_oop_bf16', 'afx_gelu_bf16', 'afx_add_scale_bf16',
    'afx_conv3x3_bf16', 'afx_conv3x3_bf16_stats', 'afx_groupnorm_nhwc_from_stats', 'afx_conv_stats_available', 'afx_upconv3x3_bf16', 'afx_groupnorm_nhwc', 'afx_groupnorm_ws_bytes', 'afx_upsample2x_nhwc', 'afx_interior_nhwc', 'afx_softmax_rows_f32',
    'afx_latent_to_nhwc', 'afx_nhwc_to_image', 'afx_latent_to_nhwc_affine', 'afx_rmsnorm_nhwc',
    'afx_embed_rows_bf16', 'afx_norm_rows_bf16', 'afx_act_mul_bf16', 'afx_rope_half_bf16', 'afx_attention_ext_ws_bytes',
    'afx_attention_ext_bf16', 'afx_linear_bf16_splitk', 'afx_finish_f32_bf16', 'afx_linear_splitk_chunks', 'afx_quant_rows_fp8', 'afx_linear_fp8', 'afx_quant_rows_mx8', 'afx_linear_fp8_mx', 'afx_linear_fp8_to_mx8',
    'afx_linear_bf16_pre', 'afx_linear_bf16_sk', 'afx_linear_sk_ws_bytes', 'afx_linear_sk_last_split', 'afx_gemm_set_mode', 'afx_gemm_dropres_available', 'afx_attn_set_impl', 'afx_attn_bwd_set_impl', 'afx_mmdit_prepare_steps', 'afx_mmdit_use_prepared_step', 'afx_lora_dropout_bf16', 'afx_mmdit_forward_stage', 'afx_mmdit_import_tokens',
    'afx_coldot_bf16', 'afx_gate_residual_bf16', 'afx_gemv_t_bf16', 'afx_set_temb_override', 'afx_set_fp8_linear',
    'afx_arcflow_step_dropout', 'afx_arcflow_backward', 'afx_mse_loss', 'afx_euler_roll', 'afx_axpby_rows', 'afx_cfg_combine',
    'afx_head_grad', 'afx_linear_bf16_f32out', 'afx_linear_tn_f32out', 'afx_linear_tn_f32out_ws', 'afx_linear_tn_ws_bytes', 'afx_linear_bf16_dropres', 'afx_transpose_bf16', 'afx_colsum_bf16', 'afx_normout_backward', 'afx_normout_backward_split',
    'afx_outer_accum', 'afx_mmdit_export', 'afx_sumsq', 'afx_adamw_step', 'afx_adamw8bit_step', 'afx_ema_lerp', 'afx_cast_f32_bf16',
]


class ModelDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'family', 'num_double', 'num_single', 'heads', 'head_dim', 'in_channels', 'joint_dim',
        'pooled_dim', 'guidance_embeds', 'num_gaussians', 'logweights_channels', 'head_mode')]


class ArcflowHipError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen the engine; raises ArcflowHipError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get('ARCFLOW_HIP_LIB', lib_path())
    if not os.path.exists(path):
        raise ArcflowHipError(
            f'{path} not found: the HIP engine is not built. Run `python -m arcflow_amd.build` '
            '(needs hipcc, cross-compiles for gfx950 without a GPU). There is no CPU fallback.')
    lib = C.CDLL(path)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    lib.afx_last_error.restype = C.c_char_p
    lib.afx_version.restype = C.c_char_p
    lib.afx_create.argtypes = [C.POINTER(ModelDesc), C.POINTER(vp)]
    lib.afx_destroy.argtypes = [vp]
    lib.afx_bind_weight.argtypes = [vp, C.c_char_p, vp, i32, i32, C.POINTER(i64)]
    lib.afx_finalize.argtypes = [vp]
    lib.afx_workspace_bytes.argtypes = [vp, i32, i32, i32]
    lib.afx_workspace_bytes.restype = i64
    lib.afx_set_workspace.argtypes = [vp, vp, i64]
    lib.afx_mmdit_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]
    lib.afx_set_checkpoint_buffer.argtypes = [vp, vp]
    lib.afx_profile_enable.argtypes = [vp, i32]
    lib.afx_profile_read.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double)]
    lib.afx_arcflow_step.argtypes = [vp, vp, vp, vp, i32, f32, f32, f32, vp, f32, vp, i32, i32, i32, i32, i32, vp]
    lib.afx_arcflow_velocity.argtypes = [vp, vp, vp, i32, f32, f32, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.afx_linear_bf16.argtypes = [vp, i64, vp, i64, vp, vp, i64, i32, i32, i32, i32, i32, vp, i64, i32, vp, i64, vp]
    lib.afx_attention_ws_bytes.argtypes = [i32, i32, i32]
    lib.afx_attention_ws_bytes.restype = i64
    lib.afx_attention_bf16.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, vp, i32, i32, i32, vp]
    lib.afx_attention_to_mx8.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i32, i32, i32, vp]
    lib.afx_norm_modulate_bf16.argtypes = [vp, i64, vp, i64, i32, i32, vp, vp, i64, i32, i32, vp]
    lib.afx_qk_norm_rope_bf16.argtypes = [vp, i64, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.afx_gemv_bf16.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.afx_attention_fwd_lse_bf16.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, i32, i32, i32, vp]
    lib.afx_attention_bwd_ws_bytes.argtypes = [i32, i32, i32]
    lib.afx_attention_bwd_ws_bytes.restype = i64
    lib.afx_attention_bwd_bf16.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, i64, vp, i64, vp, i64, vp, i32, i32, i32, vp]
    lib.afx_ln_modulate_backward.argtypes = [vp, i64, vp, i64, vp, i64, i32, vp, i64, vp, i64, i32, i32, vp]
    lib.afx_qk_norm_rope_oop_bf16.argtypes = [vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.afx_gelu_bf16.argtypes = [vp, i64, vp, i64, vp, i64, i64, i32, vp]
    lib.afx_add_scale_bf16.argtypes = [vp, i64, vp, i64, vp, i64, i32, vp, i64, i64, i32, vp]
    lib.afx_conv3x3_bf16.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]
    lib.afx_groupnorm_nhwc.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, f32, i32, vp]
    lib.afx_groupnorm_ws_bytes.argtypes = [i32, i32]
    lib.afx_groupnorm_ws_bytes.restype = i64
    lib.afx_conv3x3_bf16_stats.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, i32, vp]
    lib.afx_groupnorm_nhwc_from_stats.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, f32, i32, vp]
    lib.afx_conv_stats_available.argtypes = []
    lib.afx_upconv3x3_bf16.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.afx_upsample2x_nhwc.argtypes = [vp, vp, i32, i32, i32, vp]
    lib.afx_interior_nhwc.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
    lib.afx_softmax_rows_f32.argtypes = [vp, i64, vp, i64, i32, i32, f32, vp]
    lib.afx_latent_to_nhwc.argtypes = [vp, vp, i32, i32, i32, f32, f32, vp]
    lib.afx_nhwc_to_image.argtypes = [vp, vp, i32, i32, i32, vp]
    lib.afx_latent_to_nhwc_affine.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
    lib.afx_rmsnorm_nhwc.argtypes = [vp, vp, i64, i32, i32, vp, i32, vp]
    lib.afx_embed_rows_bf16.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    lib.afx_norm_rows_bf16.argtypes = [vp, i64, vp, i64, i32, i32, vp, vp, f32, i32, vp]
    lib.afx_act_mul_bf16.argtypes = [vp, i64, vp, i64, i64, i32, i32, i32, vp]
    lib.afx_rope_half_bf16.argtypes = [vp, i64, vp, vp, i32, i32, i32, vp]
    lib.afx_linear_bf16_splitk.argtypes = [vp, i64, vp, i64, vp, vp, i32, i32, i32, i32, vp]
    lib.afx_finish_f32_bf16.argtypes = [vp, i32, vp, i64, vp, i64, i64, i32, vp]
    lib.afx_quant_rows_fp8.argtypes = [vp, i64, vp, i64, vp, i32, i32, vp]
    lib.afx_linear_fp8.argtypes = [vp, i64, vp, vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, i32, vp, i64, i32, vp, i64, vp]
    lib.afx_quant_rows_mx8.argtypes = [vp, i64, vp, i64, vp, i64, i32, i32, vp]
    lib.afx_linear_fp8_to_mx8.argtypes = [vp, i64, vp, i64, vp, vp, i64, vp, vp, vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, i32, vp]
    lib.afx_linear_fp8_mx.argtypes = [vp, i64, vp, i64, vp, vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, i32, vp, i64, i32, vp, i64, vp]
    lib.afx_linear_splitk_chunks.argtypes = [i32, i32, i32, i32]
    lib.afx_linear_bf16_pre.argtypes = [vp, i64, vp, i64, vp, vp, i64, i32, i32, i32, i32, i32, vp, i64, i32, vp, i64, vp, i64, vp]
    lib.afx_linear_bf16_sk.argtypes = [vp, i64, vp, i64, vp, vp, i64, i32, i32, i32, i32, i32, vp, i64, i32, vp, i64, vp, vp]
    lib.afx_linear_sk_ws_bytes.argtypes = []
    lib.afx_linear_sk_ws_bytes.restype = i64
    lib.afx_mmdit_prepare_steps.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    lib.afx_mmdit_use_prepared_step.argtypes = [vp, i32]
    lib.afx_gemm_set_mode.argtypes = [i32, i32]
    lib.afx_gemm_set_mode.restype = i32
    lib.afx_gemm_dropres_available.argtypes = []
    lib.afx_gemm_dropres_available.restype = i32
    lib.afx_attn_set_impl.argtypes = [i32]
    lib.afx_attn_set_impl.restype = i32
    lib.afx_attn_bwd_set_impl.argtypes = [i32]
    lib.afx_attn_bwd_set_impl.restype = i32
    lib.afx_lora_dropout_bf16.argtypes = [vp, i64, vp, i64, i64, i32, i64, f32, C.c_uint32, i32, vp]
    lib.afx_mmdit_forward_stage.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, i32, vp]
    lib.afx_mmdit_import_tokens.argtypes = [vp, vp, i32, i32, i32, vp]
    lib.afx_coldot_bf16.argtypes = [vp, i64, vp, i64, vp, i32, i32, vp]
    lib.afx_gate_residual_bf16.argtypes = [vp, i64, vp, vp, i64, vp, i64, i64, i32, vp]
    lib.afx_gemv_t_bf16.argtypes = [vp, i64, vp, i64, vp, i32, i64, i32, vp]
    lib.afx_set_temb_override.argtypes = [vp, vp]
    lib.afx_set_fp8_linear.argtypes = [vp, i32]
    lib.afx_linear_splitk_chunks.restype = i32
    lib.afx_attention_ext_ws_bytes.argtypes = [i32, i32, i32, i32]
    lib.afx_attention_ext_ws_bytes.restype = i64
    lib.afx_attention_ext_bf16.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, vp, i32, i32, i32, i32, i32, f32, i32, vp, vp]
    lib.afx_arcflow_step_dropout.argtypes = [vp, vp, vp, vp, i32, vp, vp, f32, vp, i32, i32, i32, i32, i32, vp]
    lib.afx_arcflow_backward.argtypes = [vp, vp, vp, vp, i32, f32, f32, f32, vp, vp, f32, f32, vp, vp, vp, i32, i32, i32,
                                         i32, i32, i32, i32, vp]
    lib.afx_mse_loss.argtypes = [vp, vp, f32, vp, vp, i64, vp]
    lib.afx_euler_roll.argtypes = [vp, vp, vp, vp, vp, i32, i64, vp]
    lib.afx_axpby_rows.argtypes = [vp, vp, vp, vp, vp, i32, i64, vp]
    lib.afx_cfg_combine.argtypes = [vp, vp, f32, vp, i64, vp]
    lib.afx_head_grad.argtypes = [vp, vp, vp, vp, vp, i64, i64, i32, i32, i32, vp]
    lib.afx_linear_bf16_f32out.argtypes = [vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, vp]
    lib.afx_linear_tn_f32out.argtypes = [vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, vp]
    lib.afx_linear_tn_f32out_ws.argtypes = [vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, vp, vp]
    lib.afx_linear_tn_ws_bytes.argtypes = [i32, i32, i32]
    lib.afx_linear_tn_ws_bytes.restype = i64
    lib.afx_linear_bf16_dropres.argtypes = [vp, i64, vp, i64, vp, i64, i32, i32, i32, vp, i64, f32, C.c_uint32, i64, vp]
    lib.afx_transpose_bf16.argtypes = [vp, i64, vp, i64, i32, i32, vp]
    lib.afx_colsum_bf16.argtypes = [vp, i64, vp, i32, i32, vp]
    lib.afx_normout_backward.argtypes = [vp, i64, vp, i64, vp, i32, i32, i32, vp]
    lib.afx_normout_backward_split.argtypes = [vp, i64, vp, i64, vp, vp, i32, i32, vp]
    lib.afx_outer_accum.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    lib.afx_mmdit_export.argtypes = [vp, C.c_char_p, vp, i32, i32, i32, vp]
    lib.afx_sumsq.argtypes = [vp, vp, i64, vp]
    lib.afx_adamw_step.argtypes = [vp, vp, vp, vp, f32, f32, f32, f32, f32, i32, f32, i64, vp]
    lib.afx_adamw8bit_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, i32, f32, i64, vp]
    lib.afx_ema_lerp.argtypes = [vp, vp, f32, i64, vp]
    lib.afx_cast_f32_bf16.argtypes = [vp, vp, i64, vp]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int:       # default restype: status code
            fn.restype = C.c_int32
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise ArcflowHipError(f'libarcflow_hip error {rc}: {load().afx_last_error().decode()}')
