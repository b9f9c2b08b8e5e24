"""ctypes binding of libaha_hip.so -- the same C ABI (include/aha_hip.h) the Rust shim in INTEGRATION.md binds.

There is deliberately no fallback: if the HIP library is missing or cannot be loaded this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libaha_hip.so")

AHA_BF16, AHA_F16, AHA_F32, AHA_U32, AHA_U8 = 0, 1, 2, 3, 4
AHA_ARCH_QWEN3, AHA_ARCH_QWEN3VL, AHA_ARCH_QWEN3ASR = 0, 1, 2
ACT_NONE, ACT_GELU_TANH, ACT_GELU_ERF, ACT_SILU, ACT_SILU_MUL_PAIRS = 0, 1, 2, 3, 4


class ModelDesc(C.Structure):
    _fields_ = [
        ("arch", C.c_int32),
        ("hidden_size", C.c_int32), ("intermediate_size", C.c_int32), ("num_hidden_layers", C.c_int32),
        ("num_attention_heads", C.c_int32), ("num_key_value_heads", C.c_int32), ("head_dim", C.c_int32),
        ("vocab_size", C.c_int32),
        ("rms_norm_eps", C.c_float), ("rope_theta", C.c_float),
        ("tie_word_embeddings", C.c_int32),
        ("mrope_section", C.c_int32 * 3),
        ("vis_depth", C.c_int32), ("vis_hidden_size", C.c_int32), ("vis_num_heads", C.c_int32),
        ("vis_intermediate_size", C.c_int32), ("vis_in_channels", C.c_int32), ("vis_patch_size", C.c_int32),
        ("vis_temporal_patch_size", C.c_int32), ("vis_spatial_merge_size", C.c_int32),
        ("vis_out_hidden_size", C.c_int32), ("vis_num_position_embeddings", C.c_int32),
        ("vis_deepstack_indexes", C.c_int32 * 8),
        ("vis_num_deepstack", C.c_int32),
        ("image_token_id", C.c_int32), ("video_token_id", C.c_int32),
        ("vision_start_token_id", C.c_int32), ("vision_end_token_id", C.c_int32),
        ("kv_reserve_tokens", C.c_int32),
        ("n_stop_tokens", C.c_int32),
        ("stop_tokens", C.c_uint32 * 8),
        ("aud_d_model", C.c_int32), ("aud_encoder_layers", C.c_int32), ("aud_attention_heads", C.c_int32),
        ("aud_ffn_dim", C.c_int32), ("aud_num_mel_bins", C.c_int32), ("aud_downsample_hidden_size", C.c_int32),
        ("aud_output_dim", C.c_int32), ("aud_n_window", C.c_int32),
        ("audio_token_id", C.c_int32),
        ("tp_rank", C.c_int32), ("tp_size", C.c_int32),
        ("compute_dtype", C.c_int32),
    ]


class TensorView(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("dtype", C.c_int32), ("ndim", C.c_int32),
                ("shape", C.c_int64 * 5), ("on_device", C.c_int32)]


class MmInput(C.Structure):
    _fields_ = [("pixel_values", C.c_void_p), ("pixel_dtype", C.c_int32), ("n_patches", C.c_int64),
                ("image_grid_thw", C.POINTER(C.c_uint32)), ("n_images", C.c_int32),
                ("audio_features", C.POINTER(C.c_float)), ("n_frames", C.c_int64),
                ("audio_samples", C.POINTER(C.c_float)), ("n_samples", C.c_int64),
                ("image_embeds", C.c_void_p), ("n_image_tokens", C.c_int64),
                ("pixel_values_video", C.c_void_p), ("n_patches_video", C.c_int64),
                ("video_grid_thw", C.POINTER(C.c_uint32)), ("n_videos", C.c_int32)]


# every symbol include/aha_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p)
REDUCE_SCATTER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p)   # (buf_f32_dev, count_per_rank, user)
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p)       # (buf_dev, bytes_per_rank, user)

SIGNATURES = {
    "aha_hip_init": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "aha_hip_shutdown": (None, [_P]),
    "aha_hip_last_error": (C.c_char_p, []),
    "aha_hip_version": (C.c_char_p, []),
    "aha_hip_get_dtype": (C.c_int, [C.c_int32, C.c_char_p, C.POINTER(C.c_int32)]),
    "aha_hip_check_dtype": (C.c_int, [C.c_int32]),
    "aha_hip_model_create": (C.c_int, [_P, C.POINTER(ModelDesc), C.POINTER(TensorView), C.c_size_t, C.POINTER(_P)]),
    "aha_hip_model_destroy": (None, [_P]),
    "aha_hip_forward_initial": (C.c_int, [_P, C.POINTER(C.c_uint32), C.c_size_t, C.c_size_t, C.POINTER(MmInput),
                                          C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    "aha_hip_forward_step": (C.c_int, [_P, C.c_uint32, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    "aha_hip_clear_cache": (C.c_int, [_P]),
    "aha_hip_stop_token_ids": (C.c_int, [_P, C.POINTER(C.c_uint32), C.c_size_t]),
    "aha_hip_decode_greedy": (C.c_int, [_P, C.c_uint32, C.c_size_t, C.c_size_t, C.POINTER(C.c_uint32)]),
    "aha_hip_sample_candidates": (C.c_int, [_P, C.POINTER(C.c_uint32), C.c_size_t, C.c_float, C.c_float, C.c_int32,
                                            C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_float),
                                            C.POINTER(C.c_float)]),
    "aha_hip_last_logits": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "aha_hip_rng_create": (C.c_int, [C.c_uint64, C.POINTER(_P)]),
    "aha_hip_rng_destroy": (None, [_P]),
    "aha_hip_rng_next_u32": (C.c_uint32, [_P]),
    "aha_hip_rng_weighted_index": (C.c_int, [_P, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_uint32)]),
    "aha_hip_debug_chacha_block": (C.c_int, [C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint32)]),
    "aha_hip_img_smart_resize": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32),
                                           C.POINTER(C.c_uint32)]),
    "aha_hip_image_resize": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32, _P]),
    "aha_hip_debug_resize_taps": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_float),
                                            C.c_int64]),
    "aha_hip_debug_resample_taps": (C.c_int64, [C.c_int32, C.c_int32, C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_int32),
                                                C.POINTER(C.c_int32)]),
    "aha_hip_audio_resample": (C.c_int64, [_P, C.POINTER(C.c_float), C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                           C.POINTER(C.c_float), C.c_int64]),
    "aha_hip_cache_len": (C.c_size_t, [_P]),
    "aha_hip_debug_steps_executed": (C.c_int64, [_P]),
    "aha_hip_debug_graph_step": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "aha_hip_kv_export": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]),
    "aha_hip_kv_import": (C.c_int, [_P, _P, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_size_t, C.c_int64]),
    "aha_hip_set_profiling": (C.c_int, [_P, C.c_int]),
    "aha_hip_get_profile": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                      C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "aha_hip_debug_scramble_pages": (C.c_int, [_P, C.c_int]),
    "aha_hip_debug_last_hidden": (C.c_int, [_P, C.POINTER(C.c_float), C.c_size_t]),
    "aha_hip_debug_image_embeds": (C.c_int, [_P, C.c_int, C.POINTER(C.c_float), C.c_size_t]),
    "aha_hip_rmsnorm": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, C.c_float, _P]),
    "aha_hip_gemv": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P, C.c_float, _P, _P]),
    "aha_hip_gemv_gate_up": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, _P, C.c_float, _P]),
    "aha_hip_gemm": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P,
                               C.c_int32, _P]),
    "aha_hip_qknorm_rope": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_float, C.c_float, _P]),
    "aha_hip_attn_decode": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, _P]),
    "aha_hip_attn_prefill": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_float, _P]),
    "aha_hip_argmax": (C.c_int, [_P, C.c_int64, _P, _P]),
    "aha_hip_logmel": (C.c_int, [_P, C.c_int64, _P, _P]),
    "aha_hip_debug_poison_lds": (C.c_int, [C.c_uint32, _P]),
    "aha_hip_debug_gemm_plan": (C.c_int, [C.c_int32, C.c_int32]),
    "aha_hip_debug_attn_variant": (C.c_int, [C.c_int32]),
    "aha_hip_debug_attn_form": (C.c_int, [C.c_int32]),
    "aha_hip_debug_gemm_grouped": (C.c_int, [_P, _P, _P] + [C.c_int32] * 10 + [_P]),
    "aha_hip_debug_streamk_plan": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_size_t, _P, C.c_int32, _P, _P]),
    "aha_hip_set_gemm_reserved_cus": (C.c_int, [C.c_int32]),
    "aha_hip_debug_plan_gemm": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_size_t, _P]),
    "aha_hip_get_rope_index": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_int32, _P, _P]),
    "aha_hip_get_rope_index_mm": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_int32, _P, C.c_int32, _P, _P]),
    "aha_hip_embed": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "aha_hip_config_parse": (C.c_int, [C.c_char_p, _P]),
    "aha_hip_config_torch_dtype": (C.c_int, [C.c_char_p, C.c_char_p, C.c_size_t]),
    "aha_hip_weights_open": (C.c_int, [C.c_char_p, _P]),
    "aha_hip_weights_count": (C.c_size_t, [_P]),
    "aha_hip_weights_get": (C.c_int, [_P, C.c_size_t, _P]),
    "aha_hip_weights_close": (None, [_P]),
    "aha_hip_model_load": (C.c_int, [_P, C.c_char_p, C.c_size_t, _P]),
    "aha_hip_set_allreduce": (C.c_int, [_P, _P, _P]),
    "aha_hip_set_seq_parallel": (C.c_int, [_P, _P, _P, _P]),
    "aha_hip_tp_unique_id": (C.c_int, [_P]),
    "aha_hip_tp_init_rccl": (C.c_int, [_P, _P]),
    "aha_hip_set_context_parallel": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P]),
    "aha_hip_cp_init_rccl": (C.c_int, [_P, _P]),
    "aha_hip_debug_cp_plan": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    "aha_hip_debug_allreduce": (C.c_int, [_P, _P, C.c_size_t]),
    "aha_hip_vision_encode": (C.c_int, [_P, C.POINTER(MmInput), _P, C.POINTER(C.c_int64)]),
    "aha_hip_debug_audio_embeds": (C.c_int, [_P, C.POINTER(C.c_float), C.c_size_t]),
    "aha_hip_image_to_patches": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float),
                                           C.POINTER(C.c_float), _P]),
    "aha_hip_video_smart_resize": (C.c_int, [C.c_uint32] * 8 + [_P, _P]),
    "aha_hip_video_sample_frames": (C.c_int, [C.c_uint32, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P]),
    "aha_hip_video_timestamps": (C.c_int64, [_P, C.c_size_t, C.c_float, C.c_uint32, _P, C.c_size_t]),
    "aha_hip_video_to_patches": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float),
                                           C.POINTER(C.c_float), _P]),
}

_lib = None


class AhaHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"aha_hip error {code}: {msg}")
        self.code = code


def lib() -> C.CDLL:
    """Load libaha_hip.so (once).  Raises if it has not been built -- the product path has no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -m aha_amd.build` (or __graft_entry__.build()) first; "
                              "there is no CPU fallback for the HIP path")
        # Load order matters: torch bundles its own HIP / HSA / RCCL runtime (same SONAMEs as /opt/rocm's).  If
        # libaha_hip.so is dlopen'ed first it pulls in /opt/rocm's copies, torch then mixes both and device discovery
        # fails ("no ROCm-capable device is detected").  Importing torch first makes the process use one runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError here means the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int) -> int:
    if rc < 0:
        raise AhaHipError(rc, lib().aha_hip_last_error().decode(errors="replace"))
    return rc
