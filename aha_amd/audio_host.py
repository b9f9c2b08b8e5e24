"""Host-side pieces of the Qwen3-ASR request (mirror of /root/reference/src/models/qwen3_asr/processor.rs).

Only index arithmetic lives here; the log-mel frontend and the encoder run on the GPU (csrc/audio_tower.hip)."""
from __future__ import annotations

from typing import List, Sequence


def get_feat_extract_output_lengths(n_frames: int) -> int:
    """Audio tokens produced by `n_frames` log-mel frames (processor.rs:187-195): full 100-frame windows give 13
    tokens each, the remainder goes through the three stride-2 convolutions."""
    r = n_frames % 100
    full = (n_frames // 100) * 13
    if r == 0:
        return full
    f = (r - 1) // 2 + 1
    return ((f - 1) // 2 + 1 - 1) // 2 + 1 + full


def audio_prompt_ids(cfg, n_samples: int, prefix: Sequence[int], suffix: Sequence[int]) -> List[int]:
    """prefix + <|audio_start|> + n x <|audio_pad|> + <|audio_end|> + suffix for `n_samples` of 16 kHz audio
    (hop 160 -> n_samples // 160 frames; the default template is qwen3_asr/generate.rs:85)."""
    n_tok = get_feat_extract_output_lengths(n_samples // 160)
    return list(prefix) + [cfg.audio_start_token_id] + [cfg.audio_token_id] * n_tok + [cfg.audio_end_token_id] + list(suffix)


def resample_audio_from_vec_f32(ctx_handle, audio_vec, channels: int, orig_sr: int, target_sample_rate: int):
    """resample_audio_from_vec_f32 (reference src/utils/audio_utils.rs:590-616) through aha_hip_audio_resample: interleaved PCM
    f32 -> mono f32 at target_sample_rate (channel mean + sinc/Hann resampling on the GPU).  ``ctx_handle`` is the aha_ctx the
    model was created on (HipInferenceModel.ctx)."""
    import ctypes as C

    import numpy as np

    from ._lib import check, lib
    a = np.ascontiguousarray(np.asarray(audio_vec, dtype=np.float32).reshape(-1))
    frames = a.size // channels
    pf = C.POINTER(C.c_float)
    n = check(lib().aha_hip_audio_resample(ctx_handle, a.ctypes.data_as(pf), frames, channels, orig_sr, target_sample_rate, None, 0))
    out = np.empty(max(n, 1), dtype=np.float32)
    n = check(lib().aha_hip_audio_resample(ctx_handle, a.ctypes.data_as(pf), frames, channels, orig_sr, target_sample_rate,
                                           out.ctypes.data_as(pf), out.size))
    return out[:n]
