"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy f32) of the reference's audio resampling step.  Nothing under aha_amd/
may import this module.

Path: `resample_audio_from_vec_f32` (reference src/utils/audio_utils.rs:590-616) -> `resample_simple` (:247-255) -> `resample`
(:216-245) -> `get_sinc_resample_kernel` (:66-151) + `apply_sinc_resample_kernel` (:154-214).  The code is the reference's own
(a port of torchaudio.functional.resample); only the tensor primitives it calls (arange / affine / cos / sin / conv1d) are
candle's.  PARITY UNPINNED: the reference holds no test vector for it, torchaudio is not installed here and the reference cannot
be built, so the restatement is checked against an independent float64 evaluation of the same windowed-sinc definition
(tests/test_audio_pre_cpu.py), not against reference output.
"""
from __future__ import annotations

import math

import numpy as np

F = np.float32


def get_sinc_resample_kernel(orig_freq: int, new_freq: int, gcd_val: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """audio_utils.rs:66-151, SincInterpHann.  Returns (kernels (new, klen) f32, width)."""
    if orig_freq <= 0 or new_freq <= 0:
        raise ValueError("Frequencies must be positive")
    if lowpass_filter_width <= 0:
        raise ValueError("Low pass filter width should be positive")
    orig, new = orig_freq // gcd_val, new_freq // gcd_val
    base_freq = float(min(orig, new)) * rolloff
    width = int(math.ceil(lowpass_filter_width * orig / base_freq))
    idx = np.arange(-width, width + orig, dtype=F) * F(1.0 / orig)                       # arange(..).affine(1/orig, 0)
    t = (np.arange(0, -new, -1, dtype=F) * F(1.0 / new))[:, None] + idx[None, :]        # arange_step(..).affine(1/new) + idx
    t = (t * F(base_freq)).astype(F)
    t = np.clip(t, F(-lowpass_filter_width), F(lowpass_filter_width))
    window = np.cos(t * F(math.pi / lowpass_filter_width / 2.0), dtype=F) ** 2            # cos().sqr()
    scale = base_freq / orig
    ts = (t * F(math.pi)).astype(F)
    with np.errstate(invalid="ignore", divide="ignore"):
        sinc = np.where(ts == 0, F(1.0), np.sin(ts, dtype=F) / ts).astype(F)
    return (sinc * window * F(scale)).astype(F), width


def apply_sinc_resample_kernel(waveform: np.ndarray, orig_freq: int, new_freq: int, gcd_val: int, kernel: np.ndarray, width: int):
    """audio_utils.rs:154-214 for a (1, length) waveform: zero-pad (width, width + orig), conv1d stride orig, interleave, narrow."""
    orig, new = orig_freq // gcd_val, new_freq // gcd_val
    x = np.asarray(waveform, dtype=F).reshape(-1)
    length = x.shape[0]
    padded = np.concatenate([np.zeros(width, F), x, np.zeros(width + orig, F)])
    klen = kernel.shape[1]
    rows = (padded.shape[0] - klen) // orig + 1
    win = np.lib.stride_tricks.as_strided(padded, shape=(rows, klen), strides=(padded.strides[0] * orig, padded.strides[0]))
    conv = (win.astype(F) @ kernel.T.astype(F)).astype(F)       # (rows, new): out[i, j] = sum_k kernel[j, k] * padded[i*orig + k]
    flat = conv.reshape(-1)
    target = int(math.ceil(new * length / orig))
    return flat[: min(target, flat.shape[0])]


def resample_simple(waveform: np.ndarray, orig_freq: int, new_freq: int) -> np.ndarray:
    """audio_utils.rs:216-255 (lowpass_filter_width 6, rolloff 0.99, SincInterpHann); unchanged when the rates are equal."""
    if orig_freq <= 0 or new_freq <= 0:
        raise ValueError("Frequencies must be positive")
    x = np.asarray(waveform, dtype=F).reshape(-1)
    if orig_freq == new_freq:
        return x.copy()
    g = math.gcd(orig_freq, new_freq)
    kernel, width = get_sinc_resample_kernel(orig_freq, new_freq, g)
    return apply_sinc_resample_kernel(x, orig_freq, new_freq, g, kernel, width)


def resample_audio_from_vec_f32(audio_vec: np.ndarray, channels: int, orig_sr, target_sample_rate) -> np.ndarray:
    """audio_utils.rs:590-616: interleaved samples -> (frames, channels).mean(1) -> resample when both rates are given and differ."""
    a = np.asarray(audio_vec, dtype=F).reshape(-1)
    frame_len = a.shape[0] // channels
    a = a[: frame_len * channels]
    if channels > 1:
        m = a.reshape(frame_len, channels)
        s = np.zeros(frame_len, F)
        for c in range(channels):          # f32 sum in channel order, then / channels (mean_keepdim)
            s = (s + m[:, c]).astype(F)
        a = (s / F(channels)).astype(F)
    if target_sample_rate is not None and orig_sr is not None and target_sample_rate != orig_sr:
        a = resample_simple(a, int(orig_sr), int(target_sample_rate))
    return a
