"""Audio resampling (reference src/utils/audio_utils.rs:66-255, 590-616) on the CPU: the oracle restatement against an
independent float64 evaluation of the same windowed-sinc definition, shape/edge behaviour, and signal-level sanity."""
import math

import numpy as np
import pytest

from oracle import audio_pre as ap


def direct_f64(x, orig_sr, new_sr, lpw=6, rolloff=0.99):
    """out[n] = sum_m x[m] * h(n/new - m/orig), h(tau) = scale * sinc(pi * base * tau) * cos^2(pi * base * tau / (2 lpw)) on
    |base * tau| <= lpw -- the continuous-time form of the polyphase kernel, evaluated per output sample in float64."""
    g = math.gcd(orig_sr, new_sr)
    orig, new = orig_sr // g, new_sr // g
    base = min(orig, new) * rolloff
    width = math.ceil(lpw * orig / base)
    x = np.asarray(x, np.float64)
    n_out = min(math.ceil(new * len(x) / orig), (len(x) // orig + 1) * new)
    out = np.zeros(n_out)
    for n in range(n_out):
        i, j = divmod(n, new)
        ks = np.arange(-width, width + orig)            # input offsets relative to i * orig
        m = i * orig + ks
        ok = (m >= 0) & (m < len(x))
        t = np.clip((-j / new + ks / orig) * base, -lpw, lpw)
        w = np.cos(t * math.pi / lpw / 2.0) ** 2
        ts = t * math.pi
        s = np.where(ts == 0, 1.0, np.sin(ts) / np.where(ts == 0, 1.0, ts))
        out[n] = np.sum((s * w * (base / orig))[ok] * x[m[ok]])
    return out


@pytest.mark.parametrize("orig,new", [(44100, 16000), (48000, 16000), (8000, 16000), (22050, 16000), (16000, 24000), (11025, 16000)])
def test_restatement_matches_direct_evaluation(orig, new):
    g = np.random.default_rng(orig + new)
    x = np.clip(g.normal(0, 0.2, 700), -1, 1).astype(np.float32)
    got = ap.resample_simple(x, orig, new)
    ref = direct_f64(x, orig, new)
    assert got.dtype == np.float32 and got.shape == ref.shape
    # f32 kernel (sin / cos of f32 arguments up to 6 pi) and f32 accumulation over <= 2*width + orig taps
    assert np.abs(got - ref).max() <= 2e-5


def test_kernel_shape_and_width():
    k, width = ap.get_sinc_resample_kernel(44100, 16000, 100)
    assert width == math.ceil(6 * 441 / (160 * 0.99)) == 17 and k.shape == (160, 2 * 17 + 441)
    k, width = ap.get_sinc_resample_kernel(48000, 16000, 16000)
    assert width == math.ceil(6 * 3 / 0.99) == 19 and k.shape == (1, 41)
    # every polyphase branch has (nearly) unit DC gain when downsampling: a constant stays the same constant
    assert np.allclose(ap.get_sinc_resample_kernel(44100, 16000, 100)[0].sum(1), 1.0, atol=2e-3)
    with pytest.raises(ValueError):
        ap.get_sinc_resample_kernel(0, 16000, 1)


@pytest.mark.parametrize("n", [0, 1, 2, 440, 441, 442, 1000])
def test_output_length(n):
    x = np.ones(n, np.float32)
    y = ap.resample_simple(x, 44100, 16000)
    assert y.shape[0] == min(math.ceil(160 * n / 441), (n // 441 + 1) * 160)
    assert ap.resample_simple(x, 16000, 16000).shape[0] == n            # equal rates: unchanged (audio_utils.rs:227-229)


def test_sine_survives_and_above_nyquist_is_removed():
    sr, new = 48000, 16000
    t = np.arange(4800) / sr
    low = np.sin(2 * np.pi * 1000 * t).astype(np.float32)
    y = ap.resample_simple(low, sr, new)
    tt = np.arange(len(y)) / new
    assert np.abs(y[100:-100] - np.sin(2 * np.pi * 1000 * tt)[100:-100]).max() < 2e-3
    high = np.sin(2 * np.pi * 12000 * t).astype(np.float32)            # above the new Nyquist (8 kHz): filtered out
    assert np.abs(ap.resample_simple(high, sr, new)[100:-100]).max() < 2e-2


def test_channel_mean_and_passthrough():
    g = np.random.default_rng(0)
    st = g.normal(0, 0.1, (500, 2)).astype(np.float32)
    mono = ap.resample_audio_from_vec_f32(st.reshape(-1), 2, 16000, 16000)
    np.testing.assert_array_equal(mono, ((st[:, 0] + st[:, 1]) / np.float32(2)).astype(np.float32))
    np.testing.assert_array_equal(ap.resample_audio_from_vec_f32(st[:, 0].copy(), 1, 16000, None), st[:, 0])
    a = ap.resample_audio_from_vec_f32(st.reshape(-1), 2, 44100, 16000)
    np.testing.assert_array_equal(a, ap.resample_simple(mono, 44100, 16000))
    # a trailing partial frame is dropped (audio_vec[0..frame_len * channels])
    assert ap.resample_audio_from_vec_f32(np.ones(7, np.float32), 2, 16000, 16000).shape[0] == 3


@pytest.mark.parametrize("orig_sr,new_sr", [(44100, 16000), (48000, 16000), (8000, 16000), (22050, 16000), (11025, 16000), (16000, 24000)])
def test_library_taps_against_the_restatement(orig_sr, new_sr):
    """The product's host-side kernel builder (csrc/audio_pre.hip sinc_resample_taps, libm sinf / cosf) against the numpy
    restatement: same width / length, taps equal to a few f32 ulps of the kernel's scale (sin / cos implementations differ)."""
    import ctypes as C
    from aha_amd import _lib
    g = math.gcd(orig_sr, new_sr)
    orig, new = orig_sr // g, new_sr // g
    ref, width = ap.get_sinc_resample_kernel(orig_sr, new_sr, g)
    cap = ref.size
    buf = (C.c_float * cap)()
    w, k = C.c_int32(), C.c_int32()
    n = _lib.lib().aha_hip_debug_resample_taps(orig, new, buf, cap, C.byref(w), C.byref(k))
    assert n == ref.size and w.value == width and k.value == ref.shape[1]
    got = np.frombuffer(buf, dtype=np.float32).reshape(ref.shape)
    assert np.abs(got - ref).max() <= 4e-7 * max(1.0, float(np.abs(ref).max()))
