"""-m gpu: aha_hip_audio_resample (channel mean + sinc/Hann polyphase FIR on the GPU) against the oracle restatement of the
reference's resample_audio_from_vec_f32 (oracle/audio_pre.py).  f32 FIR over <= ~500 taps in a different summation order, taps
from libm sinf/cosf instead of numpy's: absolute 2e-5 on signals in [-1, 1]; equal-rate and mean-only paths are bit-exact."""
import numpy as np
import pytest

from aha_amd import _lib
from aha_amd.audio_host import resample_audio_from_vec_f32
from oracle import audio_pre as ap

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(gpu):
    from aha_amd.model import HipContext
    c = HipContext(0)
    yield c
    c.close()


@pytest.mark.parametrize("orig,new", [(44100, 16000), (48000, 16000), (8000, 16000), (22050, 16000), (11025, 16000), (16000, 24000)])
@pytest.mark.parametrize("channels", [1, 2])
def test_resample_matches_oracle(ctx, orig, new, channels):
    g = np.random.default_rng(orig * 3 + new + channels)
    for frames in (1, 5, 441, 4000, 60001):
        pcm = np.clip(g.normal(0, 0.2, frames * channels), -1, 1).astype(np.float32)
        got = resample_audio_from_vec_f32(ctx.handle, pcm, channels, orig, new)
        ref = ap.resample_audio_from_vec_f32(pcm, channels, orig, new)
        assert got.shape == ref.shape, (frames, got.shape, ref.shape)
        assert np.abs(got - ref).max() <= 2e-5


def test_equal_rates_and_mean_only_are_exact(ctx):
    g = np.random.default_rng(1)
    pcm = g.normal(0, 0.3, 3001 * 3).astype(np.float32)
    got = resample_audio_from_vec_f32(ctx.handle, pcm, 3, 16000, 16000)
    np.testing.assert_array_equal(got, ap.resample_audio_from_vec_f32(pcm, 3, 16000, 16000))
    mono = g.normal(0, 0.3, 777).astype(np.float32)
    np.testing.assert_array_equal(resample_audio_from_vec_f32(ctx.handle, mono, 1, 16000, 16000), mono)


def test_thirty_seconds_of_cd_audio(ctx):
    """The size the ASR path sees (30 s window): 1 323 000 stereo frames at 44.1 kHz -> 480 000 samples at 16 kHz."""
    g = np.random.default_rng(2)
    pcm = np.clip(g.normal(0, 0.1, 1323000 * 2), -1, 1).astype(np.float32)
    got = resample_audio_from_vec_f32(ctx.handle, pcm, 2, 44100, 16000)
    assert got.shape[0] == 480000
    # property: resampling is linear -- R(a x) == a R(x) up to rounding; and a slice far from the edges matches the oracle
    got2 = resample_audio_from_vec_f32(ctx.handle, (pcm * np.float32(0.5)), 2, 44100, 16000)
    assert np.abs(got2 - got * np.float32(0.5)).max() <= 1e-6
    # an interior window against the oracle: output row i (160 samples) reads input frames [441 i - 17, 441 (i + 1) + 17), so the
    # oracle on frames [441 r0, 441 (r0 + 40)) reproduces global rows r0 + 1 .. r0 + 38 (its first / last row see zero padding)
    r0 = 1000
    seg = pcm.reshape(-1, 2)[441 * r0: 441 * (r0 + 40)]
    ref = ap.resample_audio_from_vec_f32(seg.reshape(-1), 2, 44100, 16000)
    assert np.abs(got[(r0 + 1) * 160: (r0 + 39) * 160] - ref[160: 39 * 160]).max() <= 2e-5


def test_argument_errors(ctx):
    x = np.zeros(10, np.float32)
    for args in ((x, 1, 0, 16000), (x, 1, 16000, 0), (x, 0, 16000, 16000)):
        with pytest.raises((_lib.AhaHipError, ZeroDivisionError)):
            resample_audio_from_vec_f32(ctx.handle, *args)
