"""Host-side container decode (aha_amd/media_host.py): the reference's source selection (file:// / data: URIs / raw bytes), its
container sniffing, the per-format normalisation constants of load_audio_use_symphonia and its channel rules, and get_image."""
import base64
import io
import struct
import wave

import numpy as np
import pytest

from aha_amd import media_host as mh


def _wav_bytes(samples: np.ndarray, sr: int, width: int) -> bytes:
    """samples: (frames, channels) integers already in range for `width` bytes."""
    buf = io.BytesIO()
    with wave.open(buf, "wb") as w:
        w.setnchannels(samples.shape[1])
        w.setsampwidth(width)
        w.setframerate(sr)
        if width == 2:
            w.writeframes(samples.astype("<i2").tobytes())
        else:
            raw = bytearray()
            for v in samples.reshape(-1):
                raw += struct.pack("<i", int(v))[:3]
            w.writeframes(bytes(raw))
    return buf.getvalue()


def _wav_f32_bytes(samples: np.ndarray, sr: int) -> bytes:
    data = samples.astype("<f4").tobytes()
    ch = samples.shape[1]
    fmt = struct.pack("<HHIIHH", 3, ch, sr, sr * ch * 4, ch * 4, 32)
    return b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(data)) + b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + \
        b"data" + struct.pack("<I", len(data)) + data


def test_format_sniffing_matches_the_reference_table():
    assert mh.get_audio_format_from_bytes(b"RIFF\0\0\0\0WAVEfmt ") == "wav"
    assert mh.get_audio_format_from_bytes(b"RIFF\0\0\0\0AVI LIST") == "riff"
    assert mh.get_audio_format_from_bytes(b"\xff\xfb" + b"\0" * 10) == "mp3"
    assert mh.get_audio_format_from_bytes(b"ID3" + b"\0" * 9) == "mp3"
    assert mh.get_audio_format_from_bytes(b"fLaC" + b"\0" * 8) == "flac"
    assert mh.get_audio_format_from_bytes(b"OggS" + b"\0" * 8) == "ogg"
    assert mh.get_audio_format_from_bytes(b"FORM" + b"\0" * 8) == "aiff"
    assert mh.get_audio_format_from_bytes(b"\0\0\0\x20mp4 " + b"\0" * 4) == "m4a"
    with pytest.raises(ValueError, match="too short"):
        mh.get_audio_format_from_bytes(b"RIFF")
    with pytest.raises(ValueError, match="Unknown format"):
        mh.get_audio_format_from_bytes(b"\0" * 16)


def test_s16_s24_f32_normalisation_and_channel_rules(tmp_path):
    g = np.random.default_rng(0)
    s16 = g.integers(-32768, 32768, size=(500, 2))
    a, sr = mh.load_audio_use_symphonia(_wav_bytes(s16, 44100, 2), target_channels=2)
    assert sr == 44100 and a.shape == (2, 500) and a.dtype == np.float32
    np.testing.assert_array_equal(a, (s16.T.astype(np.float32) / np.float32(32768.0)))          # S16: s / 32768 (audio_utils.rs:537)
    mono, _ = mh.load_audio_use_symphonia(_wav_bytes(s16, 44100, 2), target_channels=1)
    np.testing.assert_allclose(mono[0], a.mean(axis=0), rtol=0, atol=1e-7)                        # multi -> mono: mean over channels
    s24 = g.integers(-(1 << 23), 1 << 23, size=(300, 1))
    b, _ = mh.load_audio_use_symphonia(_wav_bytes(s24, 16000, 3), target_channels=1)
    np.testing.assert_array_equal(b[0], s24[:, 0].astype(np.float32) / np.float32(8388608.0))   # S24: s / 8388608 (:552)
    rep, _ = mh.load_audio_use_symphonia(_wav_bytes(s24, 16000, 3), target_channels=2)           # mono -> multi: repeat
    np.testing.assert_array_equal(rep[0], rep[1])
    f = g.standard_normal((200, 1)).astype(np.float32)
    c, sr = mh.load_audio_use_symphonia(_wav_f32_bytes(f, 22050), target_channels=1)
    assert sr == 22050
    np.testing.assert_array_equal(c[0], f[:, 0])                                                 # F32: as is
    with pytest.raises(ValueError, match="can't change directly"):
        mh.load_audio_use_symphonia(_wav_bytes(s16, 44100, 2), target_channels=3)
    with pytest.raises(NotImplementedError):
        mh.load_audio_use_symphonia(b"fLaC" + b"\0" * 32)


def test_audio_sources(tmp_path):
    s16 = np.arange(-50, 50).reshape(-1, 1)
    blob = _wav_bytes(s16, 16000, 2)
    p = tmp_path / "a b.wav"
    p.write_bytes(blob)
    assert mh.get_audio_bytes_vec("file://" + str(p)) == blob
    assert mh.get_audio_bytes_vec("data:audio/wav;base64," + base64.b64encode(blob).decode()) == blob
    assert mh.get_audio_bytes_vec(blob) == blob
    with pytest.raises(RuntimeError, match="get audio path error"):
        mh.get_audio_bytes_vec("/not/a/url/and/not/a/container")
    # equal rates: no resampler (and no GPU) involved
    out = mh.load_audio_with_resample(None, "file://" + str(p), target_sample_rate=16000)
    np.testing.assert_array_equal(out, s16[:, 0].astype(np.float32) / np.float32(32768.0))


def test_get_image_sources(tmp_path):
    from PIL import Image
    g = np.random.default_rng(1)
    rgb = g.integers(0, 256, size=(17, 23, 3), dtype=np.uint8)
    p = tmp_path / "x.png"
    Image.fromarray(rgb).save(p)
    np.testing.assert_array_equal(mh.get_image("file://" + str(p)), rgb)
    b64 = base64.b64encode(p.read_bytes()).decode()
    np.testing.assert_array_equal(mh.get_image("data:image/png;base64," + b64), rgb)
    grey = Image.fromarray(rgb[:, :, 0])            # to_rgb8 of a luma image replicates the channel
    q = tmp_path / "g.png"
    grey.save(q)
    got = mh.get_image("file://" + str(q))
    assert got.shape == (17, 23, 3) and (got[:, :, 0] == got[:, :, 2]).all()
    with pytest.raises(RuntimeError, match="get image from message failed"):
        mh.get_image(str(p))                        # a bare path is not one of the reference's three forms
    with pytest.raises(RuntimeError, match="Failed to open file"):
        mh.get_image("file://" + str(tmp_path / "missing.png"))
