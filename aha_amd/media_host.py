"""Host-side container decode in front of the GPU preprocessing: where a request's image / audio bytes come
from and how they become the arrays the C ABI takes.  Mirrors of the reference's *selection and conversion* logic; the codecs themselves
are third-party on both sides (reference: crates `image`, `symphonia`; here Pillow and a RIFF/WAVE PCM reader) and are not restated.

  get_image                  /root/reference/src/utils/img_utils.rs:55-89   (file://, data:image;base64 -- http(s) is out: no network)
  get_audio_format_from_bytes /root/reference/src/utils/audio_utils.rs:432-471
  get_audio_bytes_vec        audio_utils.rs:345-379
  load_audio_use_symphonia   audio_utils.rs:476-588  (WAV only here: S16 / 32768, S24 / 8388608, F32 as is; channel repeat / mean)
  load_audio_with_resample   audio_utils.rs:636-647  (decode -> aha_hip_audio_resample through audio_host.resample_audio_from_vec_f32)
"""
from __future__ import annotations

import base64
import io
import struct
from typing import Tuple
from urllib.parse import unquote, urlparse

import numpy as np


def _file_url_path(s: str) -> str:
    """file:// URL -> local path (url::Url::to_file_path, falling back to dropping the 7-character prefix as the reference does)."""
    try:
        u = urlparse(s)
        if u.scheme == "file" and u.netloc in ("", "localhost"):
            return unquote(u.path)
    except ValueError:
        pass
    return s[7:]


def get_image(file: str) -> np.ndarray:
    """img_utils.rs:55-89 -> DynamicImage; returned as the (H, W, 3) uint8 RGB array `to_rgb8` yields (what img_smart_resize /
    aha_hip_image_resize / process_images take)."""
    from PIL import Image
    img = None
    if file.startswith("http://") or file.startswith("https://"):
        raise RuntimeError("get image from url: no network in this environment")
    if file.startswith("file://"):
        try:
            img = Image.open(_file_url_path(file))
            img.load()
        except OSError as e:
            raise RuntimeError(f"Failed to open file: {e}") from e
    if file.startswith("data:image") and "base64," in file:
        data = file.split("base64,")[1]
        try:
            img = Image.open(io.BytesIO(base64.b64decode(data)))
            img.load()
        except Exception as e:  # noqa: BLE001
            raise RuntimeError(f"Failed to decode image: {e}") from e
    if img is None:
        raise RuntimeError("get image from message failed")
    return np.asarray(img.convert("RGB"), dtype=np.uint8)


def get_audio_format_from_bytes(b: bytes) -> str:
    """audio_utils.rs:432-471: container sniffing by magic bytes."""
    if len(b) < 12:
        raise ValueError(f"bytes too short: {len(b)}")
    if b[:4] == b"RIFF":
        return "wav" if b[8:12] == b"WAVE" else "riff"
    if b[:2] in (b"\xff\xfb", b"\xff\xf3", b"\xff\xf2"):
        return "mp3"
    if b[:3] == b"ID3":
        return "mp3"
    if b[:4] == b"FORM":
        return "aiff"
    if b[:4] == b"OggS":
        return "ogg"
    if b[:4] == b"fLaC":
        return "flac"
    if b[4:8] == b"mp4 ":
        return "m4a"
    if b[4:8] == b"mp4a":
        return "mp4"
    raise ValueError("Unknown format ")


def get_audio_bytes_vec(path_str) -> bytes:
    """audio_utils.rs:345-379: file:// path, data:audio base64, or the container bytes themselves."""
    if isinstance(path_str, (bytes, bytearray)):
        get_audio_format_from_bytes(bytes(path_str))
        return bytes(path_str)
    if path_str.startswith("http://") or path_str.startswith("https://"):
        raise RuntimeError("load audio from url: no network in this environment")
    if path_str.startswith("file://"):
        with open(_file_url_path(path_str), "rb") as f:
            return f.read()
    if path_str.startswith("data:audio") and "base64," in path_str:
        return base64.b64decode(path_str.split("base64,")[1])
    raw = path_str.encode("latin-1", errors="replace")
    try:
        get_audio_format_from_bytes(raw)
    except ValueError as e:
        raise RuntimeError(f"get audio path error {path_str[:64]}, et_audio_format error: {e}") from e
    return raw


def _wav_pcm(b: bytes) -> Tuple[np.ndarray, int]:
    """RIFF/WAVE -> (channels, frames) f32 in the sample types symphonia hands to the reference's match arms."""
    if b[:4] != b"RIFF" or b[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE stream")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(b):
        cid, size = b[pos:pos + 4], struct.unpack("<I", b[pos + 4:pos + 8])[0]
        body = b[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
            if fmt[0] == 0xFFFE and len(body) >= 26:   # WAVE_FORMAT_EXTENSIBLE: the sub-format's first two bytes are the real tag
                fmt = (struct.unpack("<H", body[24:26])[0],) + fmt[1:]
        elif cid == b"data":
            data = body
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError("WAVE stream without fmt / data chunk")
    tag, ch, sr, _, _, bits = fmt
    if tag == 3 and bits == 32:                        # AudioBufferRef::F32: as is
        x = np.frombuffer(data[:len(data) // 4 * 4], dtype="<f4").astype(np.float32)
    elif tag == 1 and bits == 16:                      # S16: s / 32768
        x = np.frombuffer(data[:len(data) // 2 * 2], dtype="<i2").astype(np.float32) / np.float32(32768.0)
    elif tag == 1 and bits == 24:                      # S24: s / 8388608
        raw = np.frombuffer(data[:len(data) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = raw[:, 0] | (raw[:, 1] << 8) | (raw[:, 2] << 16)
        v = np.where(v & 0x800000, v - 0x1000000, v)
        x = v.astype(np.float32) / np.float32(8388608.0)
    else:
        raise ValueError(f"unsupported WAVE sample format (tag {tag}, {bits} bits): the reference prints '不支持的音频格式' and yields no samples")
    frames = x.size // ch
    return np.ascontiguousarray(x[:frames * ch].reshape(frames, ch).T), int(sr)


def load_audio_use_symphonia(audio_vec: bytes, target_channels: int = 1) -> Tuple[np.ndarray, int]:
    """audio_utils.rs:476-588: decode, then bring the channel count to target_channels (mono -> repeat, multi -> mean over channels,
    anything else is an error).  Returns ((target_channels, frames) f32, sample_rate)."""
    ext = get_audio_format_from_bytes(audio_vec)
    if ext != "wav":
        raise NotImplementedError(f"container '{ext}': only RIFF/WAVE PCM is decoded here (the reference leaves mp3 / flac / ogg to symphonia)")
    audio, sr = _wav_pcm(audio_vec)
    channels = audio.shape[0]
    if target_channels == channels:
        return audio, sr
    if channels == 1:
        return np.repeat(audio, target_channels, axis=0), sr
    if target_channels == 1:
        return audio.mean(axis=0, keepdims=True, dtype=np.float32), sr
    raise ValueError(f"target_channels: {target_channels}, audio channels: {channels}, can't change directly")


def load_audio_with_resample(ctx_handle, path, target_sample_rate=16000, target_channels: int = 1) -> np.ndarray:
    """audio_utils.rs:636-647 -> resample_audio_from_bytes (:620-633): decode to target_channels (the channel mean happens in
    load_audio_use_symphonia), then resample_simple when the rates differ -- on the GPU here (aha_hip_audio_resample through
    audio_host.resample_audio_from_vec_f32, called with one channel).  Returns the mono f32 signal."""
    from .audio_host import resample_audio_from_vec_f32
    if target_channels != 1:
        raise NotImplementedError("the Qwen3-ASR path asks for one channel")
    audio, sr = load_audio_use_symphonia(get_audio_bytes_vec(path), target_channels=1)
    if target_sample_rate is None or int(target_sample_rate) == sr:
        return audio[0]
    return resample_audio_from_vec_f32(ctx_handle, audio[0], 1, sr, int(target_sample_rate))
