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


# ---- Qwen3AsrProcessor (qwen3_asr/processor.rs:27-185) and the helpers it calls ---------------------------------------------------
SUPPORT_LANGUAGE = ["Chinese", "English", "Cantonese", "Arabic", "German", "French", "Spanish", "Portuguese", "Indonesian", "Italian",
                    "Korean", "Russian", "Thai", "Vietnamese", "Japanese", "Turkish", "Hindi", "Malay", "Dutch", "Swedish", "Danish",
                    "Finnish", "Polish", "Czech", "Filipino", "Persian", "Greek", "Romanian", "Hungarian", "Macedonian"]   # processor.rs:29-63
DEFAULT_TEMPLATE = ("<|im_start|>system\n<|im_end|>\n<|im_start|>user\n<|audio_start|><|audio_pad|><|audio_end|><|im_end|>\n"
                    "<|im_start|>assistant\n")                                                                            # generate.rs:85
AUDIO_RUN = "<|audio_start|><|audio_pad|><|audio_end|>"


def capitalize_first_letter(s: str) -> str:
    """utils/mod.rs:549-558: first char upper-cased, the rest lower-cased."""
    return s if not s else s[0].upper() + s[1:].lower()


def float_range_normalize(x):
    """common/modules.rs:1353-1368: peak = max |x| (f32); 0 -> unchanged; > 1 -> x * (1 / peak) (Candle affine: the f64 scalar cast to
    the tensor dtype); then clamp to [-1, 1]."""
    import numpy as np
    x = np.asarray(x, dtype=np.float32)
    peak = np.float32(np.abs(x).max()) if x.size else np.float32(0)
    if peak == 0:
        return x.copy()
    if peak > 1:
        x = x * np.float32(1.0 / float(peak))
    return np.clip(x, np.float32(-1), np.float32(1)).astype(np.float32)


def split_audio_into_chunks(wav, sr: int, max_chunk_sec: float):
    """audio_utils.rs:1743-1760: one chunk up to max_chunk_sec (f32 comparison), else pieces of round(max_chunk_sec * sr) samples
    plus the remainder -- pushed even when it is EMPTY (total a multiple of the piece length), as the reference does."""
    import numpy as np
    wav = np.asarray(wav, dtype=np.float32).reshape(-1)
    total = wav.size
    if np.float32(total) / np.float32(sr) <= np.float32(max_chunk_sec):
        return [wav]
    q = np.float32(max_chunk_sec) * np.float32(sr)
    max_len = int(np.floor(q) + (1 if np.float32(q - np.floor(q)) >= np.float32(0.5) else 0))
    out = [wav[i * max_len:(i + 1) * max_len] for i in range(total // max_len)]
    out.append(wav[(total // max_len) * max_len:])
    return out


def extract_audio_url(messages) -> List[str]:
    """audio_utils.rs:740-755: the audio_url.url of every Audio part of the USER messages (untagged parts: a part is Audio when it has
    `type` and `audio_url` and is neither Text (`text`) nor Image (`image_url`), params/chat.rs:608-615)."""
    out = []
    for m in messages:
        if m.get("role") != "user" or not isinstance(m.get("content"), list):
            continue
        for p in m["content"]:
            if isinstance(p, dict) and "type" in p and "text" not in p and "image_url" not in p and isinstance(p.get("audio_url"), dict) \
                    and "url" in p["audio_url"]:
                out.append(p["audio_url"]["url"])
    return out


class Qwen3AsrProcessor:
    """Qwen3AsrProcessor::process_info / process_audio_tensor for the Python host.  `audio_loader(url) -> mono f32 samples at 16 kHz`
    stands for extract_audios = load_audio_with_resample(url, 16000, 1) (media_host.load_audio_with_resample over the GPU
    resampler for WAV; other containers are symphonia's).  The log-mel features themselves are computed by the library from the
    raw samples (MultiModalData.audio_samples), so a chunk's frame count is len // hop and only the index arithmetic lives here."""

    def __init__(self, cfg, audio_loader, sample_rate: int = 16000, max_asr_input_seconds: float = 1200.0, hop_length: int = 160):
        self.cfg, self.audio_loader = cfg, audio_loader
        self.sample_rate, self.max_asr_input_seconds, self.hop = sample_rate, max_asr_input_seconds, hop_length
        self.audio_token = "<|audio_pad|>"

    def validate_language(self, lang: str) -> bool:
        return lang in SUPPORT_LANGUAGE

    def replace_special_tokens(self, text: str, token_len: int) -> str:
        """processor.rs:93-97: the FIRST <|audio_pad|> becomes token_len of them."""
        return text.replace(self.audio_token, "<|audio_placeholder|>" * token_len, 1).replace("<|audio_placeholder|>", self.audio_token)

    def _one(self, render: str, wav, tokenizer):
        from .model import MultiModalData
        import numpy as np
        wav = np.asarray(wav, dtype=np.float32).reshape(-1)
        if wav.size < self.hop:
            raise ValueError("audio chunk shorter than one feature frame (the reference pushes an empty remainder chunk when the length "
                             "is a multiple of the chunk size, audio_utils.rs:1753-1755, and fails in the feature extractor)")
        n_tok = get_feat_extract_output_lengths(wav.size // self.hop)
        return tokenizer.text_encode(self.replace_special_tokens(render, n_tok)), MultiModalData(audio_samples=wav)

    def process_audio_tensor(self, render: str, audio, tokenizer):
        """processor.rs:99-124 (the VAD / streaming entry, default template generate.rs:85): one tensor, length check, normalise."""
        import numpy as np
        audio = np.asarray(audio, dtype=np.float32).reshape(-1)
        if np.float32(audio.size) > np.float32(self.sample_rate) * np.float32(self.max_asr_input_seconds):
            raise ValueError("vad_res orig_audio is too long!")
        return self._one(render, float_range_normalize(audio), tokenizer)

    def process_info(self, messages, render: str, tokenizer, metadata=None):
        """processor.rs:126-179 -> [(input_ids, MultiModalData)] per <= 1200 s chunk (what generate_asr iterates over): repeated audio
        runs in the rendered text collapse into one, `language X'<asr_text>'` is appended for a supported metadata language, the
        number of audio parts must equal the number of runs, every audio is normalised and cut into chunks."""
        count = render.count(AUDIO_RUN)
        if count > 1:
            render = render.replace(AUDIO_RUN * count, AUDIO_RUN)
        if metadata and "language" in metadata:
            lang = capitalize_first_letter(metadata["language"])
            if self.validate_language(lang):
                render = f"{render}language {lang}'<asr_text>'"
        wavs = [float_range_normalize(self.audio_loader(u)) for u in extract_audio_url(messages)]
        if len(wavs) != count:
            raise ValueError("audio_pad num != audio num")
        out = []
        for w in wavs:
            for chunk in split_audio_into_chunks(w, self.sample_rate, self.max_asr_input_seconds):
                out.append(self._one(render, chunk, tokenizer))
        return out
