"""Model dimension records for the hot path (host-side mirror of the reference's config surface).

The reference reads these from each checkpoint's ``config.json`` at run time:
  * Qwen3Config            -- /root/reference/src/models/qwen3/config.rs:4-27
  * Qwen3VLTextConfig      -- /root/reference/src/models/qwen3vl/config.rs:59-80
  * Qwen3VLVisionConfig    -- /root/reference/src/models/qwen3vl/config.rs:108-123
  * Qwen3VLConfig          -- /root/reference/src/models/qwen3vl/config.rs:125-133
No checkpoint is on disk here, so the published HF values are hard-coded (SURVEY.md section 8 header).
Field names follow the reference structs so a real ``config.json`` can be loaded with ``from_json``.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field, asdict
from typing import List, Optional


@dataclass
class Qwen3Config:
    hidden_size: int
    intermediate_size: int
    num_hidden_layers: int
    num_attention_heads: int
    num_key_value_heads: int
    head_dim: int
    vocab_size: int
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6
    tie_word_embeddings: bool = True
    attention_bias: bool = False
    hidden_act: str = "silu"
    # Qwen3-VL text tower only (reference: RopeScaling, qwen3vl/config.rs:52-57)
    mrope_section: Optional[List[int]] = None
    eos_token_ids: List[int] = field(default_factory=lambda: [151645, 151643])

    @property
    def q_dim(self) -> int:
        return self.num_attention_heads * self.head_dim

    @property
    def kv_dim(self) -> int:
        return self.num_key_value_heads * self.head_dim

    @classmethod
    def from_json(cls, path: str) -> "Qwen3Config":
        with open(path) as f:
            d = json.load(f)
        if "text_config" in d:  # Qwen3-VL style nesting
            top = d
            d = dict(d["text_config"])
            d["tie_word_embeddings"] = top.get("tie_word_embeddings", False)
            d["mrope_section"] = d.get("rope_scaling", {}).get("mrope_section")
        keys = {f for f in cls.__dataclass_fields__}
        return cls(**{k: v for k, v in d.items() if k in keys})

    def to_dict(self):
        return asdict(self)


@dataclass
class Qwen3VLVisionConfig:
    depth: int = 27
    hidden_size: int = 1152
    num_heads: int = 16
    intermediate_size: int = 4304
    in_channels: int = 3
    patch_size: int = 16
    temporal_patch_size: int = 2
    spatial_merge_size: int = 2
    out_hidden_size: int = 4096
    num_position_embeddings: int = 2304
    deepstack_visual_indexes: List[int] = field(default_factory=lambda: [8, 16, 24])
    hidden_act: str = "gelu_pytorch_tanh"

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @property
    def patch_dim(self) -> int:
        return self.in_channels * self.temporal_patch_size * self.patch_size * self.patch_size


@dataclass
class Qwen3VLConfig:
    text: Qwen3Config
    vision: Qwen3VLVisionConfig
    image_token_id: int = 151655
    video_token_id: int = 151656
    vision_start_token_id: int = 151652
    vision_end_token_id: int = 151653
    tie_word_embeddings: bool = False


# ---- the BASELINE.json configurations (SURVEY.md section 8 header) -------------------------------------

def qwen3_0_6b() -> Qwen3Config:
    return Qwen3Config(hidden_size=1024, intermediate_size=3072, num_hidden_layers=28,
                       num_attention_heads=16, num_key_value_heads=8, head_dim=128,
                       vocab_size=151936, rope_theta=1e6, tie_word_embeddings=True)


def qwen3vl_8b_text() -> Qwen3Config:
    return Qwen3Config(hidden_size=4096, intermediate_size=12288, num_hidden_layers=36,
                       num_attention_heads=32, num_key_value_heads=8, head_dim=128,
                       vocab_size=151936, rope_theta=5e6, tie_word_embeddings=False,
                       mrope_section=[24, 20, 20])


def qwen3vl_8b() -> Qwen3VLConfig:
    return Qwen3VLConfig(text=qwen3vl_8b_text(), vision=Qwen3VLVisionConfig(), tie_word_embeddings=False)


def tiny_qwen3(layers: int = 2, hidden: int = 256, heads: int = 4, kv_heads: int = 2,
               inter: int = 512, vocab: int = 1024, tie: bool = True,
               mrope_section=None, theta: float = 1e6) -> Qwen3Config:
    """Small config with the real head_dim (128) for oracle-speed parity tests."""
    return Qwen3Config(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers,
                       num_attention_heads=heads, num_key_value_heads=kv_heads, head_dim=128,
                       vocab_size=vocab, rope_theta=theta, tie_word_embeddings=tie,
                       mrope_section=mrope_section, eos_token_ids=[])


def tiny_qwen3vl(text_layers: int = 3, depth: int = 3) -> Qwen3VLConfig:
    text = tiny_qwen3(layers=text_layers, hidden=256, heads=4, kv_heads=2, inter=512, vocab=2048,
                      tie=False, mrope_section=[24, 20, 20], theta=5e6)
    vis = Qwen3VLVisionConfig(depth=depth, hidden_size=144, num_heads=2, intermediate_size=320,
                              out_hidden_size=256, num_position_embeddings=64,
                              deepstack_visual_indexes=[0, 1, 2][:depth])
    return Qwen3VLConfig(text=text, vision=vis, image_token_id=2000, video_token_id=2001,
                         vision_start_token_id=2002, vision_end_token_id=2003)


# ---- Qwen3-ASR (reference: Qwen3ASRAudioConfig / ThinkerConfig, /root/reference/src/models/qwen3_asr/config.rs) ----------

@dataclass
class Qwen3ASRAudioConfig:
    d_model: int = 896
    encoder_layers: int = 18
    encoder_attention_heads: int = 14
    encoder_ffn_dim: int = 3584
    num_mel_bins: int = 128
    downsample_hidden_size: int = 480
    output_dim: int = 1024
    n_window: int = 50
    n_window_infer: int = 800
    conv_chunksize: int = 500
    activation_function: str = "gelu"

    @property
    def head_dim(self) -> int:
        return self.d_model // self.encoder_attention_heads


@dataclass
class Qwen3ASRConfig:
    text: Qwen3Config
    audio: Qwen3ASRAudioConfig
    audio_token_id: int = 151676
    audio_start_token_id: int = 151669
    audio_end_token_id: int = 151670


def qwen3_asr_0_6b() -> Qwen3ASRConfig:
    """BASELINE cfg 4.  Audio dims are the published Qwen3-ASR-0.6B values as recalled in SURVEY.md section 8 -- flagged
    [unverified] there (no config.json on disk); text tower = Qwen3-0.6B dims."""
    t = qwen3_0_6b()
    return Qwen3ASRConfig(text=t, audio=Qwen3ASRAudioConfig())


def tiny_qwen3_asr(layers: int = 2) -> Qwen3ASRConfig:
    text = tiny_qwen3(layers=2, hidden=256, heads=4, kv_heads=2, inter=512, vocab=2048, tie=True)
    aud = Qwen3ASRAudioConfig(d_model=128, encoder_layers=layers, encoder_attention_heads=2, encoder_ffn_dim=256,
                              downsample_hidden_size=32, output_dim=256)
    return Qwen3ASRConfig(text=text, audio=aud, audio_token_id=2000, audio_start_token_id=2001, audio_end_token_id=2002)
