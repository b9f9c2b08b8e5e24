"""Checkpoint directories: the on-disk layout ``XxxGenerateModel::init(path, ..)`` reads
(/root/reference/src/models/qwen3/generate.rs:22-50, qwen3vl/generate.rs:33-63, qwen3_asr/generate.rs:51-87):
``config.json`` + ``generation_config.json`` + every ``*.safetensors`` file of the directory.

* ``parse_config`` / ``open_weights`` / ``HipInferenceModel.from_pretrained`` go through the C ABI's native loader
  (csrc/loader.hip: JSON reader, safetensors mmap) -- the same code a Rust caller would use.
* ``save_checkpoint`` WRITES such a directory from a config record + tensors (HF field names, the nesting the reference's
  serde structs expect).  No checkpoint can be downloaded here, so tests and examples make their own.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import ModelDesc, TensorView, check, lib
from .configs import Qwen3ASRConfig, Qwen3Config, Qwen3VLConfig


def _text_dict(t: Qwen3Config, dtype_key: str) -> dict:
    return {
        "attention_bias": t.attention_bias, "attention_dropout": 0.0, "bos_token_id": 151643, "eos_token_id": 151645,
        "head_dim": t.head_dim, "hidden_act": t.hidden_act, "hidden_size": t.hidden_size, "initializer_range": 0.02,
        "intermediate_size": t.intermediate_size, "max_position_embeddings": 40960, "max_window_layers": t.num_hidden_layers,
        "num_attention_heads": t.num_attention_heads, "num_hidden_layers": t.num_hidden_layers,
        "num_key_value_heads": t.num_key_value_heads, "rms_norm_eps": t.rms_norm_eps, "rope_theta": t.rope_theta,
        "tie_word_embeddings": t.tie_word_embeddings, dtype_key: "bfloat16", "use_cache": True,
        "use_sliding_window": False, "vocab_size": t.vocab_size,
    }


def config_json(cfg) -> dict:
    """The config.json the reference's serde structs deserialise (field names: qwen3/config.rs:4-27,
    qwen3vl/config.rs:51-133, qwen3_asr/config.rs:6-22)."""
    if isinstance(cfg, Qwen3VLConfig):
        t = _text_dict(cfg.text, "dtype")
        t["rope_scaling"] = {"rope_type": "default", "mrope_section": list(cfg.text.mrope_section), "mrope_interleaved": True}
        v = cfg.vision
        return {
            "architectures": ["Qwen3VLForConditionalGeneration"], "model_type": "qwen3_vl",
            "image_token_id": cfg.image_token_id, "video_token_id": cfg.video_token_id,
            "vision_start_token_id": cfg.vision_start_token_id, "vision_end_token_id": cfg.vision_end_token_id,
            "tie_word_embeddings": cfg.tie_word_embeddings, "text_config": t,
            "vision_config": {
                "deepstack_visual_indexes": list(v.deepstack_visual_indexes), "depth": v.depth, "hidden_act": v.hidden_act,
                "hidden_size": v.hidden_size, "in_channels": v.in_channels, "initializer_range": 0.02,
                "intermediate_size": v.intermediate_size, "num_heads": v.num_heads,
                "num_position_embeddings": v.num_position_embeddings, "out_hidden_size": v.out_hidden_size,
                "patch_size": v.patch_size, "spatial_merge_size": v.spatial_merge_size,
                "temporal_patch_size": v.temporal_patch_size,
            },
        }
    if isinstance(cfg, Qwen3ASRConfig):
        t = _text_dict(cfg.text, "dtype")
        t["rope_scaling"] = {"interleaved": True, "mrope_interleaved": True, "mrope_section": [24, 20, 20],
                             "rope_type": "default", "type": "default"}
        a = cfg.audio
        return {
            "model_type": "qwen3_asr", "support_languages": [],
            "thinker_config": {
                "model_type": "qwen3_asr_thinker", "audio_token_id": cfg.audio_token_id,
                "audio_start_token_id": cfg.audio_start_token_id, "audio_end_token_id": cfg.audio_end_token_id,
                "dtype": "bfloat16", "initializer_range": 0.02, "text_config": t,
                "audio_config": {
                    "d_model": a.d_model, "encoder_layers": a.encoder_layers,
                    "encoder_attention_heads": a.encoder_attention_heads, "encoder_ffn_dim": a.encoder_ffn_dim,
                    "num_mel_bins": a.num_mel_bins, "downsample_hidden_size": a.downsample_hidden_size,
                    "output_dim": a.output_dim, "n_window": a.n_window, "n_window_infer": a.n_window_infer,
                    "conv_chunksize": a.conv_chunksize, "max_source_positions": 1500,
                    "activation_function": "gelu", "model_type": "qwen3_asr_audio_encoder",
                },
            },
        }
    d = _text_dict(cfg, "torch_dtype")
    d["architectures"] = ["Qwen3ForCausalLM"]
    d["model_type"] = "qwen3"
    return d


def generation_config_json(cfg) -> dict:
    t = cfg.text if isinstance(cfg, (Qwen3VLConfig, Qwen3ASRConfig)) else cfg
    return {"bos_token_id": 151643, "pad_token_id": 151643, "do_sample": True, "eos_token_id": list(t.eos_token_ids),
            "top_p": 0.95, "top_k": 20, "temperature": 0.6}


def save_checkpoint(path: str, cfg, weights: Dict[str, torch.Tensor], shards: int = 1) -> List[str]:
    """Write config.json, generation_config.json and `shards` safetensors files (tensors dealt round-robin, the way a
    sharded HF checkpoint splits them arbitrarily across files)."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(config_json(cfg), f, indent=2)
    with open(os.path.join(path, "generation_config.json"), "w") as f:
        json.dump(generation_config_json(cfg), f, indent=2)
    names = list(weights)
    files = []
    for s in range(shards):
        part = {n: weights[n].detach().cpu().contiguous() for n in names[s::shards]}
        fn = os.path.join(path, f"model-{s + 1:05d}-of-{shards:05d}.safetensors")
        save_file(part, fn, metadata={"format": "pt"})
        files.append(fn)
    return files


def parse_config(path: str) -> ModelDesc:
    """<path>/config.json (+ generation_config.json) -> aha_model_desc, by the native parser (host only)."""
    d = ModelDesc()
    check(lib().aha_hip_config_parse(os.fsencode(path), C.byref(d)))
    return d


def open_weights(path: str) -> Dict[str, Tuple[int, Tuple[int, ...], bytes]]:
    """Every tensor of every *.safetensors file in `path` through the native mmap reader: name -> (aha_dtype or -1,
    shape, raw little-endian bytes).  Host only; copies the bytes out before closing the mappings."""
    h = C.c_void_p()
    check(lib().aha_hip_weights_open(os.fsencode(path), C.byref(h)))
    out = {}
    try:
        for i in range(lib().aha_hip_weights_count(h)):
            v = TensorView()
            check(lib().aha_hip_weights_get(h, i, C.byref(v)))
            shape = tuple(int(v.shape[j]) for j in range(v.ndim))
            elem = {_lib.AHA_BF16: 2, _lib.AHA_F16: 2, _lib.AHA_F32: 4, _lib.AHA_U32: 4, _lib.AHA_U8: 1}.get(v.dtype)
            raw = b"" if elem is None else C.string_at(v.data, int(np.prod(shape, dtype=np.int64)) * elem)
            out[v.name.decode()] = (int(v.dtype), shape, raw)
    finally:
        lib().aha_hip_weights_close(h)
    return out


def config_from_desc(d: ModelDesc):
    """aha_model_desc (as parsed from a checkpoint directory) -> the Python config record of aha_amd.configs."""
    from .configs import Qwen3ASRAudioConfig, Qwen3VLVisionConfig
    ms = [int(d.mrope_section[i]) for i in range(3)]
    t = Qwen3Config(hidden_size=d.hidden_size, intermediate_size=d.intermediate_size, num_hidden_layers=d.num_hidden_layers,
                    num_attention_heads=d.num_attention_heads, num_key_value_heads=d.num_key_value_heads,
                    head_dim=d.head_dim, vocab_size=d.vocab_size, rms_norm_eps=float(d.rms_norm_eps),
                    rope_theta=float(d.rope_theta), tie_word_embeddings=bool(d.tie_word_embeddings),
                    mrope_section=ms if any(ms) else None,
                    eos_token_ids=[int(d.stop_tokens[i]) for i in range(d.n_stop_tokens)])
    if d.arch == _lib.AHA_ARCH_QWEN3VL:
        v = Qwen3VLVisionConfig(depth=d.vis_depth, hidden_size=d.vis_hidden_size, num_heads=d.vis_num_heads,
                                intermediate_size=d.vis_intermediate_size, in_channels=d.vis_in_channels,
                                patch_size=d.vis_patch_size, temporal_patch_size=d.vis_temporal_patch_size,
                                spatial_merge_size=d.vis_spatial_merge_size, out_hidden_size=d.vis_out_hidden_size,
                                num_position_embeddings=d.vis_num_position_embeddings,
                                deepstack_visual_indexes=[int(d.vis_deepstack_indexes[i]) for i in range(d.vis_num_deepstack)])
        return Qwen3VLConfig(text=t, vision=v, image_token_id=d.image_token_id, video_token_id=d.video_token_id,
                             vision_start_token_id=d.vision_start_token_id, vision_end_token_id=d.vision_end_token_id,
                             tie_word_embeddings=bool(d.tie_word_embeddings))
    if d.arch == _lib.AHA_ARCH_QWEN3ASR:
        a = Qwen3ASRAudioConfig(d_model=d.aud_d_model, encoder_layers=d.aud_encoder_layers,
                                encoder_attention_heads=d.aud_attention_heads, encoder_ffn_dim=d.aud_ffn_dim,
                                num_mel_bins=d.aud_num_mel_bins, downsample_hidden_size=d.aud_downsample_hidden_size,
                                output_dim=d.aud_output_dim, n_window=d.aud_n_window)
        return Qwen3ASRConfig(text=t, audio=a, audio_token_id=d.audio_token_id)
    return t
