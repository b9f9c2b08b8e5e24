"""Synthetic checkpoints with the exact tensor names the reference looks up.

Names follow SURVEY.md Appendix C:
  * Qwen3      -- /root/reference/src/models/qwen3/model.rs:104-134, common/modules.rs:482-513,66-71
  * Qwen3-VL   -- /root/reference/src/models/qwen3vl/model.rs:847-870 (visual.*, language_model.*, lm_head)
Distribution is BASELINE.md section 4: N(0, 0.02^2) matrices/embeddings/biases, norm weights 1 + N(0, 0.02^2),
stored bf16.  There is no network, hence no real checkpoint; shapes are those of the named architecture.
"""
from __future__ import annotations

from typing import Dict

import torch

from .configs import Qwen3Config, Qwen3VLConfig


def _randn(gen, shape, std=0.02, mean=0.0, dtype=torch.bfloat16):
    # gen.device decides where the tensor is made: CPU generators give the reproducible fixtures the oracle shares;
    # a CUDA generator lets bench.py build a full-size checkpoint in HBM in seconds (passed to the ABI on_device).
    t = torch.empty(shape, dtype=torch.float32, device=gen.device)
    t.normal_(mean, std, generator=gen)
    return t.to(dtype)


def _gen(seed: int, device="cpu") -> torch.Generator:
    return torch.Generator(device=device).manual_seed(seed)


def qwen3_text_weights(cfg: Qwen3Config, seed: int = 0, prefix: str = "model.", device="cpu",
                       lm_head_name: str = "lm_head.weight", dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    gen = _gen(seed, device)
    w: Dict[str, torch.Tensor] = {}
    H, I = cfg.hidden_size, cfg.intermediate_size
    w[f"{prefix}embed_tokens.weight"] = _randn(gen, (cfg.vocab_size, H), dtype=dtype)
    for i in range(cfg.num_hidden_layers):
        p = f"{prefix}layers.{i}."
        w[p + "input_layernorm.weight"] = _randn(gen, (H,), mean=1.0, dtype=dtype)
        w[p + "self_attn.q_proj.weight"] = _randn(gen, (cfg.q_dim, H), dtype=dtype)
        w[p + "self_attn.k_proj.weight"] = _randn(gen, (cfg.kv_dim, H), dtype=dtype)
        w[p + "self_attn.v_proj.weight"] = _randn(gen, (cfg.kv_dim, H), dtype=dtype)
        w[p + "self_attn.o_proj.weight"] = _randn(gen, (H, cfg.q_dim), dtype=dtype)
        w[p + "self_attn.q_norm.weight"] = _randn(gen, (cfg.head_dim,), mean=1.0, dtype=dtype)
        w[p + "self_attn.k_norm.weight"] = _randn(gen, (cfg.head_dim,), mean=1.0, dtype=dtype)
        w[p + "post_attention_layernorm.weight"] = _randn(gen, (H,), mean=1.0, dtype=dtype)
        w[p + "mlp.gate_proj.weight"] = _randn(gen, (I, H), dtype=dtype)
        w[p + "mlp.up_proj.weight"] = _randn(gen, (I, H), dtype=dtype)
        w[p + "mlp.down_proj.weight"] = _randn(gen, (H, I), dtype=dtype)
    w[f"{prefix}norm.weight"] = _randn(gen, (H,), mean=1.0, dtype=dtype)
    if not cfg.tie_word_embeddings:
        w[lm_head_name] = _randn(gen, (cfg.vocab_size, H), dtype=dtype)
    return w


def qwen3vl_vision_weights(cfg: Qwen3VLConfig, seed: int = 100, prefix: str = "model.visual.", device="cpu",
                           dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    v = cfg.vision
    gen = _gen(seed, device)
    w: Dict[str, torch.Tensor] = {}
    D = v.hidden_size
    w[prefix + "patch_embed.proj.weight"] = _randn(
        gen, (D, v.in_channels, v.temporal_patch_size, v.patch_size, v.patch_size), dtype=dtype)
    w[prefix + "patch_embed.proj.bias"] = _randn(gen, (D,), dtype=dtype)
    w[prefix + "pos_embed.weight"] = _randn(gen, (v.num_position_embeddings, D), dtype=dtype)
    for i in range(v.depth):
        p = f"{prefix}blocks.{i}."
        for n in ("norm1", "norm2"):
            w[p + n + ".weight"] = _randn(gen, (D,), mean=1.0, dtype=dtype)
            w[p + n + ".bias"] = _randn(gen, (D,), dtype=dtype)
        w[p + "attn.qkv.weight"] = _randn(gen, (3 * D, D), dtype=dtype)
        w[p + "attn.qkv.bias"] = _randn(gen, (3 * D,), dtype=dtype)
        w[p + "attn.proj.weight"] = _randn(gen, (D, D), dtype=dtype)
        w[p + "attn.proj.bias"] = _randn(gen, (D,), dtype=dtype)
        w[p + "mlp.linear_fc1.weight"] = _randn(gen, (v.intermediate_size, D), dtype=dtype)
        w[p + "mlp.linear_fc1.bias"] = _randn(gen, (v.intermediate_size,), dtype=dtype)
        w[p + "mlp.linear_fc2.weight"] = _randn(gen, (D, v.intermediate_size), dtype=dtype)
        w[p + "mlp.linear_fc2.bias"] = _randn(gen, (D,), dtype=dtype)
    M = D * v.spatial_merge_size ** 2

    def merger(p, postshuffle):
        nd = M if postshuffle else D
        w[p + "norm.weight"] = _randn(gen, (nd,), mean=1.0, dtype=dtype)
        w[p + "norm.bias"] = _randn(gen, (nd,), dtype=dtype)
        w[p + "linear_fc1.weight"] = _randn(gen, (M, M), dtype=dtype)
        w[p + "linear_fc1.bias"] = _randn(gen, (M,), dtype=dtype)
        w[p + "linear_fc2.weight"] = _randn(gen, (v.out_hidden_size, M), dtype=dtype)
        w[p + "linear_fc2.bias"] = _randn(gen, (v.out_hidden_size,), dtype=dtype)

    merger(prefix + "merger.", False)
    for k in range(len(v.deepstack_visual_indexes)):
        merger(f"{prefix}deepstack_merger_list.{k}.", True)
    return w


def qwen3vl_weights(cfg: Qwen3VLConfig, seed: int = 0, dtype=torch.bfloat16, device="cpu") -> Dict[str, torch.Tensor]:
    w = qwen3_text_weights(cfg.text, seed=seed, prefix="model.language_model.",
                           lm_head_name="lm_head.weight", dtype=dtype, device=device)
    w.update(qwen3vl_vision_weights(cfg, seed=seed + 100, dtype=dtype, device=device))
    return w


def qwen3_asr_weights(cfg, seed: int = 0, dtype=torch.bfloat16, device="cpu") -> Dict[str, torch.Tensor]:
    """Names the reference looks up for Qwen3-ASR (qwen3_asr/model.rs:42-70,102-169,238-256,316-334)."""
    w = qwen3_text_weights(cfg.text, seed=seed, prefix="thinker.model.", lm_head_name="thinker.lm_head.weight",
                           dtype=dtype, device=device)
    a = cfg.audio
    gen = _gen(seed + 200, device)
    p = "thinker.audio_tower."
    D, H = a.d_model, a.downsample_hidden_size
    w[p + "conv2d1.weight"] = _randn(gen, (H, 1, 3, 3), std=0.2, dtype=dtype)
    w[p + "conv2d1.bias"] = _randn(gen, (H,), dtype=dtype)
    for n in ("conv2d2", "conv2d3"):
        w[p + n + ".weight"] = _randn(gen, (H, H, 3, 3), dtype=dtype)
        w[p + n + ".bias"] = _randn(gen, (H,), dtype=dtype)
    fdim = H * ((((a.num_mel_bins + 1) // 2 + 1) // 2 + 1) // 2)
    w[p + "conv_out.weight"] = _randn(gen, (D, fdim), dtype=dtype)
    for i in range(a.encoder_layers):
        q = f"{p}layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            w[q + f"self_attn.{n}.weight"] = _randn(gen, (D, D), dtype=dtype)
            w[q + f"self_attn.{n}.bias"] = _randn(gen, (D,), dtype=dtype)
        for n in ("self_attn_layer_norm", "final_layer_norm"):
            w[q + n + ".weight"] = _randn(gen, (D,), mean=1.0, dtype=dtype)
            w[q + n + ".bias"] = _randn(gen, (D,), dtype=dtype)
        w[q + "fc1.weight"] = _randn(gen, (a.encoder_ffn_dim, D), dtype=dtype)
        w[q + "fc1.bias"] = _randn(gen, (a.encoder_ffn_dim,), dtype=dtype)
        w[q + "fc2.weight"] = _randn(gen, (D, a.encoder_ffn_dim), dtype=dtype)
        w[q + "fc2.bias"] = _randn(gen, (D,), dtype=dtype)
    w[p + "ln_post.weight"] = _randn(gen, (D,), mean=1.0, dtype=dtype)
    w[p + "ln_post.bias"] = _randn(gen, (D,), dtype=dtype)
    w[p + "proj1.weight"] = _randn(gen, (D, D), dtype=dtype)
    w[p + "proj1.bias"] = _randn(gen, (D,), dtype=dtype)
    w[p + "proj2.weight"] = _randn(gen, (a.output_dim, D), dtype=dtype)
    w[p + "proj2.bias"] = _randn(gen, (a.output_dim,), dtype=dtype)
    return w
