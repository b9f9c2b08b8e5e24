"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's Qwen3 decoder stack.

PARITY UNPINNED (see oracle/numerics.py): the reference holds no golden vectors for this path and cannot be built
here.  The op order below follows the cited reference lines one Candle op at a time; each op's output is rounded to
the model dtype by ``Numerics.r``.  An independent cross-check against HF ``transformers`` (same architecture, not the
reference) lives in tests/test_oracle_vs_hf.py.

Reference files restated:
  src/models/qwen3/model.rs:71-87,135-189          decoder layer, model forward (last-position lm_head)
  src/models/common/modules.rs:81-87,530-579,757-813  GateUpDownMLP, QKNormAttention, eager_attention_forward
  src/position_embed/rope.rs:7-22,96-132,583-612   inv_freq, rotate_half, apply_rotary_pos_emb, RoPE::forward
  src/utils/tensor_utils.rs:78-124                 causal mask, repeat_kv
  src/models/common/generate.rs:70-159, sample.rs:7-37   greedy loop (temperature 0 -> ArgMax)
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .numerics import Numerics


def compute_default_rope_parameters(dim: int, base: float) -> torch.Tensor:
    """rope.rs:7-13 -- inv_freq_i = 1 / base^(i/dim) for i in 0,2,..,dim-2, evaluated in f32 (``powf``)."""
    i = np.arange(0, dim, 2, dtype=np.float32)
    expo = (i / np.float32(dim)).astype(np.float32)
    inv = (np.float32(1.0) / np.power(np.float32(base), expo, dtype=np.float32)).astype(np.float32)
    return torch.from_numpy(inv)


def rope_cos_sin(inv_freq: torch.Tensor, seqlen_offset: int, seq_len: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """RoPE::forward, rope.rs:593-612 -- f32 arange positions (x) inv_freq, emb = cat(freqs, freqs), cos/sin in f32."""
    pos = torch.arange(seqlen_offset, seqlen_offset + seq_len, dtype=torch.float32).reshape(seq_len, 1)
    freqs = pos * inv_freq.reshape(1, -1)          # K=1 matmul == exact f32 product
    emb = torch.cat([freqs, freqs], dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    """rope.rs:15-22 -- cat(-x[..., d/2:], x[..., :d/2])."""
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def apply_rotary_pos_emb(nm: Numerics, q, k, cos, sin):
    """rope.rs:96-132 with tof32=false: cos/sin cast to q dtype; q*cos, rot(q)*sin, add -- three materialised ops."""
    if cos.dim() == 2:
        cos, sin = cos[None, None], sin[None, None]
    elif cos.dim() == 3:
        cos, sin = cos[:, None], sin[:, None]
    cos, sin = nm.r(cos), nm.r(sin)

    def rot(x):
        return nm.r(nm.r(x * cos) + nm.r(rotate_half(x) * sin))

    return rot(q), rot(k)


def rms_norm(nm: Numerics, x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """candle_nn::RmsNorm (qwen3/model.rs:53-62,118; modules.rs:512-513): y = x / sqrt(mean(x^2) + eps) * w."""
    x32 = x.float()
    m = torch.sqrt((x32 * x32).sum(-1, keepdim=True) / x.shape[-1] + eps)
    if nm.rmsnorm_in_T:
        m = nm.r(m)
        return nm.r(nm.r(x32 / m) * w)
    return nm.r(x32 / m * w)


def prepare_causal_attention_mask(seq_len: int) -> torch.Tensor:
    """tensor_utils.rs:78-106 with offset 0: f32 (1,1,S,S), -inf strictly above the diagonal."""
    m = torch.zeros(seq_len, seq_len, dtype=torch.float32)
    m.masked_fill_(torch.triu(torch.ones(seq_len, seq_len, dtype=torch.bool), 1), float("-inf"))
    return m[None, None]


def repeat_kv(x: torch.Tensor, n_rep: int) -> torch.Tensor:
    """tensor_utils.rs:108-124 -- cat n_rep copies along dim 2 then reshape => q head i uses kv head i // n_rep."""
    if n_rep == 1:
        return x
    b, h, s, d = x.shape
    return torch.cat([x] * n_rep, dim=2).reshape(b, h * n_rep, s, d)


def attn_scale(nm: Numerics, head_dim: int) -> float:
    """modules.rs:476 scaling = 1/sqrt(head_dim) (f64); ``attn_weights * scaling`` is a Candle affine op whose scalar
    is cast to T before the multiply [unverified] -- so the effective scale is round_T(1/sqrt(d))."""
    s = 1.0 / math.sqrt(head_dim)
    return float(nm.r(torch.tensor(s, dtype=torch.float32)))


def eager_attention_forward(nm: Numerics, q, k, v, n_rep: int, mask, scale: float):
    """modules.rs:757-813 (non-flash branch): matmul, * scaling, + mask, softmax_last_dim, matmul, transpose(1,2).
    ``mask``: None, the (1,1,S,S) tensor of prepare_causal_attention_mask, or the string "causal" = the same mask built
    ``nm.attn_row_block`` query rows at a time (with kv_len == q_len, as the reference's mask assumes, qwen3/model.rs:168-173)."""
    k = repeat_kv(k, n_rep)
    v = repeat_kv(v, n_rep)
    S = q.shape[2]
    blk = nm.attn_row_block if (nm.attn_row_block > 0 and S > nm.attn_row_block) else 0
    if not blk and isinstance(mask, str):
        mask = prepare_causal_attention_mask(S)
    def scores(qq, kt):   # q.k^T -> T, * scaling -> T (modules.rs:782-783); attn_scores_rounded = False: no materialisation in T
        if nm.attn_scores_rounded:
            return nm.r(nm.matmul(qq, kt) * scale)
        return ((qq.double() @ kt.double()).float() if nm.matmul_f64 else qq @ kt) * scale

    def rr(x):            # the mask add's materialisation
        return nm.r(x) if nm.attn_scores_rounded else x

    if not blk:
        w = scores(q, k.transpose(-2, -1))
        if mask is not None:
            w = rr(w + mask)
        p = torch.softmax(w, dim=-1)
        if nm.attn_probs_rounded:
            p = nm.r(p)
        o = nm.matmul(p, v)
        return o.transpose(1, 2).contiguous()
    # row-blocked: rows [a, b) of the same computation.  Under the causal mask the columns >= b are -inf for every row of the
    # block (exp -> exactly 0, products with V exactly 0), so they are left out: the sums are the full computation's sums.
    causal = isinstance(mask, str)
    assert mask is None or causal, "row blocking takes the causal mask by name"
    kt = k.transpose(-2, -1)
    outs = []
    for a in range(0, S, blk):
        b = min(S, a + blk)
        kv_hi = b if causal else k.shape[2]
        w = scores(q[:, :, a:b], kt[..., :kv_hi])
        if causal:
            rows = torch.arange(a, b).reshape(-1, 1)
            cols = torch.arange(0, kv_hi).reshape(1, -1)
            w = rr(w + torch.where(cols > rows, float("-inf"), 0.0)[None, None])
        p = torch.softmax(w, dim=-1)
        if nm.attn_probs_rounded:
            p = nm.r(p)
        outs.append(nm.matmul(p, v[:, :, :kv_hi]))
    return torch.cat(outs, 2).transpose(1, 2).contiguous()


def silu(x: torch.Tensor) -> torch.Tensor:
    """candle Activation::Silu = x / (1 + exp(-x))."""
    return x / (1.0 + torch.exp(-x))


class OracleQwen3:
    """Qwen3Model (qwen3/model.rs:94-214) with the per-layer concat KV cache of modules.rs:558-566."""

    def __init__(self, cfg, weights: Dict[str, torch.Tensor], nm: Optional[Numerics] = None,
                 prefix: Optional[str] = None, lm_head_name: str = "lm_head.weight", consume: bool = False):
        self.cfg = cfg
        self.nm = nm or Numerics()
        if prefix is None:  # model.rs:105-109: optional "model." prefix
            prefix = "model." if "model.embed_tokens.weight" in weights else ""
        self.p = prefix
        # weights are stored in T already (bf16 checkpoints); hold them as f32 values of T-representable numbers
        # (consume: entries are popped from `weights` as they are converted -- an 8B checkpoint is 17 GB in bf16 + 35 GB as f32)
        self.w = {}
        for k in [k for k in weights if k.startswith(prefix) or k == lm_head_name]:
            self.w[k] = self.nm.r((weights.pop(k) if consume else weights[k]).float())
        self.embed = self.w[prefix + "embed_tokens.weight"]
        self.lm_head = self.embed if cfg.tie_word_embeddings else self.w[lm_head_name]
        self.inv_freq = compute_default_rope_parameters(cfg.head_dim, cfg.rope_theta)
        self.scale = attn_scale(self.nm, cfg.head_dim)
        self.kv: List[Optional[Tuple[torch.Tensor, torch.Tensor]]] = [None] * cfg.num_hidden_layers

    # -- InferenceModel (common/mod.rs:25-45) --
    def clear_cache(self):
        self.kv = [None] * self.cfg.num_hidden_layers

    def stop_token_ids(self):
        return list(self.cfg.eos_token_ids)

    def forward_step(self, input_ids, seqlen_offset: int) -> torch.Tensor:
        return self.forward(input_ids, seqlen_offset)

    forward_initial = forward_step

    # -- layers --
    def _attn(self, li: int, x, cos, sin, mask):
        c, nm, p = self.cfg, self.nm, f"{self.p}layers.{li}.self_attn."
        b, s, _ = x.shape
        q = nm.linear(x, self.w[p + "q_proj.weight"]).reshape(b, s, c.num_attention_heads, c.head_dim)
        q = rms_norm(nm, q, self.w[p + "q_norm.weight"], c.rms_norm_eps).transpose(1, 2)
        k = nm.linear(x, self.w[p + "k_proj.weight"]).reshape(b, s, c.num_key_value_heads, c.head_dim)
        k = rms_norm(nm, k, self.w[p + "k_norm.weight"], c.rms_norm_eps).transpose(1, 2)
        v = nm.linear(x, self.w[p + "v_proj.weight"]).reshape(b, s, c.num_key_value_heads, c.head_dim).transpose(1, 2)
        q, k = apply_rotary_pos_emb(nm, q, k, cos, sin)
        if self.kv[li] is not None:
            pk, pv = self.kv[li]
            k = torch.cat([pk, k], dim=2)
            v = torch.cat([pv, v], dim=2)
        self.kv[li] = (k, v)
        o = eager_attention_forward(nm, q, k, v, c.num_attention_heads // c.num_key_value_heads, mask, self.scale)
        o = o.reshape(b, s, c.num_attention_heads * c.head_dim)
        return nm.linear(o, self.w[p + "o_proj.weight"])

    def _mlp(self, li: int, x):
        nm, p = self.nm, f"{self.p}layers.{li}.mlp."
        lhs = nm.r(silu(nm.linear(x, self.w[p + "gate_proj.weight"])))
        rhs = nm.linear(x, self.w[p + "up_proj.weight"])
        return nm.linear(nm.r(lhs * rhs), self.w[p + "down_proj.weight"])

    def decoder_layer(self, li: int, x, cos, sin, mask):
        """Qwen3DecoderLayer::forward, qwen3/model.rs:71-87."""
        nm, c, p = self.nm, self.cfg, f"{self.p}layers.{li}."
        r = x
        h = rms_norm(nm, x, self.w[p + "input_layernorm.weight"], c.rms_norm_eps)
        x = nm.r(r + self._attn(li, h, cos, sin, mask))
        r = x
        h = rms_norm(nm, x, self.w[p + "post_attention_layernorm.weight"], c.rms_norm_eps)
        return nm.r(r + self._mlp(li, h))

    def decoder_layer_rows(self, li: int, x, cos, sin, rows):
        """decoder_layer (qwen3/model.rs:71-87) evaluated for the query rows ``rows`` of a prefill ONLY: K / V are computed for every
        row (they are what the selected rows attend to), q / attention / o_proj / MLP for the selected rows.  Row r of a causal
        prefill depends on rows <= r alone, so for a layer whose INPUT x is known for all rows these are exactly the rows
        decoder_layer would produce (same ops, same rounding points; masked columns contribute exact zeros and are left out).
        Makes one layer at S ~ 41k checkable on the host in seconds.  x (1,S,H); cos/sin (S,d) or (1,S,d); -> (len(rows), H)."""
        c, nm, p = self.cfg, self.nm, f"{self.p}layers.{li}."
        pa = p + "self_attn."
        rows = list(rows)
        S = x.shape[1]
        cos2, sin2 = cos.reshape(S, -1), sin.reshape(S, -1)
        h = rms_norm(nm, x, self.w[p + "input_layernorm.weight"], c.rms_norm_eps)
        k = nm.linear(h, self.w[pa + "k_proj.weight"]).reshape(1, S, c.num_key_value_heads, c.head_dim)
        k = rms_norm(nm, k, self.w[pa + "k_norm.weight"], c.rms_norm_eps).transpose(1, 2)
        v = nm.linear(h, self.w[pa + "v_proj.weight"]).reshape(1, S, c.num_key_value_heads, c.head_dim).transpose(1, 2)
        hq = h[:, rows]
        q = nm.linear(hq, self.w[pa + "q_proj.weight"]).reshape(1, len(rows), c.num_attention_heads, c.head_dim)
        q = rms_norm(nm, q, self.w[pa + "q_norm.weight"], c.rms_norm_eps).transpose(1, 2)
        _, k = apply_rotary_pos_emb(nm, k, k, cos2, sin2)
        q, _ = apply_rotary_pos_emb(nm, q, q, cos2[rows], sin2[rows])
        self.kv[li] = (k, v)            # the cache a decode step continues from (modules.rs:558-566)
        g = c.num_attention_heads // c.num_key_value_heads
        outs = []
        for i, r in enumerate(rows):   # one query row over its r + 1 visible keys
            o = eager_attention_forward(nm, q[:, :, i:i + 1], k[:, :, :r + 1], v[:, :, :r + 1], g, None, self.scale)
            outs.append(o.reshape(1, 1, c.num_attention_heads * c.head_dim))
        o = nm.linear(torch.cat(outs, 1), self.w[pa + "o_proj.weight"])
        xr = nm.r(x[:, rows] + o)
        h2 = rms_norm(nm, xr, self.w[p + "post_attention_layernorm.weight"], c.rms_norm_eps)
        return nm.r(xr + self._mlp(li, h2))[0]

    def embed_tokens(self, input_ids) -> torch.Tensor:
        ids = torch.as_tensor(np.asarray(input_ids, dtype=np.int64)).reshape(1, -1)
        return self.embed[ids]

    def forward_hidden(self, input_ids=None, inputs_embeds=None, seqlen_offset: int = 0, all_positions=False,
                       cos_sin=None, after_layer=None):
        """Qwen3Model::forward_hidden, qwen3/model.rs:146-189.  ``cos_sin`` / ``after_layer`` let the Qwen3-VL text
        model (qwen3vl/model.rs:775-828: M-RoPE tables, DeepStack adds) reuse the same layer code."""
        x = inputs_embeds if inputs_embeds is not None else self.embed_tokens(input_ids)
        s = x.shape[1]
        if s <= 1:
            mask = None
        elif self.nm.attn_row_block > 0 and s > self.nm.attn_row_block:
            mask = "causal"                                           # the same mask, built per row block (eager_attention_forward)
        else:
            mask = prepare_causal_attention_mask(s)                   # offset hard-coded 0 (model.rs:168-173)
        cos, sin = cos_sin if cos_sin is not None else rope_cos_sin(self.inv_freq, seqlen_offset, s)
        for li in range(self.cfg.num_hidden_layers):
            x = self.decoder_layer(li, x, cos, sin, mask)
            if after_layer is not None:
                x = after_layer(li, x)
        x = rms_norm(self.nm, x, self.w[self.p + "norm.weight"], self.cfg.rms_norm_eps)
        return x if all_positions else x[:, s - 1:s, :]

    def forward(self, input_ids, seqlen_offset: int = 0) -> torch.Tensor:
        """-> logits (1,1,V) in T (returned as f32 values), qwen3/model.rs:135-145."""
        h = self.forward_hidden(input_ids, None, seqlen_offset)
        return self.nm.linear(h, self.lm_head)


def l2_normalize(t: torch.Tensor) -> torch.Tensor:
    """common/modules.rs:1287-1294: t / sqrt(sum(t^2, last dim) + 1e-6), in the tensor's dtype (f32 at the call site)."""
    return t / torch.sqrt((t * t).sum(-1, keepdim=True) + 1e-6)


def embed_one(model: "OracleQwen3", input_ids) -> torch.Tensor:
    """Qwen3Embedding::embed_one (qwen3_embedding/mod.rs:50-64): forward_hidden(ids, None, 0).squeeze(0) -> f32 ->
    clear_kv_cache -> NormalizeType::L2 over the last dim -> squeeze(0) => (hidden,) f32."""
    model.clear_cache()
    hidden = model.forward_hidden(input_ids, None, 0).squeeze(0).float()
    model.clear_cache()
    return l2_normalize(hidden).squeeze(0)


def rerank(model: "OracleQwen3", query_ids, documents_ids) -> torch.Tensor:
    """Qwen3Reranker::rerank (qwen3_reranker/mod.rs:23-31) with cosine_similarity_no_l2 (modules.rs:1381-1389)."""
    q = embed_one(model, query_ids).unsqueeze(0)
    d = torch.stack([embed_one(model, x) for x in documents_ids], 0)
    return (q @ d.transpose(-1, -2)).squeeze(0)


def greedy_generate(model, input_ids, max_tokens: int, mm=None, return_logits: bool = False):
    """generate_generic with temperature 0 (common/generate.rs:115-159; sample.rs:13-37 -> Sampling::ArgMax):
    prefill, argmax (first maximal index), then up to max_tokens-1 single-token steps; stop after pushing an eos id."""
    eos = set(model.stop_token_ids())
    generated: List[int] = []
    all_logits = []
    seqlen_offset, seq_len = 0, len(input_ids)
    logits = model.forward_initial(input_ids, seqlen_offset) if mm is None else model.forward_initial(input_ids, seqlen_offset, mm)
    lg = logits.reshape(-1).float()
    all_logits.append(lg)
    tok = int(torch.argmax(lg))
    generated.append(tok)
    for _ in range(1, max_tokens):
        seqlen_offset += seq_len
        seq_len = 1
        logits = model.forward_step([tok], seqlen_offset)
        lg = logits.reshape(-1).float()
        all_logits.append(lg)
        tok = int(torch.argmax(lg))
        generated.append(tok)
        if tok in eos:
            break
    model.clear_cache()
    return (generated, all_logits) if return_logits else generated
