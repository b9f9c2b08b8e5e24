"""Host mirror of the reference's sampling step (D11) on top of the device candidate extraction.

Reference: `GenerationContext` / `sample_and_push` (src/models/common/generate.rs:21-86), `get_logit_processor` and
`use_repeat_penalty` (src/models/common/sample.rs:7-60), candle_transformers::generation::LogitsProcessor (0.9.2).

Split of the work: the device (aha_hip_sample_candidates) applies the repeat penalty and returns the k largest logits plus
the full-vocabulary softmax normaliser; this module turns them into the weight vector candle would draw from (top-k,
top-k-then-top-p, top-p); the draw itself is the library's restatement of candle's RNG behind the C ABI (aha_hip_rng_*:
rand 0.9.2 `StdRng::seed_from_u64` = PCG32-expanded ChaCha12, `WeightedIndex<f32>`; csrc/sampler_rng.hip), so a seed defines a
token sequence.  [unverified] against the crates (not on disk): for Sampling::All / TopP the sequence is fully defined by
that restatement; for TopK / TopKThenTopP candle draws over the k probabilities in the order `select_nth_unstable_by` leaves
them, which Rust does not specify -- here they are ranked (probability desc, logit desc, index asc).  Greedy requests never come
here (forward_* already returns the arg-max token).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import ctypes as C

import numpy as np

from ._lib import check, lib

MAX_CANDIDATES = 64  # aha_hip_sample_candidates limit


class StdRng:
    """aha_hip_rng_*: rand 0.9.2 `StdRng::seed_from_u64(seed)` (candle_transformers LogitsProcessor::from_sampling / ::new)."""

    def __init__(self, seed: int):
        self._h = C.c_void_p()
        check(lib().aha_hip_rng_create(C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), C.byref(self._h)))

    def next_u32(self) -> int:
        return int(lib().aha_hip_rng_next_u32(self._h))

    def weighted_index(self, weights: np.ndarray) -> int:
        """WeightedIndex::<f32>::new(weights)?.sample(rng) -- one next_u32."""
        w = np.ascontiguousarray(weights, dtype=np.float32)
        out = C.c_uint32()
        check(lib().aha_hip_rng_weighted_index(self._h, w.ctypes.data_as(C.POINTER(C.c_float)), w.size, C.byref(out)))
        return int(out.value)

    def __del__(self):
        try:
            if self._h:
                lib().aha_hip_rng_destroy(self._h)
                self._h = None
        except Exception:
            pass


@dataclass
class Sampling:
    kind: str                   # "ArgMax" | "All" | "TopK" | "TopP" | "TopKThenTopP"  (candle Sampling)
    temperature: float = 1.0
    k: int = 0
    p: float = 1.0


def get_logit_processor(temperature: Optional[float], top_p: Optional[float], top_k: Optional[int], seed: int) -> "LogitsProcessor":
    """sample.rs:7-38 (temperature < 1e-7 => greedy; top_k None => LogitsProcessor::new's ArgMax / All / TopP)."""
    if temperature is not None and np.float32(temperature) < np.float32(1e-7):
        temperature = None
    t = None if temperature is None else float(np.float32(temperature))
    p = None if top_p is None else float(np.float32(top_p))
    if t is None:
        s = Sampling("ArgMax")
    elif top_k is None:
        s = Sampling("All", t) if p is None else Sampling("TopP", t, p=p)
    else:
        s = Sampling("TopK", t, k=int(top_k)) if p is None else Sampling("TopKThenTopP", t, k=int(top_k), p=p)
    return LogitsProcessor(seed, s)


def _topp_mask(prs: np.ndarray, top_p: float, tie_key: Optional[np.ndarray] = None) -> np.ndarray:
    """LogitsProcessor::sample_topp: descending walk (candle's `sort_by` is stable: equal probabilities are walked in position
    order; `tie_key` supplies the vocabulary positions when `prs` is a candidate list in another order), zero everything
    after the running sum has reached top_p."""
    prs = prs.astype(np.float32).copy()
    cumsum = np.float32(0.0)
    order = np.argsort(-prs, kind="stable") if tie_key is None else np.lexsort((tie_key, -prs.astype(np.float64)))
    for i in order:
        if cumsum >= np.float32(top_p):
            prs[i] = 0.0
        else:
            cumsum = np.float32(cumsum + prs[i])
    return prs


class LogitsProcessor:
    def __init__(self, seed: int, sampling: Sampling):
        self.sampling = sampling
        self.rng = StdRng(seed)

    # -- deterministic part ------------------------------------------------------------------------------------------
    def candidates_needed(self, vocab_size: int) -> int:
        """How many candidates to ask the device for; 0 = this sampler needs the full logits vector (or none at all)."""
        s = self.sampling
        if s.kind in ("TopK", "TopKThenTopP"):
            return s.k if 1 <= s.k <= MAX_CANDIDATES and s.k < vocab_size else 0
        if s.kind == "TopP":
            # candle: `if p <= 0.0 || p >= 1.0 { sample_multinomial(&prs) }` -- the whole distribution, no candidates
            return 0 if (s.p <= 0.0 or s.p >= 1.0) else min(MAX_CANDIDATES, vocab_size)
        return 0

    def weights_from_candidates(self, vals: np.ndarray, mx: float, sumexp: float, idx: Optional[np.ndarray] = None) -> Optional[np.ndarray]:
        """Weights over the candidates (same order) that candle's sampler would hand to the weighted draw, or None when the
        candidates do not cover the sampler's support (TopP nucleus wider than the candidate list).  The candidates arrive
        ranked by (logit descending, index ascending), which refines candle's ranking by probability (oracle.topk_order
        states the tie rule); `idx` (their vocabulary positions) lets the plain TopP walk break equal probabilities by
        position as candle's stable full-vocabulary sort does."""
        s = self.sampling
        inv_t = np.float32(1.0 / s.temperature)
        prs = (np.exp((vals.astype(np.float32) - np.float32(mx)) * inv_t, dtype=np.float32) / np.float32(sumexp)).astype(np.float32)
        if s.kind == "TopK":
            return prs
        if s.kind == "TopKThenTopP":
            sum_p = prs.sum(dtype=np.float32)
            return prs if (s.p <= 0.0 or s.p >= sum_p) else _topp_mask(prs, s.p)
        if s.kind == "TopP":
            if prs.sum(dtype=np.float32) < np.float32(s.p):
                return None  # the nucleus reaches past the candidates
            w = _topp_mask(prs, s.p, None if idx is None else np.asarray(idx, dtype=np.int64))
            if w[-1] > 0:
                # the cut lands ON the last candidate: a lower logit outside the list whose exp rounds to the same f32 probability
                # (at a lower vocabulary index) would be taken first by candle's stable full-vocabulary walk -- full-vector fallback
                return None
            return w
        raise ValueError(f"{s.kind} does not sample from candidates")

    def weights_from_logits(self, logits: np.ndarray) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        """Full-vector path (Sampling::All, oversized k, TopP fallback): LogitsProcessor::sample on host logits.  Returns
        (weights, ids): the vector sample_multinomial draws from and, when that vector is NOT indexed by token id (sample_topk /
        sample_topk_topp draw over the k selected probabilities and map the drawn position back through `indices`), the ids."""
        s = self.sampling
        logits = np.asarray(logits, dtype=np.float32)
        if s.kind == "ArgMax":
            w = np.zeros_like(logits)
            w[int(np.argmax(logits))] = 1.0
            return w, None
        x = logits * np.float32(1.0 / s.temperature)
        e = np.exp(x - x.max(), dtype=np.float32)
        prs = (e / e.sum(dtype=np.float32)).astype(np.float32)
        if s.kind == "All":
            return prs, None
        if s.kind == "TopP":
            return (prs if (s.p <= 0.0 or s.p >= 1.0) else _topp_mask(prs, s.p)), None
        if s.kind == "TopKThenTopP" and s.k >= logits.shape[0]:
            return _topp_mask(prs, s.p), None
        if s.k >= logits.shape[0]:
            return prs, None
        # candle selects the k largest PROBABILITIES (select_nth_unstable_by); equal probabilities: higher logit, then lower
        # index (the same refinement as the device's ranking by logit); the draw runs over them in that order
        keep = np.lexsort((np.arange(prs.shape[0]), -logits.astype(np.float64), -prs.astype(np.float64)))[: s.k]
        sub = prs[keep]
        if s.kind == "TopKThenTopP" and not (s.p <= 0.0 or s.p >= sub.sum(dtype=np.float32)):
            sub = _topp_mask(sub, s.p)
        return sub, keep

    # -- the draw ----------------------------------------------------------------------------------------------------
    def draw(self, weights: np.ndarray) -> int:
        """sample_multinomial: WeightedIndex::new(weights)?.sample(&mut self.rng) -- running f32 sums, one u32 of the ChaCha12
        stream, index of the first running sum greater than the uniform value (aha_hip_rng_weighted_index)."""
        return self.rng.weighted_index(np.asarray(weights, dtype=np.float32))


class GenerationContext:
    """common/generate.rs:21-53 (defaults: repeat_penalty 1.0, repeat_last_n 64)."""

    def __init__(self, temperature: Optional[float], top_p: Optional[float], top_k: Optional[int], repeat_penalty: Optional[float],
                 repeat_last_n: Optional[int], seed: int, initial_seq_len: int, max_tokens: int):
        self.logit_processor = get_logit_processor(temperature, top_p, top_k, seed)
        self.repeat_penalty = 1.0 if repeat_penalty is None else float(repeat_penalty)
        self.repeat_last_n = 64 if repeat_last_n is None else int(repeat_last_n)
        self.seqlen_offset = 0
        self.seq_len = initial_seq_len
        self.sample_len = max_tokens


def penalty_context(repeat_penalty: float, repeat_last_n: Optional[int], generated: Sequence[int]) -> Tuple[float, Sequence[int]]:
    """use_repeat_penalty's slicing (sample.rs:47-53): (effective penalty, ids it applies to)."""
    if np.float32(repeat_penalty) == np.float32(1.0) or repeat_last_n == 0:
        return 1.0, []
    start_at = 0 if repeat_last_n is None else max(len(generated) - repeat_last_n, 0)
    ids = generated[start_at:]
    return (float(repeat_penalty), ids) if len(ids) else (1.0, [])


def draw_from_candidates(lp: "LogitsProcessor", w: np.ndarray, idx: np.ndarray) -> int:
    """The weighted draw over a candidate list, in the order candle draws in.  sample_topk / sample_topk_topp draw over the k
    selected probabilities in selection order and map the position back through `indices` -- the candidate order.  sample_topp
    zeroes the tail INSIDE the full-vocabulary vector and calls sample_multinomial(prs): WeightedIndex takes its running sums in
    TOKEN-ID order, so the surviving candidates are put in vocabulary order first (zeros in between add nothing to an f32 running
    sum): the same u32 of the stream then picks the token the full-vector path (weights_from_logits) picks."""
    idx = np.asarray(idx)
    if lp.sampling.kind == "TopP":
        order = np.argsort(idx, kind="stable")
        return int(idx[order][lp.draw(np.asarray(w)[order])])
    return int(idx[lp.draw(w)])


def sample_and_push(ctx: GenerationContext, model, argmax_token: int, generated: List[int]) -> int:
    """common/generate.rs:70-86 on the logits of the model's last forward call."""
    lp = ctx.logit_processor
    pen, pctx = penalty_context(ctx.repeat_penalty, ctx.repeat_last_n, generated)
    V = model.text_cfg.vocab_size
    if lp.sampling.kind == "ArgMax" and pen == 1.0:
        token = int(argmax_token)
    else:
        k = 1 if lp.sampling.kind == "ArgMax" else lp.candidates_needed(V)
        token = None
        if k:
            vals, idx, mx, se = model.sample_candidates(pctx, pen, lp.sampling.temperature if lp.sampling.kind != "ArgMax" else 0.0, k)
            if lp.sampling.kind == "ArgMax":
                token = int(idx[0])  # arg-max of the penalised logits (first maximal index)
            else:
                w = lp.weights_from_candidates(vals, mx, se, idx)
                if w is not None:
                    token = draw_from_candidates(lp, w, idx)
        if token is None:  # full-vector fallback
            logits = model.last_logits()
            if pen != 1.0:
                seen = set()
                for t in pctx:
                    if t not in seen and 0 <= t < V:
                        logits[t] = logits[t] / np.float32(pen) if logits[t] >= 0 else logits[t] * np.float32(pen)
                    seen.add(t)
            w, ids = lp.weights_from_logits(logits)
            pos = lp.draw(w)
            token = pos if ids is None else int(ids[pos])
    generated.append(token)
    return token


def generate_generic_sampled(model, input_ids: Sequence[int], ctx: GenerationContext, data=None) -> List[int]:
    """generate_generic_text's token loop (common/generate.rs:87-112) with the sampler of `ctx`."""
    eos = set(model.stop_token_ids())
    generated: List[int] = []
    _, am = model.forward_initial(input_ids, ctx.seqlen_offset, data, want_logits=False)
    tok = sample_and_push(ctx, model, am, generated)
    ctx.seqlen_offset += ctx.seq_len  # prepare_for_next_token
    ctx.seq_len = 1
    for _ in range(1, ctx.sample_len):
        _, am = model.forward_step(tok, ctx.seqlen_offset, want_logits=False)
        tok = sample_and_push(ctx, model, am, generated)
        if tok in eos:
            break
        ctx.seqlen_offset += ctx.seq_len
    model.clear_cache()
    return generated


def generate_asr(model, audio_datas, temperature: float, top_p: Optional[float] = None, seed: int = 34562,
                 max_tokens: int = 1024) -> Tuple[List[int], int]:
    """Qwen3AsrGenerateModel::generate's own loop (src/models/qwen3_asr/generate.rs:130-186), which is NOT generate_generic:
      * sampler = get_logit_processor(Some(temperature), top_p, None, seed) with the default seed 34562 -- top_k is always
        None (ArgMax below 1e-7, else Sampling::All / TopP), there is no repeat penalty, and ONE processor (one RNG stream)
        serves all audio chunks of the request;
      * `audio_datas` = one (input_ids, MultiModalData) per <= 1200 s chunk (processor.rs:126-179); every chunk runs up to
        max_tokens forwards -- the audio features only on the first (seqlen_offset 0) -- and stops AFTER pushing either of the
        model's two eos ids (checked on the very first sampled token too); the KV cache is cleared between chunks;
      * the tokens of all chunks go into one list (decoded as one string by the reference).
    Returns (generated ids, prompt token count)."""
    lp = get_logit_processor(temperature, top_p, None, seed)
    ctx = GenerationContext(None, None, None, None, None, seed, 0, max_tokens)
    ctx.logit_processor = lp          # repeat_penalty 1.0: sample_and_push reduces to logit_processor.sample
    eos = set(model.stop_token_ids()[:2])
    generated: List[int] = []
    prompt_tokens = 0
    for input_ids, data in audio_datas:
        seq_len, seqlen_offset = len(input_ids), 0
        prompt_tokens += seq_len
        tok = None
        for _ in range(max_tokens):
            if seqlen_offset == 0:
                _, am = model.forward_initial(input_ids, 0, data, want_logits=False)
            else:
                _, am = model.forward_step(tok, seqlen_offset, want_logits=False)
            mark = len(generated)
            tok = sample_and_push(ctx, model, am, generated)
            assert len(generated) == mark + 1
            if tok in eos:
                break
            seqlen_offset += seq_len
            seq_len = 1
        model.clear_cache()
    return generated, prompt_tokens


def generate_stream_generic_text(model, decode, input_ids: Sequence[int], ctx: GenerationContext, data=None, device_chunk: int = 8):
    """generate_stream_generic_text (src/models/common/generate.rs:161-229) as a Python generator of decoded text pieces.

    Differences from generate_generic that the reference really has and this mirrors: the loop runs `sample_len` iterations
    including the prefill one; a piece that decodes to text containing U+FFFD (an incomplete UTF-8 sequence) is held back -- the
    token joins `error_tokens`, which are decoded together with the next token, and are dropped after more than 3 of them -- and
    such a held-back token is NOT checked against the eos ids (the `continue` precedes the check); the cache is cleared at the end.
    `decode(ids) -> str` is the tokenizer hook (TokenizerModel::token_decode, src/tokenizer/mod.rs).
    Greedy requests without a repeat penalty keep the device-resident loop: tokens are produced `device_chunk` at a time by
    aha_hip_decode_greedy and still yielded one by one (a token produced past an eos id inside a chunk is discarded)."""
    eos = set(model.stop_token_ids())
    lp = ctx.logit_processor
    generated: List[int] = []
    error_tokens: List[int] = []
    greedy_fast = lp.sampling.kind == "ArgMax" and ctx.repeat_penalty == 1.0 and hasattr(model, "decode_greedy")
    pending: List[int] = []          # tokens already produced on the device, not yet consumed by the loop
    try:
        for _ in range(ctx.sample_len):
            if ctx.seqlen_offset == 0:
                _, am = model.forward_initial(input_ids, 0, data, want_logits=False)
                tok = sample_and_push(ctx, model, am, generated)
            elif greedy_fast:
                if not pending:
                    n = max(1, min(device_chunk, ctx.sample_len - len(generated)))
                    pending = list(model.decode_greedy(generated[-1], ctx.seqlen_offset, n))
                    if not pending:
                        break
                tok = pending.pop(0)
                generated.append(tok)
            else:
                _, am = model.forward_step(generated[-1], ctx.seqlen_offset, want_logits=False)
                tok = sample_and_push(ctx, model, am, generated)
            piece = decode(error_tokens + [tok])
            # prepare_for_next_token (generate.rs:54-66)
            ctx.seqlen_offset += ctx.seq_len
            ctx.seq_len = 1
            if "�" in piece:
                error_tokens.append(tok)
                if len(error_tokens) > 3:
                    error_tokens.clear()
                continue
            error_tokens.clear()
            yield piece
            if tok in eos:
                break
    finally:
        model.clear_cache()
