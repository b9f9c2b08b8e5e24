"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy, f32) of the reference's sampling step D11.  Nothing under
aha_amd/ may import this module; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may.

Path: `sample_and_push` (reference src/models/common/generate.rs:70-86) = logits -> f32 -> `use_repeat_penalty`
(src/models/common/sample.rs:41-60) -> `LogitsProcessor::sample` built by `get_logit_processor` (sample.rs:7-38).

PARITY UNPINNED: the arithmetic lives in the third-party crate candle-transformers 0.9.2 (Cargo.lock:590-593;
`utils::apply_repeat_penalty`, `generation::{LogitsProcessor, Sampling}`), which is not under /root/reference and cannot be
built here (no cargo).  The functions below restate its published algorithm; the reference holds no golden vectors for it (the penalty rule alone is
cross-checked against transformers' RepetitionPenaltyLogitsProcessor in tests/test_sampling_cpu.py).
The random draw (`rand::distr::weighted::WeightedIndex` over a seeded `StdRng`) is NOT restated: everything here stops at
the probability vector the draw is made from, which is the deterministic part the device path has to reproduce.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np


@dataclass
class Sampling:
    """candle_transformers::generation::Sampling as get_logit_processor builds it (sample.rs:13-37)."""
    kind: str                      # "ArgMax" | "All" | "TopK" | "TopP" | "TopKThenTopP"
    temperature: float = 1.0       # f64 in the reference (cast from the request's f32)
    k: int = 0
    p: float = 1.0


def get_logit_processor(temperature: Optional[float], top_p: Optional[float], top_k: Optional[int]) -> Sampling:
    """sample.rs:7-38.  A temperature below 1e-7 means greedy (sample.rs:13)."""
    if temperature is not None and np.float32(temperature) < np.float32(1e-7):
        temperature = None
    t64 = None if temperature is None else float(np.float32(temperature))  # `temp as f64`
    p64 = None if top_p is None else float(np.float32(top_p))
    if top_k is None:
        # LogitsProcessor::new(seed, temperature, top_p): None -> ArgMax; Some(t) -> All / TopP
        if t64 is None:
            return Sampling("ArgMax")
        return Sampling("All", t64) if p64 is None else Sampling("TopP", t64, p=p64)
    if t64 is None:
        return Sampling("ArgMax")
    return Sampling("TopK", t64, k=top_k) if p64 is None else Sampling("TopKThenTopP", t64, k=top_k, p=p64)


def use_repeat_penalty(repeat_penalty: float, repeat_last_n: Optional[int], logits: np.ndarray, context: Sequence[int]) -> np.ndarray:
    """sample.rs:41-60: unchanged if penalty == 1 or repeat_last_n == Some(0); else the last repeat_last_n ids of context."""
    logits = np.asarray(logits, dtype=np.float32)
    if np.float32(repeat_penalty) == np.float32(1.0) or repeat_last_n == 0:
        return logits.copy()
    start_at = 0 if repeat_last_n is None else max(len(context) - repeat_last_n, 0)  # saturating_sub
    return apply_repeat_penalty(logits, repeat_penalty, context[start_at:])


def apply_repeat_penalty(logits: np.ndarray, penalty: float, context: Sequence[int]) -> np.ndarray:
    """candle_transformers::utils::apply_repeat_penalty: every DISTINCT token id of the context (HashSet) has its logit divided
    by the penalty when >= 0 and multiplied otherwise; ids outside the vocabulary are ignored (`logits.get_mut` is None)."""
    out = np.asarray(logits, dtype=np.float32).copy()
    pen = np.float32(penalty)
    seen = set()
    for t in context:
        t = int(t)
        if t in seen:
            continue
        seen.add(t)
        if 0 <= t < out.shape[0]:
            out[t] = out[t] / pen if out[t] >= 0 else out[t] * pen
    return out


def softmax_last_dim(x: np.ndarray) -> np.ndarray:
    """candle_nn::ops::softmax_last_dim on f32: exp(x - max) / sum."""
    x = np.asarray(x, dtype=np.float32)
    e = np.exp(x - x.max(), dtype=np.float32)
    return (e / e.sum(dtype=np.float32)).astype(np.float32)


def _topp_mask(prs: np.ndarray, top_p: float) -> np.ndarray:
    """LogitsProcessor::sample_topp: walk the probabilities in descending order (`sort_by`, a STABLE sort: equal
    probabilities stay in position order); once the running sum has reached top_p the remaining ones are zeroed (the one
    that crosses the threshold is kept)."""
    prs = prs.copy()
    order = np.argsort(-prs, kind="stable")
    cumsum = np.float32(0.0)
    for i in order:
        if cumsum >= np.float32(top_p):
            prs[i] = 0.0
        else:
            cumsum = np.float32(cumsum + prs[i])
    return prs


def topk_order(prs: np.ndarray, logits: np.ndarray) -> np.ndarray:
    """Descending order of the PROBABILITIES (candle's key in sample_topk / sample_topk_topp:
    `select_nth_unstable_by(k, |i, j| prs[j].total_cmp(&prs[i]))`).  The selection is unstable, so which of several EQUAL
    f32 probabilities makes the cut at position k is unspecified in the reference; the rule fixed here (and in the host mirror
    and on the device) is the refinement "higher logit first, then lower index" -- the f32 softmax is monotone in the logit,
    so this is one of the orders candle's selection may produce, and it is the order a ranking by logit gives."""
    idx = np.arange(prs.shape[0])
    return np.lexsort((idx, -np.asarray(logits, dtype=np.float64), -np.asarray(prs, dtype=np.float64)))


def final_weights(logits: np.ndarray, s: Sampling) -> np.ndarray:
    """The (unnormalised) weight vector over the vocabulary that LogitsProcessor::sample hands to sample_multinomial
    (WeightedIndex normalises).  ArgMax: one-hot at the first maximal index."""
    logits = np.asarray(logits, dtype=np.float32)
    V = logits.shape[0]
    if s.kind == "ArgMax":
        w = np.zeros(V, dtype=np.float32)
        w[int(np.argmax(logits))] = 1.0
        return w
    inv_t = np.float32(1.0 / s.temperature)           # `&logits / temperature`: affine by 1/T (f64) applied in f32
    prs = softmax_last_dim(logits * inv_t)
    if s.kind == "All":
        return prs
    if s.kind == "TopP":
        # LogitsProcessor::sample: `if p <= 0.0 || p >= 1.0 { sample_multinomial(&prs) } else { sample_topp(..) }`
        return prs if (s.p <= 0.0 or s.p >= 1.0) else _topp_mask(prs, s.p)
    if s.k >= V:                                       # sample_topk / sample_topk_topp fall through to the full vector
        return prs if s.kind == "TopK" else _topp_mask(prs, s.p)
    keep = topk_order(prs, logits)[: s.k]              # select_nth_unstable_by(k, descending probability): the k largest
    sub = prs[keep]
    if s.kind == "TopKThenTopP":
        sum_p = sub.sum(dtype=np.float32)
        if not (s.p <= 0.0 or s.p >= sum_p):
            sub = _topp_mask(sub, s.p)
    w = np.zeros(V, dtype=np.float32)
    w[keep] = sub
    return w


def topk_candidates(logits: np.ndarray, k: int, temperature: float):
    """What aha_hip_sample_candidates returns: the k largest logits ordered by (value desc, index asc), the max and
    sum exp((x - max) / T) over the whole vocabulary (f64 accumulation: the device sum is compared with a tolerance)."""
    logits = np.asarray(logits, dtype=np.float32)
    order = np.lexsort((np.arange(logits.shape[0]), -logits.astype(np.float64)))[:k]
    inv_t = np.float32(1.0 / temperature) if temperature > 0 else np.float32(1.0)
    mx = logits.max()
    se = np.exp((logits.astype(np.float64) - float(mx)) * float(inv_t)).sum()
    return logits[order], order.astype(np.uint32), float(mx), float(se)
