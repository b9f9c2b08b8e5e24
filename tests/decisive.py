"""Test scaffolding: synthetic checkpoints whose greedy token is DECISIVE at every step.

With N(0, 0.02^2) weights the top-1 / top-2 logit margin of a 151 936-wide vocabulary is the gap between the two largest of
~150 k Gaussians: exponential with mean ~0.2 std, i.e. below the fp tolerance of ANY bf16 implementation at most steps, so
"bit-identical greedy tokens" cannot be asserted on them (round-2 verdict, weak #1).  These helpers keep every layer's
weights as drawn (all kernels do their full-size work, the layer outputs are a large part of the residual stream) and
only restructure the embedding table and the head so that ONE logit stands out:

  * untied head (Qwen3-VL-8B): embed' = s * embed, lm_head'[pi(t)] = embed[t] for a seeded permutation pi of the text
    vocabulary  =>  logit_{pi(t)} = embed[t] . RMSNorm(s embed[t] + layer updates) ~ 0.02 H cos(x, e_t), against N(0, 0.02^2 H)
    for every other row.  Greedy decoding walks the permutation: t -> pi(t) -> pi(pi(t)) ...  (no fixed points, no eos ids).
  * tied head (Qwen3-0.6B): the final norm weight becomes a seeded sign vector D (|D| = 1, bf16-exact) and odd rows are
    embed[2i+1] = D * embed[2i]  =>  logit_{2i+1}(h = D x_hat, x ~ e_{2i}) = e_{2i} . x_hat (the spike) and vice versa: greedy
    decoding alternates 2i <-> 2i+1 (D^2 = 1 leaves no longer cycles to a diagonal norm weight).

`s` is a power of two (bf16-exact scaling).  The tests assert the oracle's own top-1/top-2 margin at every step (>= 0.5 std)
before they demand exact equality of the free-running sequences.
"""
from __future__ import annotations

import torch

TEXT_VOCAB = 151643   # ids below the first special token (eos ids 151643 / 151645 are never predicted)


def permutation(n: int, seed: int, device="cpu") -> torch.Tensor:
    """A seeded permutation of range(n) without fixed points (a single n-cycle: i -> order[(pos(i) + 1) % n])."""
    g = torch.Generator().manual_seed(seed)
    order = torch.randperm(n, generator=g)
    pi = torch.empty(n, dtype=torch.int64)
    pi[order] = torch.roll(order, -1)
    return pi.to(device)


def make_untied_decisive(w: dict, embed_name: str, head_name: str, scale: float = 128.0, seed: int = 7, n_text: int = TEXT_VOCAB):
    """In place.  Returns pi (int64, on the embedding's device)."""
    e = w[embed_name]
    n = min(n_text, e.shape[0])
    pi = permutation(n, seed, e.device)
    head = w[head_name]
    head[pi] = e[:n]                      # lm_head'[pi(t)] = embed[t]   (rows >= n keep their random values)
    w[embed_name] = (e.float() * scale).to(e.dtype)
    return pi


def make_tied_decisive(w: dict, embed_name: str, norm_name: str, scale: float = 32.0, seed: int = 7, n_text: int = TEXT_VOCAB):
    """In place.  Returns the sign vector D (float32, on the embedding's device)."""
    e = w[embed_name]
    H = e.shape[1]
    g = torch.Generator().manual_seed(seed)
    D = (torch.randint(0, 2, (H,), generator=g).float() * 2 - 1).to(e.device)
    n = min(n_text, e.shape[0]) // 2 * 2
    ef = e.float()
    ef[1:n:2] = ef[0:n:2] * D
    w[embed_name] = (ef * scale).to(e.dtype)
    w[norm_name] = D.to(w[norm_name].dtype)
    return D


def margin_std(logits) -> float:
    """(top-1 - top-2) / std of a logit vector (numpy or torch)."""
    t = torch.as_tensor(logits, dtype=torch.float32).reshape(-1)
    top = torch.topk(t, 2).values
    return float(top[0] - top[1]) / float(t.std())
