"""TEST INFRASTRUCTURE ONLY -- rounding model shared by the oracle restatements.

Nothing under ``aha_amd/`` may import this package; only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg do, and only as the checker.

PARITY UNPINNED: the reference (jhqxxx/aha v0.2.6) cannot be compiled here (no cargo/rustc) and its arithmetic
lives in the un-vendored crates candle-core / candle-nn / candle-transformers 0.9.2 (Cargo.lock:497-593).  Its own
tests assert nothing about outputs, so there are no golden vectors to pin against.  What is
restated here is the reference's *op sequence* (cited file:line per function) under the following rounding model of
a Candle op, which is an assumption ("[unverified]"):

    every Candle op computes internally in >= f32 and rounds its OUTPUT tensor to the model dtype T
    (T = bf16 on the GPU target, f16 on the reference's CPU default -- utils/mod.rs:107 -- or f32).

Known candle-CPU sub-op roundings that this model does not apply by default are exposed as switches so their
effect can be measured: ``rmsnorm_in_T`` (m -> T, x/m -> T, *w -> T), ``attn_probs_rounded`` / ``attn_scores_rounded`` (the eager
attention's intermediate tensors).
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

_DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}


@dataclass
class Numerics:
    dtype: str = "bf16"          # model dtype T
    rmsnorm_in_T: bool = False   # candle-nn CPU rms_norm: m cast to T, then x / m * w evaluated in T
    attn_probs_rounded: bool = True   # softmax output materialised in T before P.V (eager path, modules.rs:788)
    attn_scores_rounded: bool = True  # q.k^T materialised in T, then `* scaling` materialised in T (eager path, modules.rs:782-783);
                                      # False = the scores stay f32 through scale, mask and softmax (the HIP f32 score chain, round 5)
    matmul_f64: bool = False     # accumulate GEMMs in f64 (ideal) instead of f32 (order-dependent)
    attn_row_block: int = 0      # > 0: eager attention evaluated `attn_row_block` query rows at a time (same arithmetic per row --
                                 # every row's softmax is independent -- without the (h, S, S) score tensor: 17 GB at N = 16 384)

    @property
    def torch_dtype(self):
        return _DT[self.dtype]

    def r(self, x: torch.Tensor) -> torch.Tensor:
        """Round an f32 tensor to T and return it as f32 (the materialisation point of a Candle op)."""
        if self.dtype == "f32":
            return x.to(torch.float32)
        return x.to(self.torch_dtype).to(torch.float32)

    def linear(self, x: torch.Tensor, w: torch.Tensor, b: torch.Tensor | None = None) -> torch.Tensor:
        """candle_nn::Linear: x . W^T (+ b).  Matmul output rounded to T, then the bias add rounded to T."""
        if self.matmul_f64:
            y = (x.double() @ w.double().t()).float()
        else:
            y = x @ w.t()
        y = self.r(y)
        if b is not None:
            y = self.r(y + b)
        return y

    def matmul(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        if self.matmul_f64:
            return self.r((a.double() @ b.double()).float())
        return self.r(a @ b)
