"""TEST INFRASTRUCTURE ONLY -- CPU restatement (pure Python integers + numpy f32) of the random draw behind the reference's
non-greedy sampler.  Nothing under aha_amd/ may import this module.

Path: `sample_and_push` (reference src/models/common/generate.rs:70-86) -> `LogitsProcessor::sample` -> `sample_multinomial`,
on the processor `get_logit_processor` builds with `LogitsProcessor::from_sampling(seed, ..)` / `::new(seed, ..)`
(src/models/common/sample.rs:7-37; seed 299792458 in common/generate.rs:408,452, 34562 in qwen3_asr/generate.rs:134).

WHICH CRATES: the processor lives in candle-transformers 0.9.2, whose own dependency is **rand 0.9.2** (Cargo.lock:590-606:
"rand 0.9.2"), NOT the rand 0.10.1 aha lists for its own use (Cargo.toml:46).  So the stream is
    rand 0.9.2  `StdRng::seed_from_u64`  = rand_core 0.9.5 `SeedableRng::seed_from_u64` (PCG32 expansion of the u64 into 32 bytes)
                                         -> rand_chacha 0.9.0 `ChaCha12Rng::from_seed` (Cargo.lock:3405-3445, 3455-3462)
    rand 0.9.2  `distr::weighted::WeightedIndex<f32>::new / sample`  over  `distr::uniform::UniformFloat<f32>`.
None of these crates is under /root/reference and none can be built here (no cargo): the algorithms below are restated from
their published sources.  PARITY UNPINNED: checked here only against (i) the RFC 7539 section 2.3.2 ChaCha20 block test
vector (the same block function with 20 instead of 12 rounds) and (ii) the C++ implementation behind the C ABI
(csrc/sampler_rng.hip) written independently from the same description.  One vector from a cargo build
(`StdRng::seed_from_u64(299792458).next_u32()`) would pin it.

What this does NOT settle: for `Sampling::TopK` / `TopKThenTopP` candle draws over the k probabilities in the order
`select_nth_unstable_by` leaves them (candle-transformers generation/mod.rs sample_topk / sample_topk_topp), which the Rust
standard library does not specify; the mirror feeds them ranked by (probability descending, logit descending, index ascending) -- one
of the orders that selection may produce (oracle/sampling.py topk_order).  `Sampling::All` and `Sampling::TopP` (top_k = None, the
ASR loop's sampler) draw over the vocabulary in index order and are fully defined by this module.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF


def pcg32_seed_bytes(state: int, n_bytes: int = 32) -> bytes:
    """rand_core 0.9 `SeedableRng::seed_from_u64`: the seed is filled 4 bytes at a time from a PCG32 (XSH-RR) stream; the state is
    advanced BEFORE each output ("to get away from the input value, in case it has low Hamming weight")."""
    MUL, INC = 6364136223846793005, 11634580027462260723
    out = bytearray()
    state &= M64
    while len(out) < n_bytes:
        state = (state * MUL + INC) & M64
        xorshifted = (((state >> 18) ^ state) >> 27) & M32
        rot = state >> 59
        x = ((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & M32
        out += x.to_bytes(4, "little")
    return bytes(out[:n_bytes])


def _rotl(x: int, n: int) -> int:
    return ((x << n) | (x >> (32 - n))) & M32


def chacha_block(state: Sequence[int], rounds: int) -> List[int]:
    """The ChaCha block function on a 16-word state (RFC 7539 section 2.3): `rounds` / 2 double rounds (column + diagonal quarter
    rounds), then the word-wise addition of the input state."""
    x = list(state)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & M32; x[d] = _rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & M32; x[b] = _rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & M32; x[d] = _rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & M32; x[b] = _rotl(x[b] ^ x[c], 7)

    for _ in range(rounds // 2):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & M32 for a, b in zip(x, state)]


CHACHA_CONSTANTS = (0x61707865, 0x3320646E, 0x79622D32, 0x6B206574)   # "expand 32-byte k"


class StdRng:
    """rand 0.9.2 `StdRng` = rand_chacha 0.9.0 `ChaCha12Rng`: key = the 32 seed bytes (little-endian words), 64-bit block counter in
    words 12-13 starting at 0, 64-bit stream id 0 in words 14-15; a `BlockRng` buffer of 64 words = 4 consecutive blocks, handed out
    word by word (`next_u32`); the first call finds the buffer exhausted (index 64) and generates blocks 0-3."""

    def __init__(self, seed_bytes: bytes):
        assert len(seed_bytes) == 32
        self.key = [int.from_bytes(seed_bytes[4 * i:4 * i + 4], "little") for i in range(8)]
        self.counter = 0
        self.buf: List[int] = []
        self.index = 64

    @classmethod
    def seed_from_u64(cls, seed: int) -> "StdRng":
        return cls(pcg32_seed_bytes(seed, 32))

    def _refill(self):
        self.buf = []
        for _ in range(4):
            st = list(CHACHA_CONSTANTS) + self.key + [self.counter & M32, (self.counter >> 32) & M32, 0, 0]
            self.buf += chacha_block(st, 12)
            self.counter = (self.counter + 1) & M64
        self.index = 0

    def next_u32(self) -> int:
        if self.index >= 64:
            self._refill()
        v = self.buf[self.index]
        self.index += 1
        return v


def _f32_next_down(x: np.float32) -> np.float32:
    """`decrease_masked`: the next representable f32 below a positive finite x (bits - 1)."""
    return np.array([np.array([x], dtype=np.float32).view(np.uint32)[0] - np.uint32(1)], dtype=np.uint32).view(np.float32)[0]


class UniformF32:
    """rand 0.9.2 `UniformFloat<f32>::new(low, high)` (half-open) and `sample`."""

    def __init__(self, low: float, high: float):
        low, high = np.float32(low), np.float32(high)
        if not (low < high):
            raise ValueError("EmptyRange")
        max_rand = np.float32(1.0) - np.float32(2.0 ** -23)            # (u32::MAX >> 9).into_float_with_exponent(0) - 1.0
        scale = np.float32(high - low)
        if not np.isfinite(scale):
            raise ValueError("NonFinite")
        while np.float32(np.float32(scale * max_rand) + low) >= high:   # the largest sample must stay below `high`
            scale = _f32_next_down(scale)
        self.low, self.scale = low, scale

    def sample(self, rng: StdRng) -> np.float32:
        bits = (rng.next_u32() >> 9) | 0x3F800000                        # 23 random mantissa bits, exponent 0: a value in [1, 2)
        value1_2 = np.array([bits], dtype=np.uint32).view(np.float32)[0]
        value0_1 = np.float32(value1_2 - np.float32(1.0))
        return np.float32(np.float32(value0_1 * self.scale) + self.low)  # no mul_add in the crate


class WeightedIndexF32:
    """rand 0.9.2 `WeightedIndex<f32>`: running f32 sums of all weights but the last, a uniform draw in [0, total), the index of the
    first running sum that is GREATER than the draw (`partition_point(|w| w <= chosen)`)."""

    def __init__(self, weights: Sequence[float]):
        w = np.asarray(weights, dtype=np.float32)
        if w.size == 0:
            raise ValueError("InvalidInput")
        if not np.all(w >= 0):                                            # also rejects NaN
            raise ValueError("InvalidWeight")
        total = np.float32(w[0])
        cum = []
        for x in w[1:]:
            cum.append(total)
            total = np.float32(total + x)
        if total == 0:
            raise ValueError("InsufficientNonZero")
        if not np.isfinite(total):
            raise ValueError("Overflow")
        self.cum = np.asarray(cum, dtype=np.float32)
        self.total = total
        self.dist = UniformF32(0.0, total)

    def sample(self, rng: StdRng) -> int:
        chosen = self.dist.sample(rng)
        return int(np.searchsorted(self.cum, chosen, side="right"))      # count of running sums <= chosen


def sample_multinomial(rng: StdRng, prs: Sequence[float]) -> int:
    """candle-transformers 0.9.2 `LogitsProcessor::sample_multinomial`: a fresh WeightedIndex over the vector, one draw."""
    return WeightedIndexF32(prs).sample(rng)
