"""Per-token cost of the D11 sampling step at the real vocabulary (151 936): device candidates (aha_hip_sample_candidates,
k = 20, repeat penalty over 64 ids) + host top-p/draw, against the reference-shaped path (608 KB logits to the host, penalty,
full-vocabulary softmax, top-k selection there)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aha_amd import sampling as hs
from aha_amd.configs import tiny_qwen3
from aha_amd.model import HipInferenceModel
from aha_amd.weights import qwen3_text_weights

cfg = tiny_qwen3(layers=1, hidden=256, heads=4, kv_heads=2, inter=512, vocab=151936)
m = HipInferenceModel(cfg, qwen3_text_weights(cfg, seed=0))
m.forward_initial([1, 2, 3, 4], 0, want_logits=False)
ctx = list(range(1000, 1064))
lp = hs.get_logit_processor(0.6, 0.95, 20, seed=1)
n = 200
for _ in range(10):
    m.sample_candidates(ctx, 1.1, 0.6, 20)
t0 = time.perf_counter()
for _ in range(n):
    vals, idx, mx, se = m.sample_candidates(ctx, 1.1, 0.6, 20)
t1 = time.perf_counter()
for _ in range(n):
    vals, idx, mx, se = m.sample_candidates(ctx, 1.1, 0.6, 20)
    tok = int(idx[lp.draw(lp.weights_from_candidates(vals, mx, se))])
t2 = time.perf_counter()
for _ in range(20):
    logits = m.last_logits()
t3 = time.perf_counter()
for _ in range(20):
    logits = m.last_logits()
    for t in ctx:
        logits[t] = logits[t] / np.float32(1.1) if logits[t] >= 0 else logits[t] * np.float32(1.1)
    tok = lp.draw(lp.weights_from_logits(logits))
t4 = time.perf_counter()
print(f"device candidates (C ABI call, k=20, 64-id penalty): {(t1 - t0) / n * 1e6:8.1f} us/token")
print(f"  + host top-p and draw                            : {(t2 - t1) / n * 1e6:8.1f} us/token")
print(f"logits to host only (608 KB D2H)                     : {(t3 - t2) / 20 * 1e6:8.1f} us/token")
print(f"logits to host + numpy softmax/top-k/top-p/draw      : {(t4 - t3) / 20 * 1e6:8.1f} us/token")
m.close()
