"""Diagnostic: persistent decode kernel vs launch-per-op path, per-step logits diff."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aha_amd.configs import tiny_qwen3
from aha_amd.weights import qwen3_text_weights
from aha_amd.model import HipInferenceModel

layers = int(os.environ.get("L", "2"))
cfg = tiny_qwen3(layers=layers, hidden=4096, heads=32, kv_heads=8, inter=12288, vocab=2048, tie=False)
w = qwen3_text_weights(cfg, seed=3)
os.environ["AHA_DECODE_MEGA"] = "0"; multi = HipInferenceModel(cfg, w)
os.environ["AHA_DECODE_MEGA"] = "1"; mega = HipInferenceModel(cfg, w)
for S in [int(x) for x in os.environ.get("S", "1500,1500,700,2000").split(",")]:
    g = torch.Generator().manual_seed(100 + S)
    ids = torch.randint(0, cfg.vocab_size, (S,), generator=g).tolist()
    multi.clear_cache(); mega.clear_cache()
    a, am = multi.forward_initial(ids, 0); b, bm = mega.forward_initial(ids, 0)
    tok, off = am, S
    res = []
    for step in range(8):
        a, am = multi.forward_step(tok, off); b, bm = mega.forward_step(tok, off)
        ha, hb = multi.debug_last_hidden(), mega.debug_last_hidden()
        res.append((int((a != b).sum()), int((ha != hb).sum())))
        tok, off = am, off + 1
    print("S", S, "mismatched logits/hidden per step:", res, flush=True)
