"""Decode-attention bandwidth at long context: 2-layer model at Qwen3-VL-8B attention dims, prefill L tokens, then
profiled decode steps; reports the attn_decode kernel class (HIP events) as GB/s of algorithmic KV bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aha_amd import build, configs, weights
build.build()
from aha_amd.model import HipInferenceModel
cfg = configs.tiny_qwen3(layers=2, hidden=4096, heads=32, kv_heads=8, inter=1024, vocab=2048, tie=True)
w = weights.qwen3_text_weights(cfg, seed=0, device="cuda:0")
m = HipInferenceModel(cfg, w, kv_reserve_tokens=70000)
for L in [int(x) for x in os.environ.get("LENS", "2048,8192,32768,65536").split(",")]:
    m.clear_cache()
    ids = torch.randint(0, 2048, (L,)).tolist()
    # chunked prefill keeps scratch small
    off = 0
    while off < L:
        n = min(8192, L - off)
        _, tok = m.forward_initial(ids[off:off + n], off, want_logits=False)
        off += n
    m.decode_greedy(tok, off, 4)
    m.set_profiling(True)
    out = m.decode_greedy(tok, off + 4, 16)
    p = m.get_profile("attn_decode")
    m.set_profiling(False)
    us = 1e3 * p["ms"] / p["launches"]
    print(f"L={L:6d}  attn_decode {us:8.2f} us/launch  {p['bytes']/p['launches']/1e6:8.2f} MB  {p['bytes']/(p['ms']*1e-3)/1e9:8.1f} GB/s  ({p['bytes']/(p['ms']*1e-3)/8e12*100:.1f}% of 8 TB/s)", flush=True)
m.close()
