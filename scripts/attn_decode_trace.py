"""In-kernel timeline of the fused decode attention (AHA_ATTN_TRACE=1): host-driven steps at a few context lengths."""
import os, sys
os.environ["AHA_ATTN_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aha_amd import build, configs, weights
build.build()
from aha_amd.model import HipInferenceModel
cfg = configs.tiny_qwen3(layers=4, hidden=4096, heads=32, kv_heads=8, inter=1024, vocab=2048, tie=True)
w = weights.qwen3_text_weights(cfg, seed=0, device="cuda:0")
m = HipInferenceModel(cfg, w, kv_reserve_tokens=140000)
for L in [int(x) for x in os.environ.get("LENS", "1536,40960").split(",")]:
    m.clear_cache()
    ids = torch.randint(0, 2048, (L,)).tolist()
    off = 0
    while off < L:
        n = min(8192, L - off)
        _, tok = m.forward_initial(ids[off:off + n], off, want_logits=False)
        off += n
    print(f"L={L}", flush=True)
    for i in range(4):
        _, tok = m.forward_step(tok, off + i, want_logits=False)
    sys.stderr.flush()
m.close()
