"""ONE context-parallel rank of W at the cfg 5 text prompt (collectives stubbed, as scripts/shard_rank_time.py), three prefills: the workload for
`rocprofv3 --kernel-trace --stats` of a rank's launches (profiles/r05_cp_rank_kernel_stats.md).
    python scripts/cp_rank_profile.py [rank=4] [W=8] [S=40980]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from aha_amd import build
build.build()
from aha_amd.configs import qwen3vl_8b_text
from aha_amd.model import HipInferenceModel
from aha_amd.weights import qwen3_text_weights

r = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S = int(sys.argv[3]) if len(sys.argv) > 3 else 40980
cfg = qwen3vl_8b_text()
w = qwen3_text_weights(cfg, seed=0, device=torch.device("cuda:0"))
ids = [int(x) for x in np.random.default_rng(1).integers(0, 151643, size=S)]
m = HipInferenceModel(cfg, w, kv_reserve_tokens=S + 64)
m.set_context_parallel(r, W, all_gather=lambda ptr, n: None)
for _ in range(3):
    m.clear_cache()
    m.forward_initial(ids, 0, want_logits=False)
torch.cuda.synchronize()
m.close()
print("done")
