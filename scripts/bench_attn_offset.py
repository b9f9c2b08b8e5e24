import os, sys
sys.path.insert(0, "/root/repo")
os.environ.setdefault("AHA_ATTN_TIME", "5")
import torch
from aha_amd import ops
dev = torch.device("cuda:0")
nh, kvh, d = 32, 8, 128
for S, off in [(2624, 38356), (2560, 0), (2560, 17920), (5184, 35796), (1312, 39668), (10240, 30740)]:
    L = S + off
    q = torch.randn(S, nh * d, device=dev, dtype=torch.bfloat16)
    k = torch.randn(L, kvh * d, device=dev, dtype=torch.bfloat16)
    v = torch.randn(L, kvh * d, device=dev, dtype=torch.bfloat16)
    print(f"S={S} off={off}", flush=True)
    ops.attn_prefill(q, k, v, nh, kvh, d, off, True)
