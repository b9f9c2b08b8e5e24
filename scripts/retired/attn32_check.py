"""EXPERIMENT helper: the 32-row prefill attention forms (variants 4 / 5) against variant 3 on random data, and their timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aha_amd import ops, build
build.build()
dev = torch.device("cuda:0")
nh, kvh, d = 32, 8, 128
for S, causal in [(1542, True), (700, False), (2048, True)]:
    g = torch.Generator().manual_seed(S)
    q = torch.randn(S, nh * d, generator=g).bfloat16().to(dev)
    k = torch.randn(S, kvh * d, generator=g).bfloat16().to(dev)
    v = torch.randn(S, kvh * d, generator=g).bfloat16().to(dev)
    outs = {}
    for smx in (3, 4):
        ops.attn_variant(smx)
        outs[smx] = ops.attn_prefill(q, k, v, nh, kvh, d, 0, causal).float()
    ops.attn_variant(-1)
    dlt = (outs[4] - outs[3]).abs().max().item()
    print(f"S={S} causal={causal}: max |v4 - v3| = {dlt:.5f} (row scale {outs[3].abs().max().item():.3f})", flush=True)
    assert dlt < 0.02
