"""GPU microbenchmark of the prefill attention kernel alone (text decoder geometry: 32 q heads, 8 kv heads, head_dim 128).
AHA_ATTN_TIME=<reps> makes the C ABI debug op time the kernel with HIP events and print ms/launch.  A/B knobs of the kernel:
AHA_ATTN_WAVES=4|8, AHA_ATTN_SCHED=0|1.  `python scripts/bench_attn.py 16384 40980` times causal launches of those lengths.
(The elimination variants recorded in profiles/r01_attn_prefill_anatomy.md came from a temporary template parameter.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("AHA_ATTN_TIME", "5")
import torch
from aha_amd import ops, build
build.build()
dev = torch.device("cuda:0")
nh, kvh, d = 32, 8, 128
shapes = [(8192, False), (8192, True), (2048, True)]
if len(sys.argv) > 1:
    shapes = [(int(x), True) for x in sys.argv[1:]]
for S, causal in shapes:
    q = torch.randn(S, nh * d, device=dev, dtype=torch.bfloat16)
    k = torch.randn(S, kvh * d, device=dev, dtype=torch.bfloat16)
    v = torch.randn(S, kvh * d, device=dev, dtype=torch.bfloat16)
    ops.attn_prefill(q, k, v, nh, kvh, d, causal=causal)
    fl = 4.0 * S * S * d * nh * (0.5 if causal else 1.0)
    print(f"  S={S} causal={causal}: {fl/1e12:.3f} TFLOP nominal", flush=True)
