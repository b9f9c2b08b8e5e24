"""GPU microbenchmark of the weight-streaming matvec at the Qwen3-VL-8B decode shapes (rotating over enough distinct
matrices to defeat the 256 MB Infinity Cache).  Knobs: AHA_GEMV_GRID / AHA_GEMV_R / AHA_GEMV_U env vars."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aha_amd import ops, build
build.build()
dev = torch.device("cuda:0")
shapes = [("qkv", 6144, 4096, True), ("o", 4096, 4096, False), ("down", 4096, 12288, False), ("gateup", 12288, 4096, True),
          ("lm_head", 151936, 4096, True), ("qkv0.6", 4096, 1024, True), ("down0.6", 1024, 3072, False)]
sel = os.environ.get("SHAPES")
for name, N, K, norm in shapes:
    if sel and name not in sel.split(","):
        continue
    gate = name == "gateup"
    per = N * K * 2 * (2 if gate else 1)
    ncopy = int(os.environ.get('NCOPY', 0)) or max(2, min(16, int(1.2e9 // per)))
    Ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(ncopy)]
    W2 = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(ncopy)] if gate else None
    x = torch.randn(K, device=dev, dtype=torch.bfloat16)
    nw = torch.ones(K, device=dev, dtype=torch.bfloat16) if norm else None
    def run(i):
        if gate:
            return ops.gemv_gate_up(Ws[i % ncopy], W2[i % ncopy], x, nw)
        return ops.gemv(Ws[i % ncopy], x, nw)
    for i in range(ncopy): run(i)
    torch.cuda.synchronize()
    iters = ncopy * 6
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): run(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{name:8s} N={N:6d} K={K:5d} {per/1e6:8.1f} MB  {us:8.2f} us/launch  {per/us/1e6:6.2f} TB/s (incl. launch gaps)", flush=True)
