"""Per-tile time of the 256^2 GEMM kernel as the number of busy CUs grows (same K): tells a per-CU limit (time flat) from a shared
L2 / fabric limit (time grows with the tile count)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aha_amd import ops, build
build.build()
dev = torch.device("cuda:0")
K = int(os.environ.get("K", "8192"))
for M, N in [(256, 256), (4096, 4096), (8192, 8192)]:
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    W = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    ops.gemm_plan(256, 1)
    for _ in range(3): ops.gemm(A, W)
    torch.cuda.synchronize()
    it = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): ops.gemm(A, W)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / it
    tiles = (M // 256) * (N // 256)
    rounds = (tiles + 255) // 256
    print(f"M={M:5d} N={N:5d} K={K} tiles {tiles:5d} rounds {rounds}: {us:8.1f} us  {us/rounds/(K/64):6.3f} us per K tile  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s", flush=True)
