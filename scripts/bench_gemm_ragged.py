"""(AHA_GEMM_ROW5=0 switches the fifth fragment row of the 192-column kernel off: run twice for the A/B.)
What the ragged last row tile costs: the cfg 3 gate+up / qkv GEMMs at M = 1536 (six full 256-row tiles) and M = 1542 (+ a 6-row
tile) on the 256^2 and the 256 x 192 tile (debug plan override), real-activation-like operands.  us per launch, 20 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aha_amd import ops, build
build.build()
dev = torch.device("cuda:0")
def t(A, W, it=20):
    for _ in range(3): ops.gemm(A, W)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): ops.gemm(A, W)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it
for name, N, K in (("gateup", 24576, 4096), ("qkv", 6144, 4096)):
    W = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    for M in (1280, 1536, 1537, 1542, 1568, 1792, 2049, 2054):
        A = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        row = []
        for tile in (256, 192):
            ops.gemm_plan(tile, 1)
            row.append(f"{tile}: {t(A, W):7.1f} us")
        ops.gemm_plan(0, 0)
        row.append(f"auto: {t(A, W):7.1f} us")
        print(f"{name:7s} M={M:5d} N={N:6d} K={K} | " + " | ".join(row), flush=True)
