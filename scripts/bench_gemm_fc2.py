"""ViT fc2 (M = 4096, N = 1152, K = 4304 -> padded 4352) with bias + residual under every plan that can run it, and the real K = 4304
beside it (8-wave kernel: K is not a multiple of 64).  Also the other ViT projections on the 192-column split-K slabs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aha_amd import ops, build
build.build()
dev = torch.device("cuda:0")
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it
for name, M, N, K in [("fc2", 4096, 1152, 4304), ("fc2_pad", 4096, 1152, 4352), ("proj", 4096, 1152, 1152), ("o", 1542, 4096, 4096), ("down", 1542, 4096, 12288),
                      ("down_0.6b_2k", 2048, 1024, 3072), ("o_0.6b_2k", 2048, 1024, 2048), ("down_0.6b_4k", 4096, 1024, 3072), ("merger_fc2", 1024, 4096, 4608)]:
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    W = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    r = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    row = []
    ref = None
    for tile, sk in [(0, 0), (128, 1), (256, 1), (256, 2), (256, 3), (256, 4), (192, 2), (192, 3), (192, 4), (0, 0)]:
        ops.gemm_plan(tile, sk)
        try:
            us = t(lambda: ops.gemm(A, W, b, r))
            out = ops.gemm(A, W, b, r)
            if ref is None: ref = out
            d = (out.float() - ref.float()).abs().max().item()
            row.append(f"{tile}/{sk}: {us:6.1f}" + (f" (d {d:.3f})" if d else ""))
        except Exception as e:
            row.append(f"{tile}/{sk}: err")
    ops.gemm_plan(0, 0)
    print(f"{name:8s} M={M} N={N} K={K} | " + " | ".join(row), flush=True)
