"""DIAGNOSTIC ONLY (never on the product path): what the vendor library (torch.matmul -> hipBLASLt / rocBLAS) reaches on the same box
for the shapes of scripts/bench_gemm.py, with the same operand fills.  Used to check the claim in profiles/r02_gemm_anatomy.md that
the full-chip bf16 GEMM is bounded by the clock / power envelope rather than by the kernel's schedule."""
import os, sys
import torch
dev = torch.device("cuda:0")
shapes = [("qkv", 1542, 6144, 4096), ("o", 1542, 4096, 4096), ("gateup", 1542, 24576, 4096), ("down", 1542, 4096, 12288),
          ("vit_qkv", 4096, 3456, 1152), ("vit_proj", 4096, 1152, 1152), ("vit_fc1", 4096, 4304, 1152), ("vit_fc2", 4096, 1152, 4304),
          ("big", 8192, 8192, 8192)]
fills = os.environ.get("FILLS", "weights,zeros").split(",")
for fill in fills:
    for name, M, N, K in shapes:
        if fill != "weights" and name != "big":
            continue
        if fill == "zeros":
            A = torch.zeros(M, K, device=dev, dtype=torch.bfloat16); W = torch.zeros(N, K, device=dev, dtype=torch.bfloat16)
        elif fill == "randn":
            A = torch.randn(M, K, device=dev, dtype=torch.bfloat16); W = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        else:
            A = torch.randn(M, K, device=dev, dtype=torch.bfloat16); W = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
        Wt = W.t()
        for _ in range(5): torch.matmul(A, Wt)
        torch.cuda.synchronize()
        it = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): torch.matmul(A, Wt)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / it
        print(f"vendor {fill:8s} {name:9s} M={M:5d} N={N:6d} K={K:6d} {us:9.1f} us  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s", flush=True)
