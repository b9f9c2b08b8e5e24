"""GPU microbenchmark of the MFMA GEMM at the cfg3 prefill / ViT shapes (automatic plan).  AHA_GEMM_QUAD=0 forces the 8-wave 256^2 kernel, AHA_GEMM_ONLY=a,b picks shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aha_amd import ops, build, _lib
build.build()
dev = torch.device("cuda:0")
shapes = [("qkv", 1542, 6144, 4096, 0), ("o", 1542, 4096, 4096, 0), ("gateup", 1542, 24576, 4096, 4), ("down", 1542, 4096, 12288, 0),
          ("vit_qkv", 4096, 3456, 1152, 0), ("vit_proj", 4096, 1152, 1152, 0), ("vit_fc1", 4096, 4304, 1152, 1), ("vit_fc2", 4096, 1152, 4304, 0),
          ("big", 8192, 8192, 8192, 0)]
only = os.environ.get("AHA_GEMM_ONLY")
for name, M, N, K, act in shapes:
    if only and name not in only.split(","):
        continue
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    W = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    for _ in range(3): ops.gemm(A, W, act=act)
    torch.cuda.synchronize()
    it = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): ops.gemm(A, W, act=act)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / it
    print(f"{name:9s} M={M:5d} N={N:6d} K={K:6d} {us:9.1f} us  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s", flush=True)
