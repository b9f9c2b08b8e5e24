"""GPU microbenchmark of the MFMA GEMM at the cfg3 prefill / ViT shapes (automatic plan), each with the epilogue the model calls it with
(o / down: + residual; ViT: bias, bias + residual, bias + GELU; fc2 over the padded MLP width 4352 -- csrc/vision_tower.hip Ipad).  AHA_GEMM_QUAD=0 forces the 8-wave 256^2 kernel, AHA_GEMM_ONLY=a,b picks shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aha_amd import ops, build, _lib
build.build()
dev = torch.device("cuda:0")
# (name, M, N, K, act, bias, residual)
shapes = [("qkv", 1542, 6144, 4096, 0, 0, 0), ("o", 1542, 4096, 4096, 0, 0, 1), ("gateup", 1542, 24576, 4096, 4, 0, 0), ("down", 1542, 4096, 12288, 0, 0, 1),
          ("vit_qkv", 4096, 3456, 1152, 0, 1, 0), ("vit_proj", 4096, 1152, 1152, 0, 1, 1), ("vit_fc1", 4096, 4304, 1152, 1, 1, 0),
          ("vit_fc2", 4096, 1152, 4352, 0, 1, 1), ("big", 8192, 8192, 8192, 0, 0, 0)]
only = os.environ.get("AHA_GEMM_ONLY")
for name, M, N, K, act, hb, hr in shapes:
    if only and name not in only.split(","):
        continue
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    W = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    b = torch.randn(N, device=dev, dtype=torch.bfloat16) if hb else None
    r = torch.randn(M, N, device=dev, dtype=torch.bfloat16) if hr else None
    for _ in range(3): ops.gemm(A, W, b, r, act=act)
    torch.cuda.synchronize()
    it = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): ops.gemm(A, W, b, r, act=act)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / it
    epi = "+".join(x for x in (("bias" if hb else ""), ("gelu" if act == 1 else "silu*up" if act == 4 else ""), ("residual" if hr else "")) if x) or "plain"
    print(f"{name:9s} M={M:5d} N={N:6d} K={K:6d} {epi:14s} {us:9.1f} us  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s", flush=True)
