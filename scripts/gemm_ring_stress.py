"""Automatic-plan GEMMs of the cfg 2 / cfg 4 / ViT shapes (ring kernel, K slices, 256 x 128 tiles) repeated 300 times beside a competing\nkernel on another stream: every result must equal the first bit for bit (a hazard in the LDS-DMA ring would show as a rare mismatch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aha_amd import ops, build, _lib
build.build()
dev = torch.device("cuda:0")
torch.manual_seed(1)
shapes = [(406, 4096, 1024), (406, 1024, 2048), (406, 1024, 3072), (390, 896, 3584), (390, 3584, 896), (128, 4096, 12288), (4096, 1152, 1152), (2048, 4096, 1024), (2048, 1024, 2048), (70, 1736, 264)]
data = []
for (M, N, K) in shapes:
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    W = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.05
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    r = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    data.append((A, W, b, r, ops.gemm(A, W), ops.gemm(A, W, b, r, _lib.ACT_NONE)))
s2 = torch.cuda.Stream()
big = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
bad = 0
for it in range(300):
    if it % 3 == 0:
        with torch.cuda.stream(s2):
            big2 = big @ big     # a competing kernel on another stream
    for (A, W, b, r, c0, c1) in data:
        if not torch.equal(ops.gemm(A, W), c0): bad += 1
        if not torch.equal(ops.gemm(A, W, b, r, _lib.ACT_NONE), c1): bad += 1
torch.cuda.synchronize()
print("stress: 300 rounds x", len(data), "shapes x 2 epilogues, mismatches:", bad)
