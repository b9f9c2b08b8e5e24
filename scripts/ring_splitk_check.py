"""128^2 split-K plans (gemm_glds_ring_kernel<ACT_PARTIAL_F32> + reduce pass) against an f64 product: error next to the unsplit plan's,
run-to-run bit equality (a race in the staging ring would show as a difference)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aha_amd import ops, build
build.build()
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (M, N, K) in [(64, 512, 1024), (406, 1024, 2048), (406, 1024, 3072), (390, 896, 3584), (128, 4096, 12288), (16, 4096, 4096), (130, 1000, 1032), (257, 520, 4104), (4096, 1152, 1152), (2048, 4096, 1024), (700, 328, 448)]:
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    W = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.05
    ref = (A.double() @ W.double().t())
    row = []
    for tile, sk in [(2128, 1), (128, 1), (128, 2), (128, 3), (128, 4), (128, 6), (128, 8), (256, 4)]:
        ops.gemm_plan(tile, sk)
        c0 = ops.gemm(A, W)
        same = all(torch.equal(ops.gemm(A, W), c0) for _ in range(20))
        err = (c0.double() - ref).abs().max().item()
        nbad = ((c0.double() - ref).abs() > 0.02 * ref.abs().clamp(min=1.0)).sum().item()
        row.append(f"{tile}/{sk}: {err:.4f} bad {nbad} {'same' if same else 'DIFFERS'}")
    ops.gemm_plan(0, 0)
    print(M, N, K, " | ".join(row), flush=True)
