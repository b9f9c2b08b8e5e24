"""Times every (tile, split-K) plan the library can run on the cfg 3 / ViT / cfg 5 GEMM shapes (debug override of plan_gemm) and prints
the automatic choice beside them: input for the cost constants in kernels_gemm.hip plan_gemm."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aha_amd import ops, build
build.build()
dev = torch.device("cuda:0")
shapes = [("qkv", 1542, 6144, 4096), ("o", 1542, 4096, 4096), ("down", 1542, 4096, 12288), ("gateup_plain", 1542, 24576, 4096), ("big", 8192, 8192, 8192),
          ("vit_qkv", 4096, 3456, 1152), ("vit_proj", 4096, 1152, 1152), ("vit_fc1", 4096, 4304, 1152), ("vit_fc2", 4096, 1152, 4304),
          ("qkv_2k", 2048, 4096, 1024), ("o_2k", 2048, 1024, 2048), ("down_2k", 2048, 1024, 3072),
          ("gateup_2k", 2048, 6144, 1024),
          ("asr_qkv", 390, 2688, 896), ("asr_o", 390, 896, 896), ("asr_fc1", 390, 3584, 896), ("asr_fc2", 390, 896, 3584),
          ("asr_t_qkv", 406, 4096, 1024), ("asr_t_o", 406, 1024, 2048), ("asr_t_gateup", 406, 6144, 1024), ("asr_t_down", 406, 1024, 3072),
          ("c1_qkv", 128, 4096, 1024), ("c1_o", 128, 1024, 2048), ("c1_down", 128, 1024, 3072),
          ("s8b_qkv", 128, 6144, 4096), ("s8b_o", 128, 4096, 4096), ("s8b_down", 128, 4096, 12288), ("s16_o", 16, 4096, 4096),
          ("qkv_41k", 40980, 6144, 4096), ("o_41k", 40980, 4096, 4096)]
if len(sys.argv) > 1:
    shapes = [x for x in shapes if any(x[0].startswith(p) for p in sys.argv[1:])]
plans = [(0, 0), (2128, 1), (128, 1), (128, 2), (128, 3), (128, 4), (128, 6), (128, 8), (256, 1), (192, 1), (256, 2), (256, 3), (256, 4), (256, 6)]
def t(A, W, it=10):
    for _ in range(2): ops.gemm(A, W)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): ops.gemm(A, W)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it
for name, M, N, K in shapes:
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    W = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    row = []
    for tile, sk in plans:
        ops.gemm_plan(tile, sk)
        try:
            us = t(A, W)
            row.append(f"{tile}/{sk}: {us:7.1f}")
        except Exception as e:
            row.append(f"{tile}/{sk}: err")
    ops.gemm_plan(0, 0)
    print(f"{name:9s} M={M:5d} N={N:5d} K={K:5d} | " + " | ".join(row), flush=True)
