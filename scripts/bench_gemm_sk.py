"""GPU microbenchmark: the persistent GEMM kernel (csrc/kernels_gemm_sk.hip) against the one-tile-per-block plans at the cfg 3 prefill
shapes.  For every shape: the automatic plan with AHA_GEMM_STREAMK=0 semantics (forced through the plan override where needed), then the
persistent kernel on 256- and 192-column tiles with every cut style.  Times are HIP events over 20 launches (launch gaps included)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aha_amd import ops, build, _lib
build.build()
dev = torch.device("cuda:0")
PAIRS = _lib.ACT_SILU_MUL_PAIRS
shapes = [("qkv", 1542, 6144, 4096, 0, False), ("o", 1542, 4096, 4096, 0, True), ("gateup", 1542, 24576, 4096, PAIRS, False),
          ("down", 1542, 4096, 12288, 0, True), ("vit_qkv", 4096, 3456, 1152, 0, False), ("vit_fc2", 4096, 1152, 4288, 0, True),
          ("m1536_gateup", 1536, 24576, 4096, PAIRS, False), ("m2048_gateup", 2048, 24576, 4096, PAIRS, False),
          ("big", 8192, 8192, 8192, 0, False)]
only = os.environ.get("AHA_GEMM_ONLY")


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it


for name, M, N, K, act, has_res in shapes:
    if only and name not in only.split(","):
        continue
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    W = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    res = torch.randn(M, N // 2 if act == PAIRS else N, device=dev, dtype=torch.bfloat16) if has_res else None
    fn = lambda: ops.gemm(A, W, None, res, act)
    flops = 2.0 * M * N * K
    row = [f"{name:13s} M={M:5d} N={N:6d} K={K:6d}"]
    plans = [("auto", (0, 0)), ("q256", (256, 1)), ("q256sk2", (256, 2))]
    if not has_res:
        plans.append(("q192", (192, 1)))
    for tile in (1256, 1192):
        if tile == 1192 and has_res:
            continue
        for cut in (0, 1, 2, 3, 4, 11, 12, 13):
            plans.append((f"s{tile - 1000}c{cut}", (tile, cut)))
    for label, plan in plans:
        if act == PAIRS and plan == (256, 2):
            continue
        ops.gemm_plan(*plan)
        try:
            us = timeit(fn)
        finally:
            ops.gemm_plan(0, 0)
        row.append(f"{label} {us:7.1f}us {flops / us / 1e6:6.0f}TF")
    print(" | ".join(row), flush=True)
# the separate RMSNorm pass a persistent-kernel projection pays where the split-K plans fold it into their reduce pass
x = torch.randn(1542, 4096, device=dev, dtype=torch.bfloat16)
w = torch.randn(4096, device=dev, dtype=torch.bfloat16)
print(f"rmsnorm_rows 1542 x 4096: {timeit(lambda: ops.rmsnorm(x, w, 1e-6)):.1f} us", flush=True)
