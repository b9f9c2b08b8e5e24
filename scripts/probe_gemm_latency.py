"""Experiment (GPU): is the four-wave GEMM's k step bound by the latency of its operand stream?  The same launch with ldw = 0 / lda = 0
(every W / A row aliases row 0: all LDS-DMA requests hit the caches) against the real strides.  Results are garbage by construction."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aha_amd import ops, build, _lib
build.build()
lib = _lib.lib()
dev = torch.device("cuda:0")


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


for name, M, N, K, act in [("gateup1536", 1536, 24576, 4096, 4), ("gateup1542", 1542, 24576, 4096, 4), ("big", 8192, 8192, 8192, 0)]:
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    W = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    Cm = torch.empty(M, N // 2 if act == 4 else N, device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    for plan in [(256, 1), (192, 1)]:
        row = [f"{name} tile {plan[0]}"]
        for label, lda, ldw in [("real", K, K), ("ldw=0", K, 0), ("lda=0", 0, K), ("both=0", 0, 0)]:
            ops.gemm_plan(*plan)
            try:
                fn = lambda: lib.aha_hip_gemm(A.data_ptr(), W.data_ptr(), Cm.data_ptr(), M, N, K, lda, ldw, Cm.shape[1], None, None, act, st)
                us = timeit(fn)
            finally:
                ops.gemm_plan(0, 0)
            row.append(f"{label} {us:7.1f}us")
        print(" | ".join(row), flush=True)
