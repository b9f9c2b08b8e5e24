"""8192^3 GEMM on different operand fills: the chip clocks to its power budget, so TFLOP/s depends on the data's bit activity."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aha_amd import ops, build
build.build()
dev = torch.device("cuda:0")
M = N = K = 8192
fills = {"zeros": lambda *s: torch.zeros(*s, device=dev, dtype=torch.bfloat16),
         "uniform[-1,1)": lambda *s: (torch.rand(*s, device=dev) * 2 - 1).to(torch.bfloat16),
         "randn": lambda *s: torch.randn(*s, device=dev, dtype=torch.bfloat16),
         "randn*0.02 (weights) x randn (acts)": None}
for name, f in fills.items():
    if f is None:
        A = torch.randn(M, K, device=dev, dtype=torch.bfloat16); W = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    else:
        A, W = f(M, K), f(N, K)
    for _ in range(3): ops.gemm(A, W)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gemm(A, W)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"{name:38s} {us:8.1f} us  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s", flush=True)
