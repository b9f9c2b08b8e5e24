import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print("cpu_count", os.cpu_count())
W = torch.randn(12288, 4096); x = torch.randn(1, 4096)
for nt in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(nt)
    for _ in range(3): y = x @ W.t()
    t = time.perf_counter()
    for _ in range(10): y = x @ W.t()
    t1 = (time.perf_counter() - t) / 10
    t = time.perf_counter()
    for _ in range(100): z = (x * 2.0).to(torch.bfloat16).float()
    t2 = (time.perf_counter() - t) / 100
    print(nt, "matvec ms", round(t1 * 1e3, 2), "tiny-op us", round(t2 * 1e6, 1))
