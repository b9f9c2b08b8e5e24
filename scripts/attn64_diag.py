"""Diagnostic: where a kernel form of the prefill attention differs from the oracle (shapes of tests/test_ops_gpu.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from aha_amd import ops, build
build.build()
import test_ops_gpu as T
dev = torch.device("cuda:0")
cases = [(300, 37, 8, 2), (600, 0, 16, 8), (100, 333, 16, 8), (300, 0, 16, 8)]
for S, off, nh, kvh in cases:
    d, L = 128, S + off
    q, k, v = T.rnd((S, nh * d), 40), T.rnd((L, kvh * d), 41), T.rnd((L, kvh * d), 42)
    ref = T._attn_ref(q, k, v, nh, kvh, d, True, off, T.NM_F32SCORES).float()
    refe = T._attn_ref(q, k, v, nh, kvh, d, True, off).float()
    for form in (16, 64, 65):
        ops.attn_form(form)
        got = ops.attn_prefill(q.to(dev), k.to(dev), v.to(dev), nh, kvh, d, off, True).float().cpu()
        for name, r in (("f32-score oracle", ref), ("eager oracle", refe)):
            rs = r.abs().amax(-1, keepdim=True)
            ulp = T.ulp_bf16(torch.maximum(r.abs(), rs))
            e = ((got - r).abs() / ulp)
            idx = int(e.argmax())
            row, col = idx // e.shape[1], idx % e.shape[1]
            print(f"S={S} off={off} nh={nh} kvh={kvh} form {form} vs {name}: max {float(e.max()):.2f} ulp at row {row} head {col // d} dim {col % d}; "
                  f"> 3 ulp: {int((e > 3).sum())}, > 4: {int((e > 4).sum())}, rows with > 4: {sorted(set((e > 4).nonzero()[:, 0].tolist()))[:12]}", flush=True)
    ops.attn_form(-1)
