"""What ONE rank of an 8-way sharded cfg 5 text prefill (Qwen3-VL-8B text stack, 36 layers, 40 980 tokens) computes, timed on one GPU
with the collectives stubbed out (the callbacks return at once, so the K / V pages / partial sums of the other ranks are garbage and the
logits meaningless -- the launches, shapes and byte counts are the real ones).  Gives the compute side of the scaling estimate in
DESIGN.md section 6: context-parallel rank r of W (csrc/model.hip cp_make_plan) for every r, and tensor-parallel rank 0 of W.
    python scripts/shard_rank_time.py [W=8] [S=40980]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from aha_amd import build
build.build()
from aha_amd.configs import qwen3vl_8b_text
from aha_amd.model import HipInferenceModel
from aha_amd.weights import qwen3_text_weights

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 40980
dev = torch.device("cuda:0")
cfg = qwen3vl_8b_text()
w = qwen3_text_weights(cfg, seed=0, device=dev)
ids = [int(x) for x in np.random.default_rng(1).integers(0, 151643, size=S)]


def timed(m, n=2):
    best = 1e9
    for _ in range(n + 1):
        m.clear_cache()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.forward_initial(ids, 0, want_logits=False)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


m = HipInferenceModel(cfg, w, kv_reserve_tokens=S + 64)
t1 = timed(m)
print(f"one GPU, whole prompt: {1e3 * t1:8.1f} ms", flush=True)
nbytes = [0]


def ag(ptr, n):
    nbytes[0] += n * W


ts = []
for r in range(W):
    m.set_context_parallel(r, W, all_gather=ag)
    nbytes[0] = 0
    t = timed(m, 1)
    ts.append(t)
    print(f"context-parallel rank {r} of {W}: {1e3 * t:8.1f} ms   (exchange stubbed: {nbytes[0] / 2 / 1e6:.0f} MB would cross per prefill)", flush=True)
m.set_context_parallel(0, 1)
m.close()
print(f"context-parallel: slowest rank {1e3 * max(ts):.1f} ms, fastest {1e3 * min(ts):.1f} ms -> compute-side speed-up {t1 / max(ts):.2f} x on {W} GPUs", flush=True)
cnt = {"ar": 0, "rs": 0, "ag": 0}
tp = HipInferenceModel(cfg, w, kv_reserve_tokens=S + 64, tp_rank=0, tp_size=W, allreduce=lambda p, n: cnt.__setitem__("ar", cnt["ar"] + n * 4),
                       reduce_scatter=lambda p, n: cnt.__setitem__("rs", cnt["rs"] + n * 4 * W), all_gather=lambda p, n: cnt.__setitem__("ag", cnt["ag"] + n * W))
for k in cnt: cnt[k] = 0
tt = timed(tp, 1)
print(f"tensor-parallel rank 0 of {W} (sequence-parallel, collectives stubbed): {1e3 * tt:8.1f} ms -> compute-side speed-up {t1 / tt:.2f} x; "
      f"per prefill and rank the collectives carry {cnt['rs'] / 2 / 1e9:.2f} GB of f32 reduce-scatter + {cnt['ag'] / 2 / 1e9:.2f} GB of bf16 all-gather buffers", flush=True)
tp.close()

# phase breakdown (library profiler: HIP events per launch group) of the whole prompt and of context-parallel rank 0
from aha_amd import parallel
m = HipInferenceModel(cfg, w, kv_reserve_tokens=S + 64)
for label, r in (("one GPU", None), ("context-parallel rank 0", 0), (f"context-parallel rank {W // 2}", W // 2)):
    if r is not None:
        m.set_context_parallel(r, W, all_gather=lambda p, n: None)
    m.clear_cache()
    m.forward_initial(ids, 0, want_logits=False)
    m.clear_cache()
    m.set_profiling(True)
    m.forward_initial(ids, 0, want_logits=False)
    ph = parallel.read_prefill_phases(m, {})
    m.set_profiling(False)
    print(label, {k: v for k, v in ph.items() if k.endswith("_s")}, flush=True)
m.close()
