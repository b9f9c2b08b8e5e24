"""Round 5: the prefill attention's two score chains (AHA_ATTN_SMX 1 = the reference's two bf16 roundings, 3 = f32 scores) against the oracle
with f32 and with f64 accumulation, at BASELINE cfg 2 (Qwen3-0.6B, decisive-margin checkpoint of tests/decisive.py, 2048-token prefill) --
the case whose free-running worst-step bound the f32 chain touches.  GPU + ~2 min of host oracle.  -> gpurun_out/score_chain_error.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.set_num_threads(min(32, os.cpu_count() or 1))
from aha_amd import build, ops
build.build()
from aha_amd.configs import qwen3_0_6b
from aha_amd.model import HipInferenceModel
from aha_amd.weights import qwen3_text_weights
from oracle.numerics import Numerics
from oracle import qwen3 as oq
import decisive

def rel(got, ref):
    ref = np.asarray(ref, dtype=np.float32).reshape(-1); got = np.asarray(got, dtype=np.float32).reshape(-1)
    s = float(ref.std())
    return float(np.abs(got - ref).max()) / s, float(np.sqrt(((got - ref) ** 2).mean())) / s

gpu = torch.device("cuda:0")
cfg = qwen3_0_6b()
out = {}
for label, dec in (("plain_checkpoint", False), ("decisive_checkpoint", True)):
    w = qwen3_text_weights(cfg, seed=0, device=gpu)
    if dec:
        decisive.make_tied_decisive(w, "model.embed_tokens.weight", "model.norm.weight", scale=32.0, seed=7)
    m = HipInferenceModel(cfg, w)
    wc = {k: v.cpu() for k, v in w.items()}
    ids = [int(x) for x in np.random.default_rng(2).integers(0, 151643, size=2048)]
    refs = {}
    for name, nm in (("f32_rounded", Numerics("bf16", attn_row_block=1024)), ("f64_rounded", Numerics("bf16", matmul_f64=True, attn_row_block=1024)),
                     ("f64_f32scores", Numerics("bf16", matmul_f64=True, attn_scores_rounded=False, attn_row_block=1024))):
        t0 = time.time()
        o = oq.OracleQwen3(cfg, wc, nm)
        refs[name] = o.forward(ids, 0).reshape(-1).numpy()
        print(label, name, f"{time.time() - t0:.1f} s", flush=True)
    rep = {"oracle_f32_vs_f64": rel(refs["f32_rounded"], refs["f64_rounded"]), "oracle_f64_f32scores_vs_f64_rounded": rel(refs["f64_f32scores"], refs["f64_rounded"])}
    for smx in (1, 3):
        ops.attn_variant(smx)
        m.clear_cache()
        got, _ = m.forward_initial(ids, 0)
        rep[f"hip_smx{smx}"] = {k: rel(got, v) for k, v in refs.items()}
    ops.attn_variant(-1)
    out[label] = rep
    print(label, json.dumps(rep), flush=True)
    m.close()
    del w
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "score_chain_error.json"), "w"), indent=1)
