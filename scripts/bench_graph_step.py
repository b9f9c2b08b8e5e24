"""hipGraph replay of one decode step against its stream launches (aha_hip_debug_graph_step), for the BASELINE models at their
configured context: is the launch path -- not the kernels -- what a small model's decode step waits for?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aha_amd import configs, weights as W
from aha_amd.model import HipInferenceModel
import __graft_entry__
__graft_entry__.build()
dev = "cuda:0"
for name, cfg, prompt, mk in (("qwen3-0.6b", configs.qwen3_0_6b(), 2048, lambda c: W.qwen3_text_weights(c, seed=0, device=dev)),
                              ("qwen3vl-8b text", configs.qwen3vl_8b_text(), 1542, lambda c: W.qwen3_text_weights(c, seed=0, device=dev))):
    w = mk(cfg)
    m = HipInferenceModel(cfg, w, kv_reserve_tokens=4096)
    del w
    ids = torch.randint(0, 151643, (prompt,), generator=torch.Generator().manual_seed(1)).tolist()
    m.forward_initial(ids, 0, want_logits=False)
    a, b = m.debug_graph_step(50)
    print(f"{name:16s} context {prompt}: launches {a:8.1f} us/step | graph replay {b:8.1f} us/step | ratio {b / a:.3f}", flush=True)
    m.close()
    torch.cuda.empty_cache()
