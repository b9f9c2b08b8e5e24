import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, copy
from aha_amd import configs
from aha_amd.weights import qwen3_text_weights
from oracle.numerics import Numerics
from oracle.qwen3 import OracleQwen3
t = configs.qwen3vl_8b_text(); one = copy.deepcopy(t); one.num_hidden_layers = 1; one.mrope_section = None; one.tie_word_embeddings = True
w = qwen3_text_weights(one, seed=0)
for nt in (256, 64, 16):
    torch.set_num_threads(nt)
    o = OracleQwen3(one, w, Numerics("bf16"))
    o.forward(list(range(32)), 0)
    t0 = time.perf_counter(); h = o.forward_hidden([5], None, 32); t1 = time.perf_counter(); o.nm.linear(h, o.lm_head); t2 = time.perf_counter()
    t3 = time.perf_counter(); h = o.forward_hidden([5], None, 33); t4 = time.perf_counter()
    print(nt, "layer", round(t1 - t0, 4), "head", round(t2 - t1, 4), "layer again", round(t4 - t3, 4), flush=True)
