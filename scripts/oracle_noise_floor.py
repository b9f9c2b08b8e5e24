"""Noise floor of the rounding model itself: the oracle restatement of the full-depth Qwen3-VL-8B text stack (36 layers, S = 96) run
three ways -- f32 accumulation, f64 accumulation, f32 with un-rounded softmax probabilities -- same ops, same bf16 rounding points.
The differences between these runs are what ANY implementation that differs from the oracle only in accumulation order (or in
where the softmax is normalised) must be expected to show; tests/test_baseline_fullsize_parity_gpu.py takes its full-depth bound
from this table (profiles/r03_oracle_noise_floor.md).  CPU only, ~4 min on 8 cores, ~50 GB of host memory."""
import sys, time, torch, numpy as np, gc
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aha_amd.configs import qwen3vl_8b_text
from aha_amd.weights import qwen3_text_weights
from oracle.numerics import Numerics
from oracle import qwen3 as oq
torch.set_num_threads(8)
cfg=qwen3vl_8b_text()
t0=time.time()
w=qwen3_text_weights(cfg, seed=0, prefix='model.language_model.')
nm=Numerics('bf16')
o=oq.OracleQwen3(cfg,w,nm,prefix='model.language_model.',consume=True)
del w; gc.collect()
print('oracle built', time.time()-t0, flush=True)
ids=[int(x) for x in np.random.default_rng(1).integers(0,151643,size=96)]
def run(nm2, depth_probe=(1,4,8,16,24,36)):
    o.nm=nm2
    o.clear_cache()
    x=o.embed_tokens(ids); s=len(ids)
    mask=oq.prepare_causal_attention_mask(s); cos,sin=oq.rope_cos_sin(o.inv_freq,0,s)
    outs={}
    for li in range(cfg.num_hidden_layers):
        x=o.decoder_layer(li,x,cos,sin,mask)
        if li+1 in depth_probe:
            h=oq.rms_norm(nm2,x[:,-1:],o.w[o.p+'norm.weight'],cfg.rms_norm_eps)
            outs[li+1]=nm2.linear(h,o.lm_head).reshape(-1).numpy()
    return outs
a=run(Numerics('bf16',matmul_f64=False)); print('f32 done',time.time()-t0,flush=True)
b=run(Numerics('bf16',matmul_f64=True)); print('f64 done',time.time()-t0,flush=True)
c=run(Numerics('bf16',matmul_f64=False,attn_probs_rounded=False)); print('unrounded-probs done',time.time()-t0,flush=True)
for d in a:
    s=float(b[d].std())
    print(d,'f32-vs-f64 max %.4f rms %.4f | probs-unrounded-vs-f32 max %.4f rms %.4f'%(np.abs(a[d]-b[d]).max()/s, np.sqrt(((a[d]-b[d])**2).mean())/s, np.abs(a[d]-c[d]).max()/s, np.sqrt(((a[d]-c[d])**2).mean())/s),flush=True)
