"""GPU diagnostic: error statistics of each HIP op vs the oracle (not a test; prints a table)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from test_ops_gpu import *  # noqa
from test_ops_gpu import _attn_ref, _rope_ref
from aha_amd import ops, _lib, build
build.build(); _lib.lib()
gpu = torch.device("cuda:0")

def stats(name, got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    d = (got - ref).abs()
    rms = ref.pow(2).mean().sqrt()
    u = ulp_bf16(ref)
    print(f"{name:40s} rms {float(rms):9.4f} maxabs {float(d.max()):9.5f} max/ulp_rms {float(d.max()/ulp_bf16(rms)):6.2f} "
          f"max_ulp_elem {float((d/u).max()):8.2f} exact {float((got==ref).float().mean()):.4f} nan {int((~torch.isfinite(got)).sum())}")

for N, K in [(4096, 1024), (4096, 12288), (1000, 512)]:
    W, x = rnd((N, K), 3, 0.02), rnd((K,), 4)
    ref = NM.linear(x.float()[None], W.float())[0]
    ref64 = Numerics("bf16", matmul_f64=True).linear(x.float()[None], W.float())[0]
    got = ops.gemv(W.to(gpu), x.to(gpu))
    stats(f"gemv {N}x{K} vs f32 oracle", got, ref)
    stats(f"gemv {N}x{K} vs f64 oracle", got, ref64)
    stats(f"   (f32 oracle vs f64 oracle)", ref, ref64)
for M, N, K in [(1542, 512, 4096), (64, 4352, 1152), (128, 128, 64)]:
    A, W = rnd((M, K), 11), rnd((N, K), 12, 0.02)
    ref = NM.linear(A.float(), W.float())
    ref64 = Numerics("bf16", matmul_f64=True).linear(A.float(), W.float())
    got = ops.gemm(A.to(gpu), W.to(gpu))
    stats(f"gemm {M}x{N}x{K} vs f32 oracle", got, ref)
    stats(f"gemm {M}x{N}x{K} vs f64 oracle", got, ref64)
S, nh, kvh, d, theta = 77, 4, 2, 128, 1e6
qkv = rnd((S, (nh + 2 * kvh) * d), 20)
qw, kw = rnd((d,), 21, 0.02, 1.0), rnd((d,), 22, 0.02, 1.0)
pos = (torch.arange(S, dtype=torch.int32) + 1234)[None].repeat(3, 1).contiguous()
axis = torch.zeros(d // 2, dtype=torch.int32)
rq, rk, rv = _rope_ref(qkv, qw, kw, pos, axis, nh, kvh, d, 1e-6, theta)
q, k, v = ops.qknorm_rope(qkv.to(gpu), qw.to(gpu), kw.to(gpu), pos.to(gpu), axis.to(gpu), nh, kvh, d, 1e-6, theta)
stats("rope q", q, rq); stats("rope k", k, rk)
# which positions differ
dq = (q.float().cpu() - rq).abs().reshape(S, nh, d)
print("rope q worst (s,h,i):", [tuple(int(x) for x in idx) for idx in (dq > 0.02).nonzero()[:8]])
for L in [64, 200, 4133]:
    nh, kvh = 16, 8
    qq, kk, vv = rnd((1, nh * d), 30), rnd((L, kvh * d), 31), rnd((L, kvh * d), 32)
    ref = _attn_ref(qq, kk, vv, nh, kvh, d, False, 0)[0]
    nm2 = Numerics("bf16", attn_probs_rounded=False)
    import oracle.qwen3 as oq2
    qq4 = qq.float().reshape(1, 1, nh, d).transpose(1, 2); kk4 = kk.float().reshape(1, L, kvh, d).transpose(1, 2); vv4 = vv.float().reshape(1, L, kvh, d).transpose(1, 2)
    ref_unr = oq2.eager_attention_forward(nm2, qq4, kk4, vv4, nh // kvh, None, oq2.attn_scale(nm2, d)).reshape(nh * d)
    got = ops.attn_decode(qq[0].contiguous().to(gpu), kk.to(gpu), vv.to(gpu), nh, kvh, d)
    stats(f"attn_decode L={L} vs oracle(P bf16)", got, ref)
    stats(f"attn_decode L={L} vs oracle(P f32)", got, ref_unr)
S = 150; nh = kvh = 4
qq, kk, vv = rnd((S, nh * d), 43), rnd((S, kvh * d), 44), rnd((S, kvh * d), 45)
ref = _attn_ref(qq, kk, vv, nh, kvh, d, False, 0)
got = ops.attn_prefill(qq.to(gpu), kk.to(gpu), vv.to(gpu), nh, kvh, d, 0, False)
stats("attn_prefill full S=150", got, ref)
ref = _attn_ref(qq, kk, vv, nh, kvh, d, True, 0)
got = ops.attn_prefill(qq.to(gpu), kk.to(gpu), vv.to(gpu), nh, kvh, d, 0, True)
stats("attn_prefill causal S=150", got, ref)
