"""Same-box A/B of the prefill attention's kernel forms (text geometry: 32 q heads, 8 kv heads, head_dim 128): 16 = 16 q rows per wave
(csrc/kernels_attn.hip), 64 = one wave per SIMD with 64 rows per wave (csrc/kernels_attn64.hip), 65 = the same pipelined inside the wave.
AHA_ATTN_TIME=<reps> makes the C ABI op time the kernel with HIP events; forms alternate A B C A B C so that box drift shows.
    python scripts/attn64_ab.py [S:causal ...]        e.g. 8192:0 8192:1 40980:1
Also prints the largest difference between the forms' outputs in units of the output's rms."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("AHA_ATTN_TIME", "5")
import torch
from aha_amd import ops, build
build.build()
dev = torch.device("cuda:0")
nh, kvh, d = 32, 8, 128
shapes = [(8192, False), (8192, True), (2048, True), (1542, True), (4096, False)]
if len(sys.argv) > 1:
    shapes = [(int(x.split(":")[0]), bool(int(x.split(":")[1]))) for x in sys.argv[1:]]
forms = [int(x) for x in os.environ.get("FORMS", "16,64,65").split(",")]
for S, causal in shapes:
    g = torch.Generator(device=dev).manual_seed(S)
    q = torch.randn(S, nh * d, device=dev, dtype=torch.bfloat16, generator=g)
    k = torch.randn(S, kvh * d, device=dev, dtype=torch.bfloat16, generator=g)
    v = torch.randn(S, kvh * d, device=dev, dtype=torch.bfloat16, generator=g)
    fl = 4.0 * S * S * d * nh * (0.5 if causal else 1.0)
    print(f"S={S} causal={causal}: {fl/1e12:.3f} TFLOP nominal", flush=True)
    outs = {}
    for rep in range(2):
        for form in forms:
            ops.attn_form(form)
            print(f"  form {form}:", end=" ", flush=True)
            sys.stderr.flush()
            outs[form] = ops.attn_prefill(q, k, v, nh, kvh, d, causal=causal).float()
            torch.cuda.synchronize()
    ops.attn_form(-1)
    rms = float(outs[forms[0]].pow(2).mean().sqrt())
    for form in forms[1:]:
        dlt = (outs[form] - outs[forms[0]]).abs()
        print(f"  form {form} vs {forms[0]}: max |diff| {float(dlt.max())/rms:.4f} rms, mean {float(dlt.mean())/rms:.5f} rms, finite {bool(torch.isfinite(outs[form]).all())}", flush=True)
