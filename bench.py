#!/usr/bin/env python3
"""bench.py -- decode tokens/s (+ prefill tok/s) of the Qwen3-VL-8B hot path on MI355X, through the C ABI.

  python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one greedy decode token (one pass of the decoder stack + lm_head over the paged KV cache) of the
configuration BASELINE.json's metric is quoted on: Qwen3-VL-"7B" (= 8B) with one 1024x1024
image + a 512-token prompt.  Weights are synthetic (random init of that architecture, bf16, built directly in HBM);
inputs are synthetic and resident in HBM when the timed region starts.  Prefill is run once before the timed region
and reported as prefill_tok_s.  Decode stays single-GPU (BASELINE.json north_star): with N > 1 every rank serves its
own request (independent replicas, no data-path collective) => "scaling": "weak".

Extra objects on the JSON line: roofline (dominant kernel = the batch-1 weight-streaming matvec, HIP-event timed over
K further steps) and cpu_baseline (the oracle restatement timed on the host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TFLOPS = 2500.0  # bf16 dense
PMC_FILE = "r06_pmc_traffic_gemv.json"      # scripts/pmc_summary.py
STATS_FILE = "r06_cfg3_kernel_stats.md"     # scripts/stats_to_md.py of `rocprofv3 --kernel-trace --stats -- python bench.py ...`


METRICS = {
    "qwen3vl8b": "decode tokens/s (greedy, batch 1) -- Qwen3-VL-8B, 1x1024^2 image + 512-token prompt; prefill tok/s alongside",
    "qwen3vl8b-cfg5": "decode tokens/s (greedy, batch 1) -- Qwen3-VL-8B, 8 x 2048^2 images + 8192-token prompt (BASELINE cfg 5 on one GPU, S ~ 41k); prefill tok/s alongside",
    "qwen3vl8b-video": "decode tokens/s (greedy, batch 1) -- Qwen3-VL-8B, 16 video frames of 448^2 (8 temporal patches, 1568 video tokens) + 512-token prompt; prefill tok/s and the video patchify kernel alongside",
    "qwen3vl8b-text": "decode tokens/s (greedy, batch 1) -- Qwen3-VL-8B text stack, 1542-token prompt; prefill tok/s alongside",
    "qwen3-0.6b": "decode tokens/s (greedy, batch 1) -- Qwen3-0.6B, 2048-token prompt (BASELINE cfg 2); prefill tok/s alongside",
    "qwen3vl8b-cfg5-tp": "prefill tokens/s -- Qwen3-VL-8B, 8 x 2048^2 images + 8192-token prompt (BASELINE cfg 5), tensor-parallel decoder stack + image-parallel ViT over all ranks",
    "qwen3vl8b-cfg5-cp": "prefill tokens/s -- Qwen3-VL-8B, 8 x 2048^2 images + 8192-token prompt (BASELINE cfg 5), context-parallel decoder stack (full weights per GPU, one K/V all-gather per layer) + image-parallel ViT over all ranks",
    "qwen3-asr": "decode tokens/s (greedy, batch 1) -- Qwen3-ASR-0.6B, 30 s of 16 kHz audio (BASELINE cfg 4); prefill (log-mel + audio encoder + text) alongside",
}


def build_workload(name: str):
    from aha_amd import configs
    if name == "qwen3vl8b":
        return configs.qwen3vl_8b(), dict(image=1024, prompt=512)
    if name == "qwen3vl8b-cfg5":   # BASELINE.md section 4 cfg 5 on ONE GPU: 8 images of 2048^2 + 8192 text ids (S ~ 41k); prefill-dominated
        return configs.qwen3vl_8b(), dict(image=2048, prompt=8192, n_images=8)
    if name == "qwen3vl8b-video":   # the video path (not a BASELINE config): 16 sampled frames of 448^2 = 8 temporal patches x 784
        return configs.qwen3vl_8b(), dict(image=0, prompt=512, video=(16, 448, 448))   # patches -> 1568 video tokens + timestamps + 512 text ids
    if name == "qwen3vl8b-text":
        return configs.qwen3vl_8b(), dict(image=0, prompt=1542)
    if name == "qwen3-0.6b":
        return configs.qwen3_0_6b(), dict(image=0, prompt=2048)
    if name == "qwen3-asr":   # BASELINE.md section 4 cfg 4: 30 s of 16 kHz audio -> 390 audio tokens + template, decode 64
        return configs.qwen3_asr_0_6b(), dict(image=0, prompt=0, audio_samples=480000)
    if name == "tiny":
        return configs.tiny_qwen3(layers=2, hidden=512, heads=4, kv_heads=2, inter=1024, vocab=2048), dict(image=0, prompt=96)
    raise SystemExit(f"unknown workload {name}")


def decode_bytes_per_token(cfg, L):
    """layer weights + lm_head + KV read/write, bf16."""
    t = cfg.text if hasattr(cfg, "text") else cfg
    H, I = t.hidden_size, t.intermediate_size
    per_layer = (t.q_dim + 2 * t.kv_dim) * H + H * t.q_dim + 3 * I * H
    w = (t.num_hidden_layers * per_layer + t.vocab_size * H) * 2
    kv = t.num_hidden_layers * 2 * t.kv_dim * 2
    return w + kv * L + kv


MFMA_PEAK_BF16_TFLOPS = 2500.0   # dense bf16 (MI355X_MICROARCH.md; AMD's headline 5 PF includes 2:1 sparsity)


def prefill_flops(cfg, S, kv_offset=0, vit_segments=()):
    """Algorithmic FLOPs of one prefill of S prompt tokens (2 per multiply-add), the decoder stack + the vision tower:
      * text GEMMs: 2 S K N for qkv, o_proj, gate+up, down_proj of every layer; lm_head for ONE row (qwen3/model.rs:187: last position only);
      * text attention, CAUSAL-counted: row i sees kv_offset + i + 1 keys, QK^T + P.V = 4 d per (query, key, head);
      * ViT (one entry of `vit_segments` per image / video frame group = its patch count n): the four GEMMs of each block over all
        patches, FULL attention inside a segment (4 n^2 D per block), the patch-embed GEMM and the mergers' two GEMMs each.
    Element-wise work (norms, rope, softmax, activations) is not counted."""
    t = cfg.text if hasattr(cfg, "text") else cfg
    H, I, L = t.hidden_size, t.intermediate_size, t.num_hidden_layers
    gemm = 2.0 * S * H * (t.q_dim + 2 * t.kv_dim) + 2.0 * S * t.q_dim * H + 2.0 * S * H * 2 * I + 2.0 * S * I * H
    attn = 4.0 * t.q_dim * (S * kv_offset + S * (S + 1) / 2.0)
    out = {"text_gemm": L * gemm + 2.0 * t.vocab_size * H, "text_attention_causal": L * attn, "vit_gemm": 0.0, "vit_attention": 0.0}
    v = getattr(cfg, "vision", None)
    if v is not None and vit_segments:
        n = float(sum(vit_segments))
        D, VI = v.hidden_size, v.intermediate_size
        out["vit_gemm"] = v.depth * (2.0 * n * D * 3 * D + 2.0 * n * D * D + 4.0 * n * D * VI) + 2.0 * n * v.patch_dim * D
        m2 = v.spatial_merge_size ** 2
        n_mergers = 1 + len(v.deepstack_visual_indexes)
        out["vit_gemm"] += n_mergers * (2.0 * (n / m2) * (D * m2) * (D * m2) + 2.0 * (n / m2) * (D * m2) * v.out_hidden_size)
        out["vit_attention"] = v.depth * 4.0 * D * sum(float(s) * s for s in vit_segments)
    out["total"] = sum(out.values())
    return out


def cpu_baseline(cfg, sample_secs=20.0, kv_len=32, prompt_tokens=0, vit_patches=0):
    """Oracle restatement timed end to end on the host cores ("port"; the Candle reference cannot be built here -- BASELINE.md
    section 2), BOTH halves of BASELINE's metric on the same request the GPU line is quoted on (round-4 verdict, next-round item 6; the
    reference's definitions: `prompt_secs` = first forward + first sample, `completion_tps` = tokens / decode-loop seconds,
    /root/reference/src/models/common/generate.rs:123-158):

      * decode: the FULL-depth text stack (every layer its own weights in memory: layer i is layer 0's tensors rotated by a layer-
        dependent offset -- same statistics, distinct bytes, a memcpy instead of 15 GB of randn) decodes greedy tokens over a cache of
        `kv_len` tokens (= the GPU line's `kv_len_mid`; the cache holds synthetic bf16 K / V -- a decode step's cost depends on the cache's
        length, not on its values -- and is re-concatenated every step as the reference does, modules.rs:558-566) for about half of
        `sample_secs`; value = decoded tokens / seconds.
      * prefill (`prefill_tok_s`): SAMPLED and EXTRAPOLATED, labelled as such -- two text layers at S = `prompt_tokens` with the causal
        mask, one ViT block at `vit_patches` patches (full attention), the final norm + lm_head on the last row, each timed once; prefill
        seconds = depth x (text layer) + ViT depth x (ViT block) + head.  Left out (small against those): patch embed, position
        embeddings, the mergers, DeepStack adds, get_rope_index.  The whole oracle prefill of this request costs 69 s at full depth
        (profiles/r04_parity_fullsize.json) -- too long for the default bench run.

    Falls back to one layer x depth for the decode half only if the host cannot hold the full stack."""
    import copy
    import torch
    from aha_amd.weights import qwen3_text_weights
    from oracle.numerics import Numerics
    from oracle import qwen3 as oq
    from oracle.qwen3 import OracleQwen3
    t = cfg.text if hasattr(cfg, "text") else cfg
    # 256 logical CPUs on the GPU box: torch's intra-op pool thrashes beyond a few dozen threads on these skinny
    # mat-vecs (measured: 1.9 s/layer at 256 threads vs 26 ms at 16-64), so the port is timed on <= 32 threads.
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 32))
    torch.set_num_threads(cores)
    one = copy.deepcopy(t)
    one.num_hidden_layers = 1
    one.mrope_section = None
    one.tie_word_embeddings = True   # lm_head timed through the (tied) embedding: same shape
    w1 = qwen3_text_weights(one, seed=0)
    depth, full = t.num_hidden_layers, None
    try:
        full_cfg = copy.deepcopy(one)
        full_cfg.num_hidden_layers = depth
        w = {k: v for k, v in w1.items() if ".layers." not in k}
        for i in range(depth):
            for k, v in w1.items():
                if ".layers.0." in k:
                    w[k.replace(".layers.0.", f".layers.{i}.")] = v if i == 0 else torch.roll(v.flatten(), 7919 * i + 1).view_as(v)
        full = OracleQwen3(full_cfg, w, Numerics("bf16"))
    except (MemoryError, RuntimeError):
        full = None
    kv_len = max(int(kv_len), 1)
    g = torch.Generator().manual_seed(0)

    def synthetic_cache(o, n_layers):   # bf16-valued K / V of kv_len tokens per layer
        for li in range(n_layers):
            o.kv[li] = tuple(torch.randn(1, t.num_key_value_heads, kv_len, t.head_dim, generator=g).bfloat16().float() for _ in range(2))

    out = {"unit": "tokens/s", "cores": cores, "kind": "port", "decode_kv_len": kv_len}
    if full is not None:
        synthetic_cache(full, depth)
        logits = full.forward([5], kv_len)       # one untimed step (page-in of the weights)
        n, pos = 0, kv_len + 1
        t_start = time.perf_counter()
        while n < 4 or (time.perf_counter() - t_start < 0.5 * sample_secs and n < 64):
            tok = int(torch.argmax(logits.reshape(-1, logits.shape[-1])[-1]))
            logits = full.forward([tok], pos)
            n, pos = n + 1, pos + 1
        secs = time.perf_counter() - t_start
        out["value"] = round(n / secs, 3)
        out["sample"] = (f"oracle restatement (torch-CPU, bf16 rounding points), {n} greedy decode steps of the full {depth}-layer text stack "
                         f"+ lm_head over a {kv_len}-token cache (synthetic K / V, re-concatenated per step as the reference does), timed end "
                         f"to end ({secs:.1f} s); Candle CPU reference not buildable here")
        full.clear_cache()
        o_text = full
    else:
        o = OracleQwen3(one, w1, Numerics("bf16"))
        synthetic_cache(o, 1)
        n, t_layer, t_head = 0, 0.0, 0.0
        t_start = time.perf_counter()
        pos = kv_len
        while time.perf_counter() - t_start < 0.5 * sample_secs and n < 64:
            t0 = time.perf_counter()
            h = o.forward_hidden([5], None, pos)
            t1 = time.perf_counter()
            o.nm.linear(h, o.lm_head)
            t2 = time.perf_counter()
            t_layer += t1 - t0
            t_head += t2 - t1
            n += 1
            pos += 1
        per_tok = depth * (t_layer / n) + t_head / n
        out["value"] = round(1.0 / per_tok, 3)
        out["sample"] = (f"oracle restatement (torch-CPU, bf16 rounding points), {n} decode steps of 1 of {depth} layers at full width + "
                         f"lm_head over a {kv_len}-token cache, extrapolated x{depth} (the host could not hold the full stack); Candle CPU "
                         "reference not buildable here")
        o.clear_cache()
        o_text = o
    # ---- the prefill half: sampled layers, extrapolated over the depth ----
    if prompt_tokens > 1:
        S = int(prompt_tokens)
        nm = o_text.nm
        nm.attn_row_block = 512   # the causal mask built per row block (the same arithmetic per row; no (32, S, S) score tensor at once)
        ids = torch.randint(0, one.vocab_size, (S,), generator=g).tolist()
        x = o_text.embed_tokens(ids)
        cos, sin = oq.rope_cos_sin(o_text.inv_freq, 0, S)
        n_l = min(2, o_text.cfg.num_hidden_layers)
        t0 = time.perf_counter()
        for li in range(n_l):
            x = o_text.decoder_layer(li, x, cos, sin, "causal" if S > nm.attn_row_block else oq.prepare_causal_attention_mask(S))
        t_layer = (time.perf_counter() - t0) / n_l
        t0 = time.perf_counter()
        h = oq.rms_norm(nm, x[:, S - 1:S], o_text.w[o_text.p + "norm.weight"], o_text.cfg.rms_norm_eps)
        nm.linear(h, o_text.lm_head)
        t_head = time.perf_counter() - t0
        o_text.clear_cache()
        nm.attn_row_block = 0
        t_vit, vit_depth = 0.0, 0
        v = getattr(cfg, "vision", None)
        if v is not None and vit_patches > 0:
            from aha_amd.weights import qwen3vl_vision_weights
            from oracle.qwen3vl import OracleVision
            vcfg = copy.deepcopy(cfg)
            vcfg.vision = copy.deepcopy(v)
            vit_depth = v.depth
            vcfg.vision.depth = 1
            vcfg.vision.deepstack_visual_indexes = []
            ov_ = OracleVision(vcfg, qwen3vl_vision_weights(vcfg, seed=100), Numerics("bf16", attn_row_block=1024))
            N = int(vit_patches)
            xv = torch.randn(N, v.hidden_size, generator=g).bfloat16().float()
            ang = torch.rand(N, v.head_dim, generator=g) * 6.0
            t0 = time.perf_counter()
            ov_.block(0, xv, ang.cos(), ang.sin(), [0, N])
            t_vit = time.perf_counter() - t0
        pre = depth * t_layer + vit_depth * t_vit + t_head
        out["prefill_tok_s"] = round(S / pre, 2)
        out["prefill_seconds_extrapolated"] = round(pre, 2)
        out["prefill_sample"] = (f"EXTRAPOLATED from sampled layers: {n_l} text layers at S = {S} with the causal mask ({t_layer:.2f} s each) x {depth}"
                                 + (f" + one ViT block at {vit_patches} patches ({t_vit:.2f} s) x {vit_depth}" if vit_depth else "")
                                 + f" + final norm and lm_head on the last row ({t_head:.3f} s); patch embed, position embeddings, mergers and "
                                 "DeepStack adds not included; prefill_tok_s = prompt tokens / those seconds (the reference's prompt_secs, "
                                 "generate.rs:123-135, without the sampler)")
    return out


def sharded_prefill_bench(rank, world, local_rank, n_images=8, image_px=2048, prompt=8192, repeats=1, mode="tp"):
    """BASELINE cfg 5 through the sharded path, the ViT image-parallel with one all-gather and the decoder stack
    either as one tensor-parallel group over all ranks (mode "tp": RCCL reduce-scatter / all-gather over xGMI inside the library, KV
    gathered to rank 0 afterwards) or context-parallel (mode "cp": full weights on every rank, the prompt's rows sharded, one K / V
    all-gather per layer, every rank ends with the whole cache).  Strong scaling: the request is the same for every N, value = prompt
    tokens / slowest rank's prefill seconds.  N = 1 is the single-GPU cfg 5 prefill."""
    import torch
    from aha_amd import configs, parallel
    from aha_amd import weights as W
    from aha_amd.vision_host import synthetic_image_request
    cfg = configs.qwen3vl_8b()
    dev = f"cuda:{local_rank}"
    w = W.qwen3vl_weights(cfg, seed=0, device=dev)            # the SAME checkpoint on every rank; the library keeps its shard
    g = torch.Generator().manual_seed(5)
    ids, data = synthetic_image_request(cfg, image_px, prompt, g, device=dev, n_images=n_images)
    torch.cuda.synchronize()
    phases = {}
    tok, secs, model = parallel.sharded_prefill(cfg, w, ids, data, rank, world, local_rank, kv_reserve_tokens=len(ids) + 4096,
                                                repeats=repeats, phases_out=phases, mode=mode)
    # BASELINE cfg 5 is "prefill + 16 tokens", and decode stays single-GPU (north_star): the head-sharded KV cache is gathered into an
    # un-sharded model on rank 0 (parallel.gather_kv_to_rank0: one RCCL gather of the packed pages), which decodes the 16 tokens alone
    from aha_amd.model import HipInferenceModel
    handback = {}
    full = model
    if world > 1 and mode == "tp":
        full = HipInferenceModel(cfg, w, device=local_rank, kv_reserve_tokens=len(ids) + 4096) if rank == 0 else None
        import torch.distributed as dist
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        parallel.gather_kv_to_rank0(model, full, rank, world)
        torch.cuda.synchronize()
        dist.barrier()
        handback["kv_handback_s"] = round(time.perf_counter() - t0, 4)
        phases["kv_gather_s"] = handback["kv_handback_s"]
    if rank == 0:
        t0 = time.perf_counter()
        out = full.decode_greedy(tok, len(ids), 16)
        handback["decode_16_after_prefill_s"] = round(time.perf_counter() - t0, 4)
        handback["decode_tokens"] = len(out)
        if full is not model:
            full.close()
    del w
    value, worst = parallel.aggregate_throughput(float(len(ids)) / world, secs, device=dev)   # sum of shares / max seconds
    toks = torch.tensor([tok], dtype=torch.int64, device=dev)
    same = True
    if world > 1:
        import torch.distributed as dist
        lst = [torch.zeros_like(toks) for _ in range(world)]
        dist.all_gather(lst, toks)
        same = all(int(t.item()) == int(lst[0].item()) for t in lst)
    model.close()
    torch.cuda.empty_cache()
    # per-phase seconds: rank 0's own diagnosis + the slowest rank's figure for every phase (one MAX all-reduce over the value vector)
    keys = sorted(k for k, v in phases.items() if isinstance(v, float))
    phases_max = {}
    if world > 1 and keys:
        import torch.distributed as dist
        allk = [None] * world
        dist.all_gather_object(allk, keys)
        keys = sorted(set().union(*[set(k) for k in allk]))
        vec = torch.tensor([float(phases.get(k, 0.0)) for k in keys], dtype=torch.float64, device=dev)
        dist.all_reduce(vec, op=dist.ReduceOp.MAX)
        phases_max = {k: round(float(v), 5) for k, v in zip(keys, vec.tolist())}
    par = (f"tp{world} (sequence-parallel: RCCL reduce-scatter of the row-parallel partial sums in column blocks and all-gather of "
           "the normalised rows in row chunks, both overlapped with the GEMMs) + image-parallel ViT (all-gather) + KV gather to rank 0 for decode") if mode == "tp" else \
          (f"cp{world} (context-parallel: full weights on every GPU, two zigzag row chunks of the prompt per rank, one RCCL all-gather of the layer's K / V "
           "pages per layer; every rank ends with the whole KV cache, rank 0 decodes) + image-parallel ViT (all-gather)")
    return {"metric": METRICS["qwen3vl8b-cfg5-" + mode], "value": round(value, 1), "unit": "tokens/s", "prefill_s": round(worst, 4),
            "prompt_tokens": len(ids), "n_images": n_images, "image": image_px, "scaling": "strong",
            "parallelism": par,
            "rccl_ranks": world, "first_token_equal_on_all_ranks": same, **handback,
            "phases_rank0": phases, "phases_max_over_ranks": phases_max,
            "phases_note": "one extra UNTIMED prefill with the library profiler on (HIP events per launch group on the model's stream; host-side phases "
                           "bracketed by device synchronisation): vit_s / embeds_all_gather_s / stack_s are wall clock, the rest event time inside stack_s; "
                           "reduce_scatter_wait_s / all_gather_wait_s = what the compute stream waited for the collectives overlapped on the communication stream; "
                           "kv_all_gather_s = the context-parallel form's per-layer K / V exchange (pack + all-gather + unpack, on the compute stream)"}

SHARDED_LEG_TIMEOUT_S = int(os.environ.get("AHA_BENCH_SHARDED_TIMEOUT_S", "420"))


def guarded(fn, seconds, on_timeout):
    """fn() on the calling thread -> (result, None), or (None, "<exception text>") if it raised.  If it has not returned after `seconds`,
    a timer thread calls on_timeout() and ends the PROCESS with os._exit(0): a hung collective cannot be interrupted any other way, and
    the launcher only returns when every rank has exited."""
    import threading
    done = threading.Event()

    def watchdog():
        if not done.wait(seconds):
            try:
                on_timeout()
            finally:
                sys.stdout.flush()
                os._exit(0)

    threading.Thread(target=watchdog, daemon=True).start()
    try:
        return fn(), None
    except BaseException as e:   # noqa: BLE001 -- including SystemExit from an assert inside a library call
        return None, f"{type(e).__name__}: {e}"[:400]
    finally:
        done.set()


SHARDED_KEYS = {"cp": ("sharded_prefill", "sharded_ok"), "tp": ("sharded_prefill_tp", "sharded_tp_ok")}


def run_sharded_legs(order, run_leg, rank, world, gpus, line, timeout_s, agree=None):
    """The sharded forms of BASELINE cfg 5's prefill, one leg per entry of `order` ("tp" = ONE tensor-parallel group over all ranks, the
    form cfg 5 names: RCCL reduce-scatter / all-gather, KV handed back to rank 0; "cp" = context-parallel: full weights per rank, the
    prompt's rows sharded, one K / V all-gather per layer).  Each leg runs under its own watchdog, writes its own object
    (`sharded_prefill_tp` / `sharded_prefill`) and its own top-level flag (`sharded_tp_ok` / `sharded_ok`) into `line` on rank 0, builds its
    own model and communicator (run_leg(mode)), and -- round-4 verdict, item 4c -- an EXCEPTION in one leg does not cost the next one its
    measurement: before a later leg every rank passes `agree()` (a barrier on the launcher's process group) INSIDE that leg's watchdog, so a
    rank that is still stuck in the failed leg's collective ends the run through the watchdog instead of deadlocking the next leg.  A hang
    ends the process (os._exit from the watchdog after rank 0 printed what it has).  Returns True when every leg that ran was clean."""
    all_clean = True
    ran = 0
    for mode in order:
        if mode not in SHARDED_KEYS:
            raise SystemExit(f"--sharded-order: unknown leg {mode!r} (cp, tp)")
        if mode == "tp" and world <= 1:
            continue                      # one rank: the tensor-parallel form is the single-GPU prefill the context-parallel leg already times
        key, okkey = SHARDED_KEYS[mode]

        def on_timeout(key=key, okkey=okkey):
            if rank == 0:
                line[key] = {"error": f"did not finish within {timeout_s} s (watchdog); everything above this object is complete"}
                line[okkey] = False
                print(json.dumps(line), flush=True)

        def leg(mode=mode, first=(ran == 0)):
            if not first and agree is not None:
                agree()
            return run_leg(mode)

        sp, err = guarded(leg, timeout_s, on_timeout)
        ran += 1
        if err is None and sp.get("rccl_ranks") != gpus:
            err = f"rccl_ranks {sp.get('rccl_ranks')} != --gpus {gpus}"
        if rank == 0:
            line[key] = sp if err is None else {"error": err}
            # top level, next to the replica numbers: the process exits 0 either way (the replica line must reach the driver), so a
            # consumer that reads only the exit code would otherwise see success after a failed or hung sharded leg
            line[okkey] = bool(err is None and sp["first_token_equal_on_all_ranks"])
        all_clean = all_clean and err is None
    return all_clean


def self_launch(n: int) -> int:
    """Re-executes this command under torch.distributed.run with n ranks on 127.0.0.1 (a free port) and returns its exit code."""
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    return subprocess.call(cmd, env=env)


def launcher_selftest(args) -> None:
    """`--workload launcher-selftest` (CPU tier, tests/test_bench_cpu.py): everything of a multi-rank run EXCEPT the GPU work -- the
    rendezvous bench.py --gpus N sets up, the barrier / max-over-ranks / sum-of-units aggregation, one JSON line from rank 0 --
    over gloo, with made-up per-rank timings.  The line says what it is; it is not a measurement."""
    from aha_amd import parallel
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dist = parallel.init_process_group("gloo") if world > 1 else None
    secs = 0.5 + 0.25 * rank
    value, worst = parallel.aggregate_throughput(float(args.steps), secs)
    if rank == 0:
        print(json.dumps({"metric": "launcher self-test (no GPU work)", "value": round(value, 3), "unit": "tokens/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * worst / args.steps, 4),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "none",
                          "config": {"workload": "launcher-selftest", "replicas": world}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def source_digest(files) -> str:
    """sha256 over the named kernel sources: profiles taken with OTHER sources must not be attached to a fresh measurement."""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "aha_amd", "csrc", f), "rb") as fh:
            h.update(f.encode())
            h.update(fh.read())
    return h.hexdigest()[:16]


GEMV_SOURCES = ("gemv_body.h", "kernels_gemv.hip", "common.h")

def attach_committed_profiles(roof: dict, workload: str, profiles_dir: str = None) -> dict:
    """Numbers that cannot be collected inside the bench process (rocprofv3 kernel-only durations, PMC traffic) come from the committed
    summaries under profiles/ -- attached ONLY when they were taken with the same matvec sources (the digest scripts/stats_to_md.py /
    scripts/pmc_summary.py stamp into them), so a kernel change without a profile refresh leaves them out instead of quoting stale
    numbers next to a fresh ms_per_step (round-2 verdict, weak #7)."""
    pdir = profiles_dir or os.path.join(ROOT, "profiles")
    dig = source_digest(GEMV_SOURCES)
    roof["gemv_source_digest"] = dig
    try:
        with open(os.path.join(pdir, PMC_FILE)) as f:
            pmc = json.load(f)
        if pmc.get("workload") == workload and pmc.get("gemv_source_digest") == dig:
            roof["traffic"] = round(pmc["traffic_bytes_per_launch"])
            roof["traffic_source"] = "profiles/" + PMC_FILE
    except (OSError, KeyError, ValueError):
        pass
    try:
        if workload == "qwen3vl8b":
            tot_us, calls, stamped = 0.0, 0, None
            with open(os.path.join(pdir, STATS_FILE)) as f:
                for line in f:
                    if "gemv_source_digest:" in line:
                        stamped = line.split("gemv_source_digest:")[1].split()[0]
                    c = [x.strip() for x in line.split("|")]
                    if len(c) >= 6 and "gemv_kernel" in c[1]:
                        calls += int(c[2])
                        tot_us += float(c[3])
            if calls and stamped == dig:
                avg = tot_us / calls
                roof["rocprof"] = {"avg_us": round(avg, 2), "achieved": round(roof["algorithmic_bytes_per_launch"] / avg / 1e3, 1),
                                   "frac": round(roof["algorithmic_bytes_per_launch"] / avg / 1e3 / HBM_PEAK_GBS, 4),
                                   "source": "profiles/" + STATS_FILE}
    except (OSError, ValueError):
        pass
    return roof



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default=os.environ.get("AHA_BENCH_WORKLOAD", "qwen3vl8b"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prompt", type=int, default=0, help="override the workload's text prompt length (diagnostics; the JSON names it)")
    ap.add_argument("--host-loop", action="store_true", help="drive decode with forward_step (one host sync per token)")
    ap.add_argument("--sharded-order", default="tp,cp",
                    help="order of the sharded cfg 5 legs of a multi-rank run: tp = the tensor-parallel form BASELINE cfg 5 names, cp = context-parallel")
    ap.add_argument("--sharded-prefill", action="store_true",
                    help="also run BASELINE cfg 5 through the TP + image-parallel path (always done when WORLD_SIZE > 1)")
    args = ap.parse_args()

    # --gpus N without a launcher (the driver runs `python bench.py --gpus N ...`): start the N ranks ourselves, one process per
    # GPU, exactly as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` would.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; they must agree")
    if args.workload == "launcher-selftest":
        return launcher_selftest(args)

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    shared_device = world > ndev          # more ranks than GPUs (tests on a 1-GPU box): ranks share devices, no RCCL (one rank per
    if shared_device:                     # device is an RCCL requirement) -- the replicas still run, collectives go over gloo
        local_rank = local_rank % max(ndev, 1)
    torch.cuda.set_device(local_rank)
    from aha_amd import parallel
    if world > 1:
        dist = parallel.init_process_group("gloo" if shared_device else "nccl", torch.device(f"cuda:{local_rank}"))

    import __graft_entry__
    __graft_entry__.build()
    if args.workload in ("qwen3vl8b-cfg5-tp", "qwen3vl8b-cfg5-cp"):   # the sharded path as the metric itself (strong scaling over --gpus)
        if shared_device:
            raise SystemExit(args.workload + " needs one GPU per rank (RCCL)")
        sp = sharded_prefill_bench(rank, world, local_rank, repeats=max(1, min(args.steps, 3)), mode=args.workload[-2:])
        if rank == 0:
            line = {"metric": sp["metric"], "value": sp["value"], "unit": "tokens/s", "n_gpus": world, "steps": max(1, min(args.steps, 3)),
                    "warmup": 1, "ms_per_step": round(1e3 * sp["prefill_s"], 2), "higher_is_better": True, "scaling": "strong",
                    "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                    "config": {"workload": args.workload, "prompt_tokens": sp["prompt_tokens"], "image": sp["image"],
                               "n_images": sp["n_images"], "parallelism": sp["parallelism"], "rccl_ranks": world},
                    "first_token_equal_on_all_ranks": sp["first_token_equal_on_all_ranks"]}
            for k in ("kv_handback_s", "decode_16_after_prefill_s", "decode_tokens", "phases_rank0", "phases_max_over_ranks", "phases_note"):
                if k in sp:
                    line[k] = sp[k]
            line["sharded_ok"] = bool(sp["first_token_equal_on_all_ranks"])
            print(json.dumps(line), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    from aha_amd import weights as W
    from aha_amd.model import HipInferenceModel, MultiModalData

    cfg, wl = build_workload(args.workload)
    if args.prompt > 0:
        wl = dict(wl, prompt=args.prompt)
    is_asr = hasattr(cfg, "audio")
    is_vl = hasattr(cfg, "text") and not is_asr
    tcfg = cfg.text if hasattr(cfg, "text") else cfg
    dev = f"cuda:{local_rank}"
    t0 = time.perf_counter()
    if is_asr:
        w = W.qwen3_asr_weights(cfg, seed=rank, device=dev)
    elif is_vl:
        w = W.qwen3vl_weights(cfg, seed=rank, device=dev) if (wl["image"] or wl.get("video")) else \
            W.qwen3_text_weights(tcfg, seed=rank, prefix="model.language_model.", device=dev)
    else:
        w = W.qwen3_text_weights(tcfg, seed=rank, device=dev)
    torch.cuda.synchronize()
    model = HipInferenceModel(cfg, w, device=local_rank, kv_reserve_tokens=4096 if wl.get("n_images", 1) == 1 else 45056)
    del w
    torch.cuda.empty_cache()
    t_load = time.perf_counter() - t0

    # synthetic request (cfg 3): template + <|vision_start|> + image pads + <|vision_end|> + text
    g = torch.Generator().manual_seed(3 + rank)
    data = None
    if wl["image"]:
        from aha_amd.vision_host import synthetic_image_request
        ids, data = synthetic_image_request(cfg, wl["image"], wl["prompt"], g, n_images=wl.get("n_images", 1))
    elif wl.get("video"):
        # frames as get_video_data hands them over (RGB24 at the resize size); patchified on the GPU (process_videos), prompt in the
        # processor's layout: per temporal patch 3 stand-in timestamp ids, <|vision_start|>, h*w/4 video pads, <|vision_end|>
        from aha_amd import ops, vision_host
        T, VH, VW = wl["video"]
        frames = torch.randint(0, 256, (T, VH, VW, 3), generator=g, dtype=torch.uint8).to(dev)
        pvv, vgrid = vision_host.process_videos([frames], cfg)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            ops.video_to_patches(frames, cfg.vision.patch_size, cfg.vision.spatial_merge_size)
        e1.record()
        torch.cuda.synchronize()
        video_patchify = {"us": round(e0.elapsed_time(e1) * 1e3 / 20, 2),
                          "GBs": round((frames.numel() + pvv.numel() * 2) / (e0.elapsed_time(e1) * 1e-3 / 20) / 1e9, 1),
                          "algorithmic_bytes": int(frames.numel() + pvv.numel() * 2)}
        hi = min(tcfg.vocab_size, 151643)
        stamps = [torch.randint(0, hi, (3,), generator=g).tolist() for _ in range(int(vgrid[:, 0].sum()))]
        ids = torch.randint(0, hi, (4,), generator=g).tolist() + vision_host.video_prompt_ids(cfg, vgrid, stamps) + \
            torch.randint(0, hi, (wl["prompt"],), generator=g).tolist()
        data = MultiModalData(pixel_values_video=pvv, video_grid_thw=vgrid)
    elif is_asr:
        # seed-4 N(0, 0.1^2) clipped to [-1, 1]; the library computes the log-mel features on the GPU from the raw samples
        n = wl["audio_samples"]
        wave = np.clip(np.random.default_rng(4 + rank).normal(0, 0.1, n), -1, 1).astype(np.float32)
        from aha_amd.audio_host import audio_prompt_ids
        hi = min(tcfg.vocab_size, 151643)
        pre = torch.randint(0, hi, (9,), generator=g).tolist()     # <|im_start|>system ... user\n
        post = torch.randint(0, hi, (5,), generator=g).tolist()    # <|im_end|>\n<|im_start|>assistant\n
        ids = audio_prompt_ids(cfg, n, pre, post)
        data = MultiModalData(audio_samples=wave)
    else:
        ids = torch.randint(0, min(tcfg.vocab_size, 151643), (wl["prompt"],), generator=g).tolist()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- prefill (reported, outside the K timed steps) ----
    model.forward_initial(ids, 0, data, want_logits=False)  # warm
    t_prefills = []
    for _ in range(3 if len(ids) < 16384 else 1):           # median of 3 (one sample for the 41k-token workloads: seconds each)
        model.clear_cache()
        barrier()
        t0 = time.perf_counter()
        _, tok = model.forward_initial(ids, 0, data, want_logits=False)
        t_prefills.append(time.perf_counter() - t0)
    t_prefill = sorted(t_prefills)[len(t_prefills) // 2]
    off = len(ids)

    # ---- decode: W warmup steps, then exactly K timed steps ----
    def run_steps(n, tok, off):
        if args.host_loop:
            for _ in range(n):
                _, tok = model.forward_step(tok, off, want_logits=False)
                off += 1
            return tok, off
        out = model.decode_greedy(tok, off, n)
        assert len(out) == n
        return out[-1], off + n

    tok, off = run_steps(args.warmup, tok, off)
    barrier()
    t0 = time.perf_counter()
    tok, off = run_steps(args.steps, tok, off)
    barrier()
    dt = time.perf_counter() - t0
    job_value, dt = parallel.aggregate_throughput(float(args.steps), dt, device=dev)  # sum of tokens / max over ranks
    kv_mid = off - args.steps // 2

    # ---- roofline of the dominant kernel (weight-streaming matvec), HIP events on the model's stream ----
    model.set_profiling(True)
    tok, off = run_steps(args.steps, tok, off)
    prof = {k: model.get_profile(k) for k in ("gemv", "attn_decode", "elem", "argmax")}
    model.set_profiling(False)
    # dominant kernel class of a decode step: the weight-streaming matvec (all projections + lm_head)
    gv = prof["gemv"]
    achieved = gv["bytes"] / (gv["ms"] * 1e-3) / 1e9 if gv["ms"] > 0 else 0.0
    roof = {"bound": "hbm",
            "kernel": "gemv_kernel (batch-1 weight streaming, all projections + lm_head)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "launches": gv["launches"], "avg_us": round(1e3 * gv["ms"] / max(gv["launches"], 1), 2),
            "algorithmic_bytes_per_launch": round(gv["bytes"] / max(gv["launches"], 1)),
            "timing": "HIP event pairs on the model's stream around every launch of the class, collected in THIS run (a pair also sees "
                      "the ~2.5 us of dispatch latency in front of a kernel: the rocprofv3 kernel-only figure, when attached, is the higher one)"}
    attach_committed_profiles(roof, args.workload)
    ad = prof["attn_decode"]
    attn_gbs = ad["bytes"] / (ad["ms"] * 1e-3) / 1e9 if ad["ms"] > 0 else 0.0

    if rank == 0:
        # the other half of BASELINE's metric: the prefill against the dense bf16 MFMA peak (attention counted causally; stated)
        segs = []
        if data is not None and getattr(data, "image_grid_thw", None) is not None:
            segs += [int(g[1]) * int(g[2]) for g in np.asarray(data.image_grid_thw).reshape(-1, 3) for _ in range(int(g[0]))]
        if data is not None and getattr(data, "video_grid_thw", None) is not None:
            segs += [int(g[1]) * int(g[2]) for g in np.asarray(data.video_grid_thw).reshape(-1, 3) for _ in range(int(g[0]))]
        pf = prefill_flops(cfg, len(ids), 0, segs if is_vl else ())
        ach = pf["total"] / t_prefill / 1e12
        roof_prefill = {"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / MFMA_PEAK_BF16_TFLOPS, 4), "flops": {k: round(v / 1e12, 3) for k, v in pf.items()},
                        "flops_unit": "TFLOP per prefill", "seconds": round(t_prefill, 5),
                        "counting": "2 FLOP per multiply-add; text attention causal (row i: i + 1 keys), ViT attention full within an image; "
                                    "lm_head for the last position only; element-wise work not counted; whole prefill wall clock (ViT + "
                                    "36 layers + lm_head + host-side position / page bookkeeping) in the denominator"}
        step_bytes = decode_bytes_per_token(cfg, kv_mid)
        line = {
            "metric": METRICS.get(args.workload, "decode tokens/s (greedy, batch 1) -- " + args.workload),
            "value": round(job_value, 3), "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": args.workload, "prompt_tokens": len(ids), "image": wl["image"],
                       "kv_len_mid": kv_mid, "replicas": world, "collective_backend": ("gloo (ranks share a GPU)" if shared_device else "rccl") if world > 1 else None,
                       "loop": "host forward_step" if args.host_loop else "device-resident greedy loop"},
            "prefill_tok_s": round(len(ids) / t_prefill, 1), "prefill_ms": round(1e3 * t_prefill, 2),
            "roofline_prefill": roof_prefill,
            "prefill_ms_samples": [round(1e3 * t, 2) for t in t_prefills],
            "decode_step_hbm_frac": round(step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
            "attn_decode_GBs": round(attn_gbs, 1),
            "load_s": round(t_load, 1),
            "roofline": roof,
        }
        if wl.get("video"):
            line["config"]["video_frames_hw"] = list(wl["video"])
            line["video_to_patches"] = video_patchify
        if not args.no_cpu_baseline and world == 1:          # rank 0 at N = 1 only (torchrun also pins its ranks to one OMP thread)
            vit_n = sum(segs) if is_vl else 0
            line["cpu_baseline"] = cpu_baseline(cfg, kv_len=kv_mid, prompt_tokens=len(ids), vit_patches=vit_n)
    model.close()
    torch.cuda.empty_cache()
    # The part of the path that SHARDS (north_star: long-context prefill + ViT over the node's GPUs): measured next to the
    # replica decode number whenever there is more than one rank, so a scaling run of this command covers both.
    clean = True
    if (world > 1 or args.sharded_prefill) and args.workload == "qwen3vl8b" and not shared_device:
        # These legs have never run on more than one GPU (no multi-GPU box in the build loop): the replica decode numbers above must reach
        # the driver whatever happens in them (run_sharded_legs: an exception becomes an "error" entry and the NEXT leg still runs; a hang
        # is ended by a watchdog on every rank).
        def agree():
            if world > 1:
                dist.barrier()
        order = [x for x in args.sharded_order.split(",") if x]
        clean = run_sharded_legs(order, lambda mode: sharded_prefill_bench(rank, world, local_rank, mode=mode), rank, world, args.gpus, line,
                                 SHARDED_LEG_TIMEOUT_S, agree)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        if clean:
            dist.barrier()
            dist.destroy_process_group()
        else:
            sys.stdout.flush()
            os._exit(0)    # the other ranks may be inside a collective this rank left: no barrier, no teardown


if __name__ == "__main__":
    main()
