"""bench.py's host-side pieces that run without a GPU: the CPU baseline leg (the oracle restatement timed end to end on a full-depth
stack) and the algorithmic byte count of a decode step."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_cpu_baseline_times_the_full_depth_stack():
    import bench
    from aha_amd import configs
    cfg = configs.tiny_qwen3(layers=3, hidden=256, heads=4, kv_heads=2, inter=512, vocab=1024)
    r = bench.cpu_baseline(cfg, sample_secs=0.5)
    assert r["kind"] == "port" and r["unit"] == "tokens/s" and r["value"] > 0 and r["cores"] >= 1
    assert "full 3-layer" in r["sample"] and "end to end" in r["sample"], r["sample"]   # not the one-layer extrapolation
    assert "prefill_tok_s" not in r                     # no prompt named: the decode half alone


def test_cpu_baseline_has_a_prefill_half_at_the_requests_shape():
    """BASELINE's metric is decode tokens/s AND prefill tokens/s (the reference's prompt_secs / completion_tps,
    /root/reference/src/models/common/generate.rs:123-158): the baseline object carries both, the decode half over a cache of the GPU
    line's kv_len_mid, the prefill half sampled (text layers at the prompt length, one ViT block at the image's patch count) and
    extrapolated over the depth -- and says so."""
    import bench
    from aha_amd import configs
    cfg = configs.tiny_qwen3vl()
    r = bench.cpu_baseline(cfg, sample_secs=0.5, kv_len=77, prompt_tokens=300, vit_patches=64)
    assert r["decode_kv_len"] == 77 and "77-token cache" in r["sample"]
    assert r["prefill_tok_s"] > 0 and r["prefill_seconds_extrapolated"] >= 0
    assert "EXTRAPOLATED" in r["prefill_sample"] and "S = 300" in r["prefill_sample"] and "64 patches" in r["prefill_sample"]


def test_decode_bytes_per_token_formula():
    import bench
    from aha_amd import configs
    cfg = configs.qwen3vl_8b()
    t = cfg.text
    b0 = bench.decode_bytes_per_token(cfg, 0)
    per_layer = (t.q_dim + 2 * t.kv_dim) * t.hidden_size + t.hidden_size * t.q_dim + 3 * t.intermediate_size * t.hidden_size
    assert b0 == (t.num_hidden_layers * per_layer + t.vocab_size * t.hidden_size) * 2 + t.num_hidden_layers * 2 * t.kv_dim * 2
    assert abs(b0 / 1e9 - 15.14) < 0.02                       # 15.14 GB + 147 456 B per cached token
    assert bench.decode_bytes_per_token(cfg, 1000) - b0 == 1000 * 147456


def test_prefill_flops_formula():
    """bench.py's roofline_prefill object: the FLOP count of BASELINE cfg 3's prefill (1542 prompt tokens, one 1024^2 image = 4096
    patches), by hand: text GEMMs 2 S H (q + 2 kv + q + 3 I) per layer, causal attention 4 q_dim S (S + 1) / 2 per layer, the ViT's
    four GEMMs per block over its patches, full attention inside the image."""
    import bench
    from aha_amd import configs
    cfg = configs.qwen3vl_8b()
    t, v = cfg.text, cfg.vision
    S, n = 1542, 4096
    pf = bench.prefill_flops(cfg, S, 0, [n])
    gemm = 2.0 * S * t.hidden_size * (2 * t.q_dim + 2 * t.kv_dim + 3 * t.intermediate_size)
    assert abs(pf["text_gemm"] - (36 * gemm + 2.0 * t.vocab_size * t.hidden_size)) < 1e6
    assert abs(pf["text_attention_causal"] - 36 * 4.0 * t.q_dim * S * (S + 1) / 2) < 1e6
    assert abs(pf["vit_attention"] - 27 * 4.0 * v.hidden_size * n * n) < 1e6
    assert 21.0e12 < pf["text_gemm"] < 21.8e12 and 3.5e12 < pf["vit_gemm"] < 3.9e12 and 27.5e12 < pf["total"] < 28.3e12
    # text only: no ViT terms; a cache offset adds S * offset (query, key) pairs
    tx = bench.prefill_flops(cfg.text, 128)
    assert tx["vit_gemm"] == 0.0 and tx["vit_attention"] == 0.0
    assert bench.prefill_flops(cfg.text, 128, 100)["text_attention_causal"] - tx["text_attention_causal"] == 36 * 4.0 * t.q_dim * 128 * 100


def test_bench_gpus_2_self_launches_two_ranks_over_gloo():
    """`python bench.py --gpus 2 ...` with no launcher around it (how the driver calls it) must start 2 ranks itself, rendezvous on
    127.0.0.1, aggregate (sum of units / max of seconds) and print ONE JSON line with n_gpus = 2.  The GPU work is replaced by the
    launcher self-test workload here (no GPU in this tier); tests/test_tp_gpu.py runs `--workload tiny` through the same launcher."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "1",
                        "--workload", "launcher-selftest"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 10 and j["config"]["replicas"] == 2
    assert abs(j["value"] - 2 * 10 / 0.75) < 1e-3 and abs(j["ms_per_step"] - 75.0) < 1e-6   # both ranks' units / the slower rank's 0.75 s


def test_bench_rejects_a_world_size_that_contradicts_gpus():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--workload", "launcher-selftest"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)


def test_committed_profiles_are_attached_only_under_a_matching_source_digest(tmp_path):
    """bench.py's roofline.traffic / roofline.rocprof come from committed rocprofv3 summaries: they must be attached when the summaries
    were taken with the CURRENT matvec sources and left out otherwise (a stale number next to a fresh ms_per_step is worse than none)."""
    import json
    import shutil
    import bench
    roof = {"algorithmic_bytes_per_launch": 104425355, "traffic": None}
    out = bench.attach_committed_profiles(dict(roof), "qwen3vl8b")
    dig = bench.source_digest(bench.GEMV_SOURCES)
    stats = open(os.path.join(ROOT, "profiles", bench.STATS_FILE)).read()
    stamped = stats.split("gemv_source_digest:")[1].split()[0]
    pmc = json.load(open(os.path.join(ROOT, "profiles", bench.PMC_FILE)))
    assert ("rocprof" in out) == (stamped == dig)
    assert (out["traffic"] is not None) == (pmc["gemv_source_digest"] == dig)
    if "rocprof" in out:
        assert 0.5 < out["rocprof"]["frac"] < 1.0 and abs(out["traffic"] / roof["algorithmic_bytes_per_launch"] - 1.0) < 0.1
    # a copy stamped with another digest is refused; another workload never gets the cfg 3 numbers
    shutil.copy(os.path.join(ROOT, "profiles", bench.PMC_FILE), tmp_path / bench.PMC_FILE)
    (tmp_path / bench.STATS_FILE).write_text(stats.replace(stamped, "0" * 16))
    pmc["gemv_source_digest"] = "0" * 16
    (tmp_path / bench.PMC_FILE).write_text(json.dumps(pmc))
    stale = bench.attach_committed_profiles(dict(roof), "qwen3vl8b", profiles_dir=str(tmp_path))
    assert "rocprof" not in stale and stale["traffic"] is None
    other = bench.attach_committed_profiles(dict(roof), "qwen3-0.6b")
    assert "rocprof" not in other and other["traffic"] is None


def test_guarded_leg_result_exception_and_watchdog():
    """bench.guarded: the multi-rank extra leg (sharded prefill) must not be able to take the replica decode line down with it -- a
    result passes through, an exception becomes text, and a call that never returns ends the process from a watchdog thread after
    on_timeout() has run (exit code 0, so that the launcher returns once every rank has left)."""
    import subprocess
    import sys
    import time
    import bench
    assert bench.guarded(lambda: 41 + 1, 5, lambda: None) == (42, None)
    res, err = bench.guarded(lambda: [][1], 5, lambda: None)
    assert res is None and err.startswith("IndexError")
    code = ("import sys, time; sys.path.insert(0, %r); import bench; "
            "bench.guarded(lambda: time.sleep(30), 0.5, lambda: print('line printed by the watchdog')); print('not reached')" % ROOT)
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and "line printed by the watchdog" in p.stdout and "not reached" not in p.stdout
    assert time.time() - t0 < 25


def test_sharded_legs_order_and_survival_of_an_exception():
    """bench.run_sharded_legs (round-4 verdict, item 4c): the tensor-parallel leg -- the form BASELINE cfg 5 names -- runs first by default, an
    exception in one leg becomes that leg's "error" object and the other leg is still measured (behind an agreement barrier inside its
    watchdog), each leg has its own top-level flag, and one rank skips the tensor-parallel form."""
    import bench
    calls, agreed = [], []

    def run_leg(mode, fail=()):
        calls.append(mode)
        if mode in fail:
            raise RuntimeError(f"{mode} broke")
        return {"rccl_ranks": 8, "first_token_equal_on_all_ranks": True, "mode": mode}

    line = {}
    assert bench.run_sharded_legs(["tp", "cp"], run_leg, 0, 8, 8, line, 30, lambda: agreed.append(1)) is True
    assert calls == ["tp", "cp"] and agreed == [1]            # the agreement precedes the SECOND leg only
    assert line["sharded_tp_ok"] is True and line["sharded_ok"] is True and line["sharded_prefill"]["mode"] == "cp"
    # the first leg raises: its object carries the error, its flag is False, the second leg still runs and is clean on its own
    calls.clear(); agreed.clear(); line = {}
    clean = bench.run_sharded_legs(["cp", "tp"], lambda m: run_leg(m, fail=("cp",)), 0, 8, 8, line, 30, lambda: agreed.append(1))
    assert clean is False and calls == ["cp", "tp"] and agreed == [1]
    assert "cp broke" in line["sharded_prefill"]["error"] and line["sharded_ok"] is False
    assert line["sharded_tp_ok"] is True and line["sharded_prefill_tp"]["mode"] == "tp"
    # a communicator smaller than --gpus is an error of that leg
    line = {}
    assert bench.run_sharded_legs(["cp"], run_leg, 0, 8, 4, line, 30) is False and "rccl_ranks 8 != --gpus 4" in line["sharded_prefill"]["error"]
    # one rank: only the context-parallel object (= the single-GPU cfg 5 prefill); other ranks write nothing
    calls.clear(); line = {}
    bench.run_sharded_legs(["tp", "cp"], lambda m: dict(run_leg(m), rccl_ranks=1), 0, 1, 1, line, 30)
    assert calls == ["cp"] and "sharded_prefill_tp" not in line
    line = {}
    bench.run_sharded_legs(["tp", "cp"], run_leg, 3, 8, 8, line, 30)
    assert line == {}
