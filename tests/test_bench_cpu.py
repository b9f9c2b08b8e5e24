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
