"""Property tests (hypothesis) of host-side logic on the request path: smart-resize arithmetic (C ABI and Python mirror), the
sampler's draw weights, resampler output lengths."""
import ctypes as C
import math

import numpy as np
import pytest

hyp = pytest.importorskip("hypothesis")
from hypothesis import given, settings, strategies as st  # noqa: E402

from aha_amd import _lib  # noqa: E402
from aha_amd import sampling as hs  # noqa: E402
from aha_amd.vision_host import img_smart_resize  # noqa: E402
from oracle import audio_pre as ap  # noqa: E402
from oracle import sampling as osamp  # noqa: E402


@settings(max_examples=200, deadline=None, derandomize=True)
@given(h=st.integers(1, 6000), w=st.integers(1, 6000))
def test_smart_resize_invariants(h, w):
    """img_smart_resize (img_utils.rs:294-331): multiples of 32, area inside [min, max] (up to one factor step of rounding),
    aspect ratio kept within the rounding granularity; C ABI == Python mirror."""
    if max(h, w) // min(h, w) > 200:
        with pytest.raises(ValueError):
            img_smart_resize(h, w)
        return
    mn, mx = 65536, 16777216
    hb, wb = img_smart_resize(h, w, 32, mn, mx)
    ho, wo = C.c_uint32(), C.c_uint32()
    assert _lib.lib().aha_hip_img_smart_resize(h, w, 32, mn, mx, C.byref(ho), C.byref(wo)) == 0
    assert (ho.value, wo.value) == (hb, wb)
    assert hb % 32 == 0 and wb % 32 == 0 and hb >= 32 and wb >= 32
    assert hb * wb <= mx
    if min(h, w) * 200 >= max(h, w) and h * w >= 4:
        assert hb * wb >= mn or max(hb, wb) / min(hb, wb) > 50      # extreme strips can stay below the minimum area
    # scale factors of the two axes agree up to the 32-pixel granularity
    sh, sw = hb / h, wb / w
    assert abs(sh - sw) <= 33 / min(h, w) + 1e-6 or min(hb, wb) == 32


@settings(max_examples=100, deadline=None, derandomize=True)
@given(seed=st.integers(0, 10_000), k=st.integers(1, 64), p=st.floats(0.05, 0.999), t=st.floats(0.2, 2.0),
       scale=st.floats(0.2, 8.0))
def test_sampler_weights_properties(seed, k, p, t, scale):
    """TopKThenTopP draw weights: non-negative, supported on the k largest logits, the best token always kept, the kept mass at
    least min(top_p, mass of the top k), and equal to the oracle's weights on the same logits."""
    V = 512
    logits = (np.random.default_rng(seed).standard_normal(V) * scale).astype(np.float32)
    lp = hs.LogitsProcessor(0, hs.Sampling("TopKThenTopP", float(np.float32(t)), k, float(np.float32(p))))
    vals, idx, mx, se = osamp.topk_candidates(logits, k, lp.sampling.temperature)
    w = lp.weights_from_candidates(vals, mx, se)
    assert w is not None and w.shape == (k,) and (w >= 0).all() and w[0] > 0
    topk_mass = float(np.exp((vals.astype(np.float64) - mx) / lp.sampling.temperature).sum() / se)
    assert float(w.sum()) >= min(lp.sampling.p, topk_mass) - 1e-4
    assert (np.diff(np.nonzero(w)[0]) == 1).all()                   # a prefix of the descending candidates survives
    want = osamp.final_weights(logits, osamp.Sampling("TopKThenTopP", lp.sampling.temperature, k=k, p=lp.sampling.p))
    got = np.zeros(V, np.float32)
    got[idx] = w
    boundary = np.abs(np.cumsum(np.sort(want[want > 0])[::-1]) - lp.sampling.p).min() < 1e-5 if (want > 0).any() else False
    if not boundary:                                                # a running sum within 1e-5 of top_p may cut one token apart
        assert set(np.nonzero(got)[0]) == set(np.nonzero(want)[0])
        np.testing.assert_allclose(got, want, rtol=3e-5, atol=1e-9)


@settings(max_examples=60, deadline=None, derandomize=True)
@given(n=st.integers(0, 5000), rates=st.sampled_from([(44100, 16000), (48000, 16000), (8000, 16000), (22050, 16000), (16000, 24000)]))
def test_resample_length_formula(n, rates):
    """ceil(new * n / orig), capped by the convolution's output rows (audio_utils.rs:200-207)."""
    orig_sr, new_sr = rates
    g = math.gcd(orig_sr, new_sr)
    orig, new = orig_sr // g, new_sr // g
    y = ap.resample_simple(np.zeros(n, np.float32), orig_sr, new_sr)
    assert y.shape[0] == min(math.ceil(new * n / orig), (n // orig + 1) * new)
    assert abs(y.shape[0] - n * new_sr / orig_sr) < 1 + 1e-9


# ---- native JSON / config parser (csrc/json.h, loader.hip) under formatting noise -------------------------------------------------
json_leaf = st.one_of(st.none(), st.booleans(), st.integers(-2**53, 2**53), st.floats(allow_nan=False, allow_infinity=False, width=64),
                      st.text(max_size=12))
json_value = st.recursive(json_leaf, lambda ch: st.one_of(st.lists(ch, max_size=4), st.dictionaries(st.text(max_size=6), ch, max_size=4)),
                          max_leaves=12)


@settings(max_examples=60, deadline=None, derandomize=True)
@given(extras=st.dictionaries(st.text(min_size=1, max_size=8).map(lambda s: "x_" + s), json_value, max_size=5),
       indent=st.sampled_from([None, 0, 1, 4]), ascii_only=st.booleans(), seed=st.integers(0, 1000),
       eps_fmt=st.sampled_from(["{:e}", "{:.12f}", "{!r}", "{:E}"]), theta_fmt=st.sampled_from(["{:e}", "{:.1f}", "{!r}", "{:.0f}"]))
def test_config_parser_ignores_formatting_and_unknown_keys(extras, indent, ascii_only, seed, eps_fmt, theta_fmt):
    """serde ignores unknown fields and JSON formatting; so must aha_hip_config_parse: shuffled keys, any indentation, escaped or
    raw unicode, numbers in any JSON spelling and arbitrary extra members leave the parsed description unchanged."""
    import json
    import random
    import tempfile

    import torch

    from aha_amd.checkpoint import config_json, parse_config, save_checkpoint
    from aha_amd.configs import tiny_qwen3
    from aha_amd.model import make_desc

    cfg = tiny_qwen3()
    want = make_desc(cfg)
    with tempfile.TemporaryDirectory() as d:
        save_checkpoint(d, cfg, {"dummy": torch.zeros(1)})
        base = json.load(open(f"{d}/config.json"))
        items = list(base.items()) + list(extras.items())
        random.Random(seed).shuffle(items)
        text = json.dumps(dict(items), indent=indent, ensure_ascii=ascii_only)
        # respell the two float fields (json.dumps always writes repr)
        for key, fmt in (("rms_norm_eps", eps_fmt), ("rope_theta", theta_fmt)):
            val = float(base[key])
            new = fmt.format(val)
            if float(new) != val:
                continue  # this spelling would change the value
            old = json.dumps(base[key])
            assert f'"{key}": {old}' in text
            text = text.replace(f'"{key}": {old}', f'"{key}": {new}')
        open(f"{d}/config.json", "w", encoding="utf-8").write(text)
        got = parse_config(d)
    for name, _ in want._fields_:
        a, b = getattr(got, name), getattr(want, name)
        assert (list(a) if hasattr(a, "__len__") else a) == (list(b) if hasattr(b, "__len__") else b), name


@settings(max_examples=150, deadline=None, derandomize=True)
@given(seed=st.integers(0, 10**6), n_items=st.integers(1, 6), max_g=st.integers(1, 8))
def test_rope_index_with_videos_differential(seed, n_items, max_g):
    """The same differential with videos mixed in (model.rs:908-925,973-981): a (t, h, w) video grid stands for t frames (1, h, w),
    each behind its own <|vision_start|> after some timestamp tokens -- images and videos interleaved in random order."""
    from aha_amd.configs import tiny_qwen3vl
    from aha_amd.vision_host import get_rope_index, video_prompt_ids
    from oracle import qwen3vl as ov
    cfg = tiny_qwen3vl()
    rng = np.random.default_rng(seed)
    ids, grids, vgrids = [int(x) for x in rng.integers(0, 1000, size=rng.integers(0, 4))], [], []
    for _ in range(n_items):
        gh, gw = int(rng.integers(1, max_g + 1)), int(rng.integers(1, max_g + 1))
        if rng.integers(0, 2):
            t = int(rng.integers(1, 5))
            vgrids.append([t, 2 * gh, 2 * gw])
            stamps = [[int(x) for x in rng.integers(0, 1000, size=rng.integers(0, 4))] for _ in range(t)]
            ids += video_prompt_ids(cfg, np.asarray([vgrids[-1]]), stamps)
        else:
            grids.append([1, 2 * gh, 2 * gw])
            ids += [cfg.vision_start_token_id] + [cfg.image_token_id] * (gh * gw) + [cfg.vision_end_token_id]
        ids += [int(x) for x in rng.integers(0, 1000, size=rng.integers(0, 5))]
    grid = np.asarray(grids, dtype=np.uint32).reshape(-1, 3)
    vgrid = np.asarray(vgrids, dtype=np.uint32).reshape(-1, 3)
    pos, delta = get_rope_index(cfg, ids, grid, vgrid)
    ref_pos, ref_delta = ov.get_rope_index(ids, grid, cfg, vgrid)
    np.testing.assert_array_equal(pos, np.asarray(ref_pos))
    assert delta == int(ref_delta) and int(pos.max()) + 1 - len(ids) == delta
    vis = np.asarray([t in (cfg.image_token_id, cfg.video_token_id) for t in ids])
    assert (pos[0, ~vis] == pos[1, ~vis]).all() and (pos[0, ~vis] == pos[2, ~vis]).all()
    # every video frame is a t = 1 grid: its T row is constant over the frame's tokens
    j = 0
    while j < len(ids):
        if ids[j] == cfg.video_token_id:
            k = j
            while k < len(ids) and ids[k] == cfg.video_token_id:
                k += 1
            assert len(set(pos[0, j:k].tolist())) == 1
            j = k
        else:
            j += 1


def test_rope_index_video_errors():
    """More <|vision_start|><|video_pad|> runs than frames in the grids, or a frame longer than the rest of the prompt: an error,
    not a read past the grid (the reference indexes out of bounds / fails the final reshape there)."""
    from aha_amd._lib import AhaHipError
    from aha_amd.configs import tiny_qwen3vl
    from aha_amd.vision_host import get_rope_index, video_prompt_ids
    cfg = tiny_qwen3vl()
    ids = video_prompt_ids(cfg, np.asarray([[3, 2, 2]]))
    with pytest.raises(AhaHipError):
        get_rope_index(cfg, ids, None, np.asarray([[2, 2, 2]], dtype=np.uint32))
    with pytest.raises(AhaHipError):
        get_rope_index(cfg, ids[:-3], None, np.asarray([[3, 4, 4]], dtype=np.uint32))


@settings(max_examples=150, deadline=None, derandomize=True)
@given(seed=st.integers(0, 10**6), n_images=st.integers(0, 6), max_g=st.integers(1, 12))
def test_rope_index_differential(seed, n_images, max_g):
    """aha_hip_get_rope_index (host C++, csrc/vision.hip) vs the oracle restatement of Qwen3VLModel::get_rope_index
    (qwen3vl/model.rs:901-1133) on random prompts: integer work, bit-exact, and the structural invariants of the positions."""
    from aha_amd.configs import tiny_qwen3vl
    from aha_amd.vision_host import get_rope_index
    from oracle import qwen3vl as ov
    cfg = tiny_qwen3vl()
    rng = np.random.default_rng(seed)
    ids, grids = [int(x) for x in rng.integers(0, 1000, size=rng.integers(0, 6))], []
    for _ in range(n_images):
        gh, gw = int(rng.integers(1, max_g + 1)), int(rng.integers(1, max_g + 1))
        grids.append([1, 2 * gh, 2 * gw])
        ids += [cfg.vision_start_token_id] + [cfg.image_token_id] * (gh * gw) + [cfg.vision_end_token_id]
        ids += [int(x) for x in rng.integers(0, 1000, size=rng.integers(0, 5))]
    if not ids:
        ids = [1]
    grid = np.asarray(grids, dtype=np.uint32).reshape(-1, 3)
    pos, delta = get_rope_index(cfg, ids, grid)
    ref_pos, ref_delta = ov.get_rope_index(ids, grid, cfg)
    np.testing.assert_array_equal(pos, np.asarray(ref_pos))
    assert delta == int(ref_delta)
    assert pos.shape == (3, len(ids)) and (pos >= 0).all()
    assert int(pos.max()) + 1 - len(ids) == delta and delta <= 0          # rope_deltas = max_pos + 1 - S (model.rs:1128-1131)
    txt = np.asarray([t != cfg.image_token_id for t in ids])
    assert (pos[0, txt] == pos[1, txt]).all() and (pos[0, txt] == pos[2, txt]).all()   # text tokens: the same index on T, H, W
