"""Host arithmetic of the Qwen3-VL video path (aha_amd/vision_host.py) against the oracle restatement and against values worked
out by hand from the reference's formulas (/root/reference/src/utils/video_utils.rs:9-59, src/models/qwen3vl/processor.rs:283-307,
386-431, 481-535).  The ffmpeg decode / swscale resize around it is third-party and is not mirrored."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from aha_amd import vision_host as vh
from oracle import qwen3vl as ov


def test_video_smart_resize_known_answers():
    # factor = lcm(32, 16) = 32; 360 / 32 = 11.25 -> 11 -> 352; 20 * 352 * 640 pixels is inside [min, max]
    assert vh.video_smart_resize(20, 360, 640) == (352, 640)
    # over max_pixels: beta = sqrt(8 * 360 * 640 / 786432) = 1.53093; floor(360 / beta / 32) = 7, floor(640 / beta / 32) = 13
    assert vh.video_smart_resize(8, 360, 640, 2, 32, 4096, 786432, 16) == (224, 416)
    # under min_pixels: beta = sqrt(4194304 / (4 * 64 * 64)) = 16; ceil(64 * 16 / 32) * 32 = 1024
    assert vh.video_smart_resize(4, 64, 64, 2, 32, 4194304, 25165824, 16) == (1024, 1024)
    # video_ratio = 24: factor lcm(32, 24) = 96; 360 / 96 = 3.75 -> 4 -> 384; 640 / 96 = 6.67 -> 7 -> 672
    assert vh.video_smart_resize(20, 360, 640, 2, 32, 4096, 25165824, 24) == (384, 672)
    with pytest.raises(ValueError):
        vh.video_smart_resize(1, 360, 640)          # fewer frames than a temporal patch
    with pytest.raises(ValueError):
        vh.video_smart_resize(8, 16, 640)           # smaller than the factor
    with pytest.raises(ValueError):
        vh.video_smart_resize(8, 32, 32 * 201)      # aspect ratio > 200


def test_frame_sampling_and_timestamps_known_answers():
    # 10 s at 25 fps, 2 frames per second: 20 frames; interval = round(12.5) = 13 (f32::round: half away from zero)
    n, iv, idx = vh.sample_video_frames(250, 25.0)
    assert (n, iv) == (20, 13) and idx == list(range(0, 250, 13)) and len(idx) == 20
    # 1 s clip: round(2) = 2 -> min_frames 4; interval = round(7.5) = 8
    assert vh.sample_video_frames(30, 30.0) == (4, 8, [0, 8, 16, 24])
    # fewer frames than min_frames: clamped to the frame count, every frame kept
    assert vh.sample_video_frames(3, 30.0) == (3, 1, [0, 1, 2])
    # long clip: capped at max_frames = 768
    n, iv, idx = vh.sample_video_frames(100000, 25.0)
    assert n == 768 and iv == 130 and idx[:3] == [0, 130, 260]
    # timestamps: [0, 13, 26] padded with 26; (0 + 0.52) / 2 = 0.26, (1.04 + 1.04) / 2 = 1.04 (f32)
    ts = vh.calculate_timestamps([0, 13, 26], 25.0)
    assert ts == [float(np.float32(0.52) / np.float32(2)), float(np.float32(26) / np.float32(25))]
    assert ["%.1f" % t for t in ts] == ["0.3", "1.0"]


def test_placeholder_expansion_known_answer():
    text = "u<|vision_start|><|image_pad|><|vision_end|>v<|vision_start|><|video_pad|><|vision_end|>w"
    out = vh.expand_vision_placeholders(text, [[1, 4, 6]], [[2, 4, 4]], [([0, 13, 26], 25.0)])
    frame = lambda s: f"<{s} seconds><|vision_start|>" + "<|video_pad|>" * 4 + "<|vision_end|>"
    assert out == "u<|vision_start|>" + "<|image_pad|>" * 6 + "<|vision_end|>v" + frame("0.3") + frame("1.0") + "w"
    # a bare <|video_pad|> (no surrounding start / end in the template) gets the same block in its place
    assert vh.expand_vision_placeholders("a<|video_pad|>b", None, [[1, 2, 2]], [([0, 5], 10.0)]) == \
        "a<0.2 seconds><|vision_start|><|video_pad|><|vision_end|>b"


@settings(max_examples=300, deadline=None, derandomize=True)
@given(nf=st.integers(2, 900), h=st.integers(32, 2200), w=st.integers(32, 4000), max_pix=st.sampled_from([786432, 25165824, 4194304]),
       min_pix=st.sampled_from([4096, 262144]), ratio=st.sampled_from([None, 16, 24]))
def test_video_smart_resize_differential(nf, h, w, max_pix, min_pix, ratio):
    if max(h, w) // min(h, w) > 200:
        return
    a = vh.video_smart_resize(nf, h, w, 2, 32, min_pix, max_pix, ratio)
    b = ov.video_smart_resize(nf, h, w, 2, 32, min_pix, max_pix, ratio)
    assert a == b
    f = 32 if ratio is None else int(np.lcm(32, ratio))
    assert a[0] % f == 0 and a[1] % f == 0 and a[0] >= f and a[1] >= f


@settings(max_examples=300, deadline=None, derandomize=True)
@given(frames=st.integers(1, 200000), rate=st.sampled_from([23.976, 24.0, 25.0, 29.97, 30.0, 50.0, 60.0, 12.5]), fps=st.integers(1, 4))
def test_frame_sampling_differential(frames, rate, fps):
    n, iv, idx = vh.sample_video_frames(frames, rate, fps)
    assert (n, iv, idx) == ov.sample_frame_indices(frames, rate, fps)
    assert 1 <= n <= min(768, frames) and iv >= 1 and idx[0] == 0 and all(b - a == iv for a, b in zip(idx, idx[1:])) and idx[-1] < frames
    ts = vh.calculate_timestamps(idx, rate)
    assert ts == ov.calculate_timestamps(idx, rate) and len(ts) == (len(idx) + 1) // 2
    assert all(b > a for a, b in zip(ts, ts[1:])) or len(idx) % 2 == 1


@settings(max_examples=100, deadline=None, derandomize=True)
@given(seed=st.integers(0, 10**6))
def test_placeholder_expansion_differential(seed):
    rng = np.random.default_rng(seed)
    n_img, n_vid = int(rng.integers(0, 3)), int(rng.integers(0, 3))
    igrid = [[1, 2 * int(rng.integers(1, 5)), 2 * int(rng.integers(1, 5))] for _ in range(n_img)]
    vgrid, meta = [], []
    for _ in range(n_vid):
        t = int(rng.integers(1, 5))
        nfr = 2 * t - int(rng.integers(0, 2))
        vgrid.append([t, 2 * int(rng.integers(1, 4)), 2 * int(rng.integers(1, 4))])
        meta.append((sorted(int(x) for x in rng.choice(500, size=nfr, replace=False)), float(rng.choice([24.0, 25.0, 29.97]))))
    parts = ["<|vision_start|><|image_pad|><|vision_end|>"] * n_img + ["<|vision_start|><|video_pad|><|vision_end|>"] * n_vid
    rng.shuffle(parts)
    text = "sys " + " and ".join(parts) + " end"
    a = vh.expand_vision_placeholders(text, igrid or None, vgrid or None, meta)
    assert a == ov.expand_placeholders(text, igrid or None, vgrid or None, meta)
    assert a.count("<|image_pad|>") == sum(g[0] * g[1] * g[2] // 4 for g in igrid)
    assert a.count("<|video_pad|>") == sum(g[0] * g[1] * g[2] // 4 for g in vgrid)
    assert a.count(" seconds>") == sum(g[0] for g in vgrid)


def test_request_pipeline_text_to_positions(tmp_path):
    """The host side of a Qwen3-VL request with an image and a video, end to end without a GPU: rendered chat text ->
    expand_vision_placeholders (process_info) -> tokenizer (TokenizerModel mirror, special tokens as added tokens) -> ids ->
    get_rope_index through the C ABI -- the ids must carry exactly the pads the grids promise, one <|vision_start|> per image and per
    temporal patch, the timestamps must survive tokenisation, and the library's positions must equal the oracle's."""
    import dataclasses
    import json
    tokenizers = pytest.importorskip("tokenizers")
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    from aha_amd import text_host as th
    from aha_amd.configs import tiny_qwen3vl
    corpus = ["describe the clip and the picture", "<0.3 seconds> <12.5 seconds> <7.0 seconds>", "user assistant system 0123456789 ."] * 4
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)
    tok.decoder = decoders.ByteLevel()
    tok.train_from_iterator(corpus, trainers.BpeTrainer(vocab_size=350, initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), special_tokens=[]))
    tok.model.save(str(tmp_path))
    base = tok.get_vocab_size()
    names = ["<|im_start|>", "<|im_end|>", "<|vision_start|>", "<|vision_end|>", "<|image_pad|>", "<|video_pad|>"]
    json.dump({"added_tokens_decoder": {str(base + i): {"content": n, "special": True} for i, n in enumerate(names)}},
              open(tmp_path / "tokenizer_config.json", "w"))
    t = th.TokenizerModel.init(str(tmp_path))
    tid = {n: t.tokenizer.token_to_id(n) for n in names}
    cfg = dataclasses.replace(tiny_qwen3vl(), vision_start_token_id=tid["<|vision_start|>"], vision_end_token_id=tid["<|vision_end|>"],
                              image_token_id=tid["<|image_pad|>"], video_token_id=tid["<|video_pad|>"])
    # a 10 s clip at 25 fps: 20 sampled frames -> 10 temporal patches; frames scaled to what video_smart_resize names
    nframes, interval, idx = vh.sample_video_frames(250, 25.0)
    rh, rw = vh.video_smart_resize(nframes, 360, 640, 2, 32, 4096, 786432, 16)
    igrid = np.asarray([[1, 8, 12]], dtype=np.uint32)
    vgrid = np.asarray([[(len(idx) + 1) // 2, rh // 16, rw // 16]], dtype=np.uint32)
    text = ("<|im_start|>user\n<|vision_start|><|image_pad|><|vision_end|><|vision_start|><|video_pad|><|vision_end|>"
            "describe the clip and the picture<|im_end|>\n<|im_start|>assistant\n")
    full = vh.expand_vision_placeholders(text, igrid, vgrid, [(idx, 25.0)])
    ids = t.text_encode(full)
    n_img_pads, n_vid_pads = int(np.prod(igrid[0])) // 4, int(np.prod(vgrid[0])) // 4
    assert ids.count(cfg.image_token_id) == n_img_pads and ids.count(cfg.video_token_id) == n_vid_pads
    assert ids.count(cfg.vision_start_token_id) == 1 + int(vgrid[0, 0]) == ids.count(cfg.vision_end_token_id)
    back = t.token_decode_with_special(ids)
    assert back == full and "<0.3 seconds>" in back and back.count(" seconds>") == int(vgrid[0, 0])
    pos, delta = vh.get_rope_index(cfg, ids, igrid, vgrid)
    ref_pos, ref_delta = ov.get_rope_index(ids, igrid, cfg, vgrid)
    assert np.array_equal(pos, ref_pos) and delta == ref_delta
    # positions advance by max(h, w) / 2 per frame, not by its token count: the prompt is far shorter in position space
    assert delta == int(pos.max()) + 1 - len(ids) and delta < -(n_vid_pads // 2)


def test_process_info_and_get_data_orchestration(tmp_path):
    """Qwen3VLProcessor::process_info + Qwen3VLGenerateModel::get_data mirrors (processor.rs:126-149,310-444; generate.rs:79-101) on
    the CPU: content parts -> sources (untagged-enum rules, user messages only), a data: URI image through media_host.get_image,
    a video through a caller-supplied loader that honours the plan (sampling + resize size), patch rows by the oracle's
    process_images / process_videos, the text rewrite, the tokenizer, and the library's get_rope_index on the result."""
    import base64
    import dataclasses
    import io
    import json
    tokenizers = pytest.importorskip("tokenizers")
    PIL = pytest.importorskip("PIL.Image")
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    from aha_amd import text_host as th
    from aha_amd.configs import tiny_qwen3vl
    from oracle.numerics import Numerics

    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)
    tok.decoder = decoders.ByteLevel()
    tok.train_from_iterator(["what happens in the clip compared to the picture user assistant <0.5 seconds>"] * 4,
                            trainers.BpeTrainer(vocab_size=330, initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), special_tokens=[]))
    tok.model.save(str(tmp_path))
    base = tok.get_vocab_size()
    names = ["<|im_start|>", "<|im_end|>", "<|vision_start|>", "<|vision_end|>", "<|image_pad|>", "<|video_pad|>"]
    template = ("{%- for m in messages %}{{- '<|im_start|>' + m.role + '\\n' }}{%- if m.content is string %}{{- m.content }}{%- else %}"
                "{%- for p in m.content %}{%- if 'image_url' in p and m.role == 'user' %}{{- '<|vision_start|><|image_pad|><|vision_end|>' }}"
                "{%- elif 'video_url' in p and m.role == 'user' %}{{- '<|vision_start|><|video_pad|><|vision_end|>' }}{%- elif 'text' in p %}{{- p.text }}{%- endif %}"
                "{%- endfor %}{%- endif %}{{- '<|im_end|>\\n' }}{%- endfor %}{%- if add_generation_prompt %}{{- '<|im_start|>assistant\\n' }}{%- endif %}")
    json.dump({"added_tokens_decoder": {str(base + i): {"content": n, "special": True} for i, n in enumerate(names)},
               "chat_template": template}, open(tmp_path / "tokenizer_config.json", "w"))
    t = th.TokenizerModel.init(str(tmp_path))
    ct = th.ChatTemplate.init(str(tmp_path))
    tid = {n: t.tokenizer.token_to_id(n) for n in names}
    cfg = dataclasses.replace(tiny_qwen3vl(), vision_start_token_id=tid["<|vision_start|>"], vision_end_token_id=tid["<|vision_end|>"],
                              image_token_id=tid["<|image_pad|>"], video_token_id=tid["<|video_pad|>"])

    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(64, 96, 3), dtype=np.uint8)           # already a smart-resize size: no resize in image_fn below
    buf = io.BytesIO()
    PIL.fromarray(img).save(buf, format="PNG")
    uri = "data:image/png;base64," + base64.b64encode(buf.getvalue()).decode()
    seen = {}

    def video_loader(url, plan):
        p = plan(90, 30.0, 70, 100)                                         # a 3 s, 30 fps, 100 x 70 clip
        seen.update(p, url=url)
        rh, rw = p["resize_hw"]
        return rng.integers(0, 256, size=(len(p["frame_indices"]), rh, rw, 3), dtype=np.uint8), p["frame_indices"], 30.0

    nm = Numerics("bf16")
    proc = vh.Qwen3VLProcessor(cfg, video_loader=video_loader, image_fn=lambda imgs: ov.process_images(nm, imgs),
                               video_fn=lambda vids: ov.process_videos(nm, vids), device="cpu")
    messages = [{"role": "system", "content": "sys"},
                {"role": "user", "content": [{"type": "image", "image_url": {"url": uri, "detail": "auto"}},
                                             {"type": "video", "video_url": {"url": "file:///clip.mp4"}},
                                             {"type": "text", "text": "what happens in the clip compared to the picture"}]},
                {"role": "assistant", "content": [{"type": "image", "image_url": {"url": "file:///ignored.png"}}]}]
    assert vh.extract_vision_info(messages) == {"image": [uri], "video": ["file:///clip.mp4"]}
    ids, data = vh.get_data(messages, ct, t, proc)
    # the plan: round(90 / 30 * 2) = 6 frames, interval round(15) = 15, frames 0..75; 100 x 70 -> factor 32: (64, 96)
    assert seen["url"] == "file:///clip.mp4" and seen["nframes"] == 6 and seen["sample_interval"] == 15
    assert seen["frame_indices"] == [0, 15, 30, 45, 60, 75] and seen["resize_hw"] == (64, 96)
    assert data.image_grid_thw.tolist() == [[1, 4, 6]] and data.video_grid_thw.tolist() == [[3, 4, 6]]
    assert data.pixel_values.shape == (24, 1536) and data.pixel_values_video.shape == (72, 1536)
    assert ids.count(cfg.image_token_id) == 6 and ids.count(cfg.video_token_id) == 18 and ids.count(cfg.vision_start_token_id) == 1 + 3
    text = t.token_decode_with_special(ids)
    assert text.count(" seconds>") == 3 and "<0.2 seconds>" in text      # (0 + 15/30) / 2 = 0.25 -> "0.2" (ties to even on the exact f32 0.25)
    pos, delta = vh.get_rope_index(cfg, ids, data.image_grid_thw, data.video_grid_thw)
    ref_pos, ref_delta = ov.get_rope_index(ids, data.image_grid_thw, cfg, data.video_grid_thw)
    assert np.array_equal(pos, ref_pos) and delta == ref_delta and not proc.warnings
    # a source that cannot be loaded is reported and skipped; with nothing left the text is not rewritten
    proc2 = vh.Qwen3VLProcessor(cfg, image_fn=lambda imgs: ov.process_images(nm, imgs), device="cpu")
    bad = [{"role": "user", "content": [{"type": "image", "image_url": {"url": "https://example.invalid/a.png"}}, {"type": "text", "text": "hi"}]}]
    out = proc2.process_info(bad, ct.apply_chat_template(bad))
    assert out["pixel_values"] is None and out["replace_text"].count("<|image_pad|>") == 1 and len(proc2.warnings) == 1
