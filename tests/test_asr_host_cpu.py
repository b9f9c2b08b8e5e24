"""CPU tier: the request-level host logic of Qwen3-ASR (aha_amd/audio_host.py Qwen3AsrProcessor) against hand-worked values of the
reference's arithmetic (/root/reference/src/models/qwen3_asr/processor.rs:84-195, src/utils/audio_utils.rs:740-755,1743-1760,
src/models/common/modules.rs:1353-1368, src/utils/mod.rs:549-558)."""
import dataclasses
import json

import numpy as np
import pytest

from aha_amd import audio_host as ah


def test_helpers_known_answers():
    assert ah.capitalize_first_letter("eNGLISH") == "English" and ah.capitalize_first_letter("") == "" and ah.capitalize_first_letter("ß") == "SS"
    # float_range_normalize: untouched below a peak of 1, scaled by 1 / peak above it (f32), clamped either way
    x = np.asarray([0.5, -0.25], dtype=np.float32)
    assert np.array_equal(ah.float_range_normalize(x), x)
    y = ah.float_range_normalize(np.asarray([0.5, -2.0, 1.0], dtype=np.float32))
    assert np.array_equal(y, np.asarray([0.25, -1.0, 0.5], dtype=np.float32))
    z = ah.float_range_normalize(np.asarray([3.0, -1.5], dtype=np.float32))
    assert z[0] == np.float32(3.0) * np.float32(1.0 / 3.0) and z[0] <= 1.0 and z.dtype == np.float32
    assert np.array_equal(ah.float_range_normalize(np.zeros(4, dtype=np.float32)), np.zeros(4, dtype=np.float32))
    # split_audio_into_chunks: <= max stays whole; else round(max * sr)-sample pieces + the remainder, which may be EMPTY
    assert [c.size for c in ah.split_audio_into_chunks(np.zeros(20), 10, 2.0)] == [20]
    assert [c.size for c in ah.split_audio_into_chunks(np.zeros(50), 10, 2.0)] == [20, 20, 10]
    assert [c.size for c in ah.split_audio_into_chunks(np.zeros(40), 10, 2.0)] == [20, 20, 0]
    w = np.arange(50, dtype=np.float32)
    assert np.array_equal(np.concatenate(ah.split_audio_into_chunks(w, 10, 2.0)), w)
    # get_feat_extract_output_lengths (processor.rs:187-195): 100-frame windows give 13 tokens, the remainder three stride-2 convs
    assert [ah.get_feat_extract_output_lengths(n) for n in (100, 3000, 1, 8, 99, 150)] == [13, 390, 1, 1, 13, 20]


def test_extract_audio_url_untagged_parts():
    msgs = [{"role": "user", "content": [{"type": "audio", "audio_url": {"url": "a.wav"}}, {"type": "text", "text": "t", "audio_url": {"url": "no"}},
                                         {"type": "x", "image_url": {"url": "i"}, "audio_url": {"url": "no2"}}, {"type": "input_audio", "audio_url": {"url": "b.wav"}}]},
            {"role": "assistant", "content": [{"type": "audio", "audio_url": {"url": "c.wav"}}]}, {"role": "user", "content": "plain"}]
    assert ah.extract_audio_url(msgs) == ["a.wav", "b.wav"]


@pytest.fixture(scope="module")
def asr_tok(tmp_path_factory):
    tokenizers = pytest.importorskip("tokenizers")
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    from aha_amd import text_host as th
    d = tmp_path_factory.mktemp("asrtok")
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)
    tok.decoder = decoders.ByteLevel()
    tok.train_from_iterator(["system user assistant language English Chinese '<asr_text>'"] * 4,
                            trainers.BpeTrainer(vocab_size=320, initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), special_tokens=[]))
    tok.model.save(str(d))
    base = tok.get_vocab_size()
    names = ["<|im_start|>", "<|im_end|>", "<|audio_start|>", "<|audio_end|>", "<|audio_pad|>"]
    json.dump({"added_tokens_decoder": {str(base + i): {"content": n, "special": True} for i, n in enumerate(names)}},
              open(d / "tokenizer_config.json", "w"))
    t = th.TokenizerModel.init(str(d))
    return t, {n: t.tokenizer.token_to_id(n) for n in names}


def test_process_info_chunks_language_and_counts(asr_tok):
    from aha_amd.configs import tiny_qwen3_asr
    t, tid = asr_tok
    cfg = tiny_qwen3_asr()
    sr = 16000
    clips = {"a.wav": np.full(3 * sr + 80, 1.5, dtype=np.float32),                  # 3.005 s, peak 1.5 -> scaled to 1.0
             "b.wav": np.random.default_rng(0).normal(0, 0.1, 5 * sr).astype(np.float32)}
    p = ah.Qwen3AsrProcessor(cfg, clips.__getitem__, max_asr_input_seconds=2.0)    # 2 s chunks so the chunking shows at test size
    msgs = [{"role": "user", "content": [{"type": "audio", "audio_url": {"url": "a.wav"}}, {"type": "audio", "audio_url": {"url": "b.wav"}}]}]
    render = "<|im_start|>user\n" + ah.AUDIO_RUN * 2 + "<|im_end|>\n<|im_start|>assistant\n"
    out = p.process_info(msgs, render, t, metadata={"language": "english"})
    # a.wav: 48080 samples -> 32000 + 16080; b.wav: 80000 -> 32000 + 32000 + 16000
    assert [d.audio_samples.size for _, d in out] == [32000, 16080, 32000, 32000, 16000]
    for ids, d in out:
        n_tok = ah.get_feat_extract_output_lengths(d.audio_samples.size // 160)
        assert ids.count(tid["<|audio_pad|>"]) == n_tok and ids.count(tid["<|audio_start|>"]) == 1 == ids.count(tid["<|audio_end|>"])
        text = t.token_decode_with_special(ids)
        assert text.endswith("<|im_start|>assistant\nlanguage English'<asr_text>'") and text.count(ah.AUDIO_RUN) == 0
    assert [ids.count(tid["<|audio_pad|>"]) for ids, _ in out] == [26, 13, 26, 26, 13]   # 200, 100 (16080 // 160), 200, 200, 100 frames
    assert np.all(out[0][1].audio_samples == np.float32(1.5) * np.float32(1.0 / 1.5)) and np.abs(out[0][1].audio_samples).max() <= 1.0
    # unsupported language: no suffix; one run, one audio: nothing collapsed
    one = p.process_info(msgs[:1] and [{"role": "user", "content": [{"type": "audio", "audio_url": {"url": "b.wav"}}]}],
                         "<|im_start|>user\n" + ah.AUDIO_RUN + "<|im_end|>\n", t, metadata={"language": "klingon"})
    assert len(one) == 3 and not t.token_decode_with_special(one[0][0]).endswith("'<asr_text>'")
    # the number of audio parts must match the number of runs (processor.rs:151-154)
    with pytest.raises(ValueError, match="audio_pad num != audio num"):
        p.process_info(msgs, "<|im_start|>user\n" + ah.AUDIO_RUN + "<|im_end|>\n", t)
    # a clip whose length is an exact multiple of the chunk: the reference's empty remainder chunk is an error here too
    clips["c.wav"] = np.zeros(4 * sr, dtype=np.float32)
    with pytest.raises(ValueError, match="shorter than one feature frame"):
        p.process_info([{"role": "user", "content": [{"type": "audio", "audio_url": {"url": "c.wav"}}]}], ah.AUDIO_RUN, t)


def test_process_audio_tensor_default_template(asr_tok):
    from aha_amd.configs import tiny_qwen3_asr
    t, tid = asr_tok
    p = ah.Qwen3AsrProcessor(tiny_qwen3_asr(), None)
    wav = np.random.default_rng(1).normal(0, 0.2, 16000 * 7 + 123).astype(np.float32)
    ids, d = p.process_audio_tensor(ah.DEFAULT_TEMPLATE, wav, t)
    assert ids.count(tid["<|audio_pad|>"]) == ah.get_feat_extract_output_lengths((16000 * 7 + 123) // 160) == 91
    assert t.token_decode_with_special(ids).startswith("<|im_start|>system\n<|im_end|>\n<|im_start|>user\n<|audio_start|><|audio_pad|>")
    with pytest.raises(ValueError, match="too long"):
        ah.Qwen3AsrProcessor(tiny_qwen3_asr(), None, max_asr_input_seconds=1.0).process_audio_tensor(ah.DEFAULT_TEMPLATE, wav, t)
