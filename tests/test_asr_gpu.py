"""-m gpu: the Qwen3-ASR path (log-mel frontend, conv stack, audio encoder, projector, audio-token scatter, text tower)
through the C ABI against the oracle restatement (oracle/qwen3_asr.py)."""
import numpy as np
import pytest
import torch

from aha_amd.configs import tiny_qwen3_asr
from aha_amd.weights import qwen3_asr_weights
from oracle import qwen3_asr as oa
from oracle.numerics import Numerics

pytestmark = pytest.mark.gpu
NM = Numerics("bf16", matmul_f64=True)


def synth_audio(n, seed):
    """BASELINE.md section 4 cfg 4: N(0, 0.1^2) clipped to [-1, 1], 16 kHz."""
    return np.clip(np.random.default_rng(seed).normal(0, 0.1, n), -1, 1).astype(np.float32)


def rel_err(got, ref):
    ref = np.asarray(ref, dtype=np.float32)
    return float(np.abs(got - ref).max()) / float(ref.std()), float(np.sqrt(((got - ref) ** 2).mean())) / float(ref.std())


@pytest.mark.parametrize("n", [16000, 48000, 480000, 16000 * 7 + 123])
def test_logmel_frontend(gpu, n):
    """A0 is f32 arithmetic: the only differences are the DFT summation order (direct 400-tap sums vs an FFT) and
    logf; bound 2e-4 absolute on the (x+4)/4 scale (values span ~[-0.5, 1.5])."""
    from aha_amd import ops
    wave = synth_audio(n, 4)
    ref = oa.log_mel(wave)
    got = ops.logmel(torch.from_numpy(wave).to(gpu)).cpu().numpy()
    assert got.shape == ref.shape == (128, n // 160)
    assert np.isfinite(got).all()
    assert float(np.abs(got - ref).max()) < 2e-4


def test_logmel_right_pad_quirk(gpu):
    """The last kept frame reaches 40 samples into the right pad, which the reference builds from x[L-400:L-200] reversed
    (tensor_utils.rs:525-549), not from a true reflection: a ramp signal makes the two differ visibly."""
    from aha_amd import ops
    wave = (np.linspace(-1, 1, 8000) ** 3).astype(np.float32)
    got = ops.logmel(torch.from_numpy(wave).to(gpu)).cpu().numpy()
    assert float(np.abs(got - oa.log_mel(wave)).max()) < 2e-4


@pytest.fixture(scope="module")
def asr(gpu):
    from aha_amd.model import HipInferenceModel
    cfg = tiny_qwen3_asr()
    w = qwen3_asr_weights(cfg, seed=0)
    m = HipInferenceModel(cfg, w)
    o = oa.OracleQwen3ASR(cfg, w, NM)
    yield cfg, m, o
    m.close()


def make_ids(cfg, n_audio_tok, seed):
    g = np.random.default_rng(seed)
    pre = [int(x) for x in g.integers(0, 1900, size=5)]
    post = [int(x) for x in g.integers(0, 1900, size=7)]
    return pre + [cfg.audio_start_token_id] + [cfg.audio_token_id] * n_audio_tok + [cfg.audio_end_token_id] + post


@pytest.mark.parametrize("seconds", [2.0, 5.37, 11.0])
def test_asr_prefill_and_decode_from_features(asr, seconds):
    from aha_amd.model import MultiModalData
    cfg, m, o = asr
    wave = synth_audio(int(16000 * seconds), 9)
    feats = oa.log_mel(wave)
    n_tok = oa.get_feat_extract_output_lengths(feats.shape[1])
    ids = make_ids(cfg, n_tok, 3)
    m.clear_cache(); o.clear_cache()
    got, am = m.forward_initial(ids, 0, MultiModalData(audio_features=feats))
    ref = o.forward_initial(ids, 0, torch.from_numpy(feats)).reshape(-1).numpy()
    e_max, e_rms = rel_err(m.debug_audio_embeds(n_tok), o.last_audio_embeds.numpy())
    assert e_max < 0.08 and e_rms < 0.02, f"audio embeds off: max {e_max:.4f} rms {e_rms:.4f} (std units)"
    l_max, l_rms = rel_err(got, ref)
    assert l_max < 0.05 and l_rms < 0.02, f"prefill logits off: max {l_max:.4f} rms {l_rms:.4f}"
    assert am == int(np.argmax(got))
    tok, off = int(np.argmax(ref)), len(ids)
    for step in range(4):
        got, _ = m.forward_step(tok, off)
        ref = o.forward_step([tok], off).reshape(-1).numpy()
        l_max, l_rms = rel_err(got, ref)
        assert l_max < 0.05 and l_rms < 0.02, f"decode step {step}"
        tok, off = int(np.argmax(ref)), off + 1


def test_asr_from_raw_samples_equals_features(asr):
    """Raw samples -> on-device log-mel -> same tokens/logits (within the frontend tolerance) as host features."""
    from aha_amd.model import MultiModalData
    cfg, m, o = asr
    wave = synth_audio(16000 * 3, 21)
    feats = oa.log_mel(wave)
    n_tok = oa.get_feat_extract_output_lengths(feats.shape[1])
    ids = make_ids(cfg, n_tok, 5)
    m.clear_cache()
    a, _ = m.forward_initial(ids, 0, MultiModalData(audio_features=feats))
    m.clear_cache()
    b, _ = m.forward_initial(ids, 0, MultiModalData(audio_samples=wave))
    assert float(np.abs(a - b).max()) <= 0.03 * float(a.std())


def test_asr_token_count_mismatch_is_an_error(asr):
    from aha_amd._lib import AhaHipError
    from aha_amd.model import MultiModalData
    cfg, m, o = asr
    feats = oa.log_mel(synth_audio(16000, 2))
    ids = make_ids(cfg, oa.get_feat_extract_output_lengths(feats.shape[1]) + 1, 1)
    m.clear_cache()
    with pytest.raises(AhaHipError, match="n_audio_tokens"):
        m.forward_initial(ids, 0, MultiModalData(audio_features=feats))
    m.clear_cache()


def test_asr_generate_loop_two_chunks_greedy(asr):
    """A4: the ASR model's own generate loop (qwen3_asr/generate.rs:130-186) through the C ABI, two audio chunks, greedy, against
    the oracle run chunk by chunk: same tokens wherever the oracle's margin is decisive, both eos ids honoured."""
    from aha_amd import sampling as hs
    from aha_amd.model import MultiModalData
    cfg, m, o = asr
    chunks, want = [], []
    for seed, secs in [(31, 2.5), (32, 4.0)]:
        wave = synth_audio(int(16000 * secs), seed)
        feats = oa.log_mel(wave)
        ids = make_ids(cfg, oa.get_feat_extract_output_lengths(feats.shape[1]), seed)
        chunks.append((ids, MultiModalData(audio_features=feats)))
        o.clear_cache()
        lg = o.forward_initial(ids, 0, torch.from_numpy(feats)).reshape(-1).numpy()
        toks, off = [int(np.argmax(lg))], len(ids)
        for _ in range(5):
            lg = o.forward_step([toks[-1]], off).reshape(-1).numpy()
            toks.append(int(np.argmax(lg)))
            off += 1
        want.append(toks)
    m.clear_cache()
    got, n_prompt = hs.generate_asr(m, chunks, temperature=0.0, max_tokens=6)
    assert n_prompt == sum(len(c[0]) for c in chunks) and len(got) == 12
    agree = sum(int(a == b) for a, b in zip(got, want[0] + want[1]))
    assert got[0] == want[0][0] and got[6] == want[1][0] and agree >= 10, (got, want)
    # an eos id stops the chunk after it has been pushed: make the first chunk's first token an eos id
    real = m.stop_token_ids
    m.stop_token_ids = lambda: [got[0], 1 << 30]
    got2, _ = hs.generate_asr(m, chunks, temperature=0.0, max_tokens=6)
    m.stop_token_ids = real
    assert got2[0] == got[0] and got2[1:7] == got[6:12]
    assert m.cache_len() == 0
