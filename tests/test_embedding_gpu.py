"""-m gpu: Qwen3-Embedding / Qwen3-Reranker entry point (aha_hip_embed) against the oracle restatement
(oracle/qwen3.py embed_one / rerank; reference: qwen3_embedding/mod.rs:50-64, qwen3_reranker/mod.rs:23-31)."""
import numpy as np
import pytest
import torch

from aha_amd.configs import tiny_qwen3
from aha_amd.weights import qwen3_text_weights
from oracle import qwen3 as oq
from oracle.numerics import Numerics

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pair(gpu):
    from aha_amd.model import HipInferenceModel
    cfg = tiny_qwen3(layers=3, hidden=512, heads=4, kv_heads=2, inter=1024, vocab=4096)
    w = qwen3_text_weights(cfg, seed=0)
    m = HipInferenceModel(cfg, w)
    o = oq.OracleQwen3(cfg, w, Numerics("bf16", matmul_f64=True))
    yield cfg, m, o
    m.close()


def seqs_for(cfg, lens, seed):
    g = np.random.default_rng(seed)
    return [[int(x) for x in g.integers(0, cfg.vocab_size, size=n)] for n in lens]


def test_embed_matches_oracle(pair):
    """Both sides hold the pooled hidden state in bf16 and normalise in f32; the hidden state itself differs by f32
    accumulation order upstream (same tolerance class as the logits tests): per element <= 0.04 of the vector's rms,
    cosine to the oracle embedding >= 0.9995, unit norm to 1e-5."""
    cfg, m, o = pair
    for ids in seqs_for(cfg, [1, 7, 64, 65, 300], 1):
        got = m.embed_one(ids)
        ref = oq.embed_one(o, ids).numpy()
        assert got.shape == ref.shape == (cfg.hidden_size,)
        assert abs(float(np.linalg.norm(got)) - 1.0) < 1e-5
        rms = float(np.sqrt((ref ** 2).mean()))
        assert float(np.abs(got - ref).max()) <= 0.04 * rms
        assert float(got @ ref) >= 0.9995
        assert m.cache_len() == 0   # clear_kv_cache after every embedding (qwen3_embedding/mod.rs:58)


def test_embed_is_prefix_independent_and_deterministic(pair):
    """The cache is cleared before and after: embedding B after A equals embedding B alone, bit for bit."""
    cfg, m, o = pair
    a, b = seqs_for(cfg, [40, 90], 2)
    m.forward_initial(a, 0)   # leave a dirty cache on purpose
    e1 = m.embed_one(b)
    m.embed_one(a)
    e2 = m.embed_one(b)
    np.testing.assert_array_equal(e1, e2)


def test_rerank_matches_oracle(pair):
    cfg, m, o = pair
    q, *docs = seqs_for(cfg, [12, 30, 5, 77], 3)
    got = m.rerank(q, docs)
    ref = oq.rerank(o, q, docs).numpy()
    assert got.shape == ref.shape == (3,)
    assert float(np.abs(got - ref).max()) < 2e-3
    assert abs(float(m.rerank(q, [q])[0]) - 1.0) < 1e-5   # a document identical to the query scores 1


def test_embed_errors(pair):
    from aha_amd._lib import AhaHipError
    cfg, m, o = pair
    with pytest.raises(AhaHipError, match="empty"):
        m.embed_one([])
    with pytest.raises(ValueError, match="cannot be empty"):
        m.embed_multi([])
    with pytest.raises(AhaHipError, match="out of range"):
        m.embed_one([cfg.vocab_size])
