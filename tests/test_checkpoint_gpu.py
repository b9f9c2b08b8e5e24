"""-m gpu: aha_hip_model_load(<checkpoint dir>) == aha_hip_model_create(in-memory tensors): same logits, bit for bit, for
the three architectures, with the checkpoint split over several safetensors shards and stored in bf16 / f16 / f32."""
import numpy as np
import pytest
import torch

from aha_amd.checkpoint import save_checkpoint
from aha_amd.configs import tiny_qwen3, tiny_qwen3_asr, tiny_qwen3vl
from aha_amd.weights import qwen3_asr_weights, qwen3_text_weights, qwen3vl_weights

pytestmark = pytest.mark.gpu


def ids_for(cfg_vocab, n, seed):
    return [int(x) for x in np.random.default_rng(seed).integers(0, cfg_vocab, size=n)]


@pytest.mark.parametrize("store", [torch.bfloat16, torch.float32])
def test_text_model_from_directory(gpu, tmp_path, store):
    from aha_amd.model import HipInferenceModel
    cfg = tiny_qwen3(layers=2, hidden=256, heads=4, kv_heads=2, inter=512, vocab=1024, tie=False)
    w = qwen3_text_weights(cfg, seed=1)
    save_checkpoint(str(tmp_path), cfg, {k: v.to(store) for k, v in w.items()}, shards=3)
    a = HipInferenceModel(cfg, w)
    b = HipInferenceModel.from_pretrained(str(tmp_path))
    assert b.stop_token_ids() == a.stop_token_ids()
    ids = ids_for(cfg.vocab_size, 70, 3)
    la, ta = a.forward_initial(ids, 0)
    lb, tb = b.forward_initial(ids, 0)
    np.testing.assert_array_equal(la, lb)   # bf16 -> f32 -> bf16 is the identity, so the f32 store changes nothing
    assert a.decode_greedy(ta, 70, 12) == b.decode_greedy(tb, 70, 12)
    a.close(); b.close()


def test_vl_model_from_directory(gpu, tmp_path):
    from aha_amd.model import HipInferenceModel
    from aha_amd.vision_host import synthetic_image_request
    cfg = tiny_qwen3vl()
    w = qwen3vl_weights(cfg, seed=2)
    save_checkpoint(str(tmp_path), cfg, w, shards=2)
    a = HipInferenceModel(cfg, w)
    b = HipInferenceModel.from_pretrained(str(tmp_path))
    ids, mm = synthetic_image_request(cfg, 256, 7, torch.Generator().manual_seed(4), device=gpu)
    la, _ = a.forward_initial(ids, 0, mm)
    lb, _ = b.forward_initial(ids, 0, mm)
    np.testing.assert_array_equal(la, lb)
    a.close(); b.close()


def test_asr_model_from_directory(gpu, tmp_path):
    from aha_amd.model import HipInferenceModel, MultiModalData
    from oracle import qwen3_asr as oa
    cfg = tiny_qwen3_asr()
    w = qwen3_asr_weights(cfg, seed=3)
    save_checkpoint(str(tmp_path), cfg, w)
    a = HipInferenceModel(cfg, w)
    b = HipInferenceModel.from_pretrained(str(tmp_path))
    wave = np.clip(np.random.default_rng(1).normal(0, 0.1, 32000), -1, 1).astype(np.float32)
    n_tok = oa.get_feat_extract_output_lengths(len(wave) // 160)
    ids = [5, 6, cfg.audio_start_token_id] + [cfg.audio_token_id] * n_tok + [cfg.audio_end_token_id, 9]
    la, _ = a.forward_initial(ids, 0, MultiModalData(audio_samples=wave))
    lb, _ = b.forward_initial(ids, 0, MultiModalData(audio_samples=wave))
    np.testing.assert_array_equal(la, lb)
    a.close(); b.close()


def test_missing_tensor_in_directory_is_an_error(gpu, tmp_path):
    from aha_amd._lib import AhaHipError
    from aha_amd.model import HipInferenceModel
    cfg = tiny_qwen3()
    w = qwen3_text_weights(cfg, seed=1)
    w.pop("model.layers.1.mlp.down_proj.weight")
    save_checkpoint(str(tmp_path), cfg, w)
    with pytest.raises(AhaHipError, match="down_proj"):
        HipInferenceModel.from_pretrained(str(tmp_path))


def test_plain_c_host_over_the_abi(gpu, tmp_path):
    """examples/c_harness.c: a C program (no Python, no torch, no HIP headers) loads the checkpoint directory through
    the C ABI and runs the reference's greedy loop host-driven and device-resident; both must equal the Python mirror."""
    import os
    import subprocess
    from aha_amd._lib import LIB_PATH
    from aha_amd.model import HipInferenceModel, generate_generic
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_harness")
    libdir = os.path.dirname(LIB_PATH)
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", "c_harness.c"), "-o", exe, "-L" + libdir, "-laha_hip",
                    "-Wl,-rpath," + libdir, "-lm"], check=True)
    cfg = tiny_qwen3(layers=2, hidden=256, heads=4, kv_heads=2, inter=512, vocab=1024, tie=True)
    w = qwen3_text_weights(cfg, seed=4)
    ckpt = tmp_path / "ckpt"
    save_checkpoint(str(ckpt), cfg, w, shards=2)
    ids = ids_for(cfg.vocab_size, 33, 8)
    out = subprocess.run([exe, str(ckpt), "24"] + [str(i) for i in ids], check=True, capture_output=True, text=True,
                         timeout=300).stdout.strip().splitlines()
    host = [int(x) for x in out[0].split()[1:]]
    dev = [int(x) for x in out[1].split()[1:]]
    assert out[0].startswith("host:") and out[1].startswith("device:")
    m = HipInferenceModel(cfg, w)
    want, _ = generate_generic(m, ids, 24, device_loop=False)
    assert host == dev == want
    # the sampled path and the resampler from C agree with the Python mirror over the same ABI
    m.forward_initial(ids, 0, want_logits=False)
    _, idx, _, _ = m.sample_candidates(want, 1.1, 0.6, 4)
    m.close()
    assert out[2].split() == ["candidates:"] + [str(int(i)) for i in idx]
    # the draw from C: the weights it used (printed as hex floats) through the Python mirror's StdRng and through the independent
    # restatement oracle/rand_stdrng.py give the same eight tokens of the reference's default-seed stream
    from aha_amd import sampling as hs
    from oracle import rand_stdrng as R
    wts = np.array([float.fromhex(x) for x in out[3].split()[1:]], dtype=np.float32)
    assert out[3].startswith("weights:") and wts.shape == (4,) and (wts > 0).all() and (np.diff(wts) <= 0).all()   # softmax values of the ranked candidates
    rng, ref = hs.StdRng(299792458), R.StdRng.seed_from_u64(299792458)
    want_draws = [int(idx[rng.weighted_index(wts)]) for _ in range(8)]
    assert want_draws == [int(idx[R.sample_multinomial(ref, wts)]) for _ in range(8)]
    assert out[4].split() == ["draws:"] + [str(t) for t in want_draws]
    assert out[5].split() == ["resampled:", "16000"]
