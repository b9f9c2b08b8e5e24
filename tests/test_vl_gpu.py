"""-m gpu: the Qwen3-VL path (V0 patchify, ViT, scatter, DeepStack, interleaved M-RoPE, rope_delta decode) through the
C ABI against the oracle restatement (oracle/qwen3vl.py), on a tiny model with the real head dims (72 / 128)."""
import numpy as np
import pytest
import torch

from aha_amd.configs import tiny_qwen3vl
from aha_amd.weights import qwen3vl_weights
from oracle.numerics import Numerics
from oracle import qwen3vl as ov

pytestmark = pytest.mark.gpu
NM = Numerics("bf16", matmul_f64=True)


def rel_err(got, ref):
    ref = np.asarray(ref, dtype=np.float32)
    return float(np.abs(got - ref).max()) / float(ref.std()), float(np.sqrt(((got - ref) ** 2).mean())) / float(ref.std())


def make_request(cfg, sizes, n_text, seed):
    g = np.random.default_rng(seed)
    imgs = [g.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for (h, w) in sizes]
    pv, grid = ov.process_images(NM, imgs)
    ids = [int(x) for x in g.integers(0, 1900, size=3)]
    for gi in grid.tolist():
        ids += [cfg.vision_start_token_id] + [cfg.image_token_id] * (gi[0] * gi[1] * gi[2] // 4) + [cfg.vision_end_token_id]
        ids += [int(x) for x in g.integers(0, 1900, size=2)]
    ids += [int(x) for x in g.integers(0, 1900, size=n_text)]
    return imgs, pv, grid, ids


@pytest.fixture(scope="module")
def vl(gpu):
    from aha_amd.model import HipInferenceModel
    cfg = tiny_qwen3vl()
    w = qwen3vl_weights(cfg, seed=0)
    m = HipInferenceModel(cfg, w)
    o = ov.OracleQwen3VL(cfg, w, NM)
    yield cfg, m, o
    m.close()


@pytest.mark.parametrize("hw", [(64, 64), (96, 160), (224, 32)])
def test_image_to_patches_bit_exact(gpu, hw):
    """V0 is elementwise f32 arithmetic with one rounding: must be bit-identical to the restatement."""
    from aha_amd import ops
    g = np.random.default_rng(5)
    img = g.integers(0, 256, size=(hw[0], hw[1], 3), dtype=np.uint8)
    ref, grid = ov.process_images(NM, [img])
    got = ops.image_to_patches(torch.from_numpy(img).to(gpu))
    assert got.shape == ref.shape and tuple(grid[0]) == (1, hw[0] // 16, hw[1] // 16)
    assert torch.equal(got.float().cpu(), ref)


@pytest.mark.parametrize("sizes", [[(64, 64)], [(96, 160)], [(160, 96), (64, 128)]])
def test_vl_prefill_and_decode(vl, gpu, sizes):
    from aha_amd.model import MultiModalData
    cfg, m, o = vl
    imgs, pv, grid, ids = make_request(cfg, sizes, 9, 11)
    m.clear_cache(); o.clear_cache()
    data = MultiModalData(pv.to(torch.bfloat16), grid)
    got, am = m.forward_initial(ids, 0, data)
    ref = o.forward_initial(ids, 0, (pv, grid)).reshape(-1).numpy()
    # intermediate tensors: image embeddings and DeepStack features
    n4 = int(sum(g[0] * g[1] * g[2] for g in grid.tolist()) // 4)
    emb = m.debug_image_embeds(0, n4)
    e_max, e_rms = rel_err(emb, o.last_image_embeds.numpy())
    assert e_max < 0.08 and e_rms < 0.02, f"image embeds off: max {e_max:.4f} rms {e_rms:.4f} (in std units)"
    for k in range(len(cfg.vision.deepstack_visual_indexes)):
        d_max, d_rms = rel_err(m.debug_image_embeds(k + 1, n4), o.last_deepstack[k].numpy())
        assert d_max < 0.08 and d_rms < 0.02, f"deepstack {k} off: max {d_max:.4f} rms {d_rms:.4f}"
    l_max, l_rms = rel_err(got, ref)
    assert l_max < 0.05 and l_rms < 0.02, f"prefill logits off: max {l_max:.4f} rms {l_rms:.4f}"
    assert am == int(np.argmax(got))
    # decode with rope_delta positions (qwen3vl/model.rs:1235-1264), teacher-forced on the oracle's tokens
    tok, off = int(np.argmax(ref)), len(ids)
    for step in range(6):
        got, _ = m.forward_step(tok, off)
        ref = o.forward_step([tok], off).reshape(-1).numpy()
        l_max, l_rms = rel_err(got, ref)
        assert l_max < 0.05 and l_rms < 0.02, f"decode step {step}: max {l_max:.4f} rms {l_rms:.4f}"
        tok, off = int(np.argmax(ref)), off + 1
    assert o.rope_delta < 0  # images compress positions


def test_vl_device_pixel_values_and_v0(vl, gpu):
    """Same request with pixel_values produced on the GPU by the V0 kernel: identical logits to host pixel_values."""
    from aha_amd import vision_host
    from aha_amd.model import MultiModalData
    cfg, m, o = vl
    imgs, pv, grid, ids = make_request(cfg, [(256, 256)], 5, 13)
    m.clear_cache()
    a, _ = m.forward_initial(ids, 0, MultiModalData(pv.to(torch.bfloat16), grid))
    m.clear_cache()
    data = vision_host.process_images([torch.from_numpy(imgs[0]).to(gpu)], cfg)
    assert np.array_equal(data.image_grid_thw, grid)
    b, _ = m.forward_initial(ids, 0, data)
    assert np.array_equal(a, b)


def test_vl_text_only_positions(vl):
    """No image: M-RoPE rows are all arange, delta 0 (model.rs:1119-1132) -- must equal the oracle too."""
    cfg, m, o = vl
    ids = [int(x) for x in np.random.default_rng(3).integers(0, 1900, size=70)]
    m.clear_cache(); o.clear_cache()
    got, _ = m.forward_initial(ids, 0)
    ref = o.forward_initial(ids, 0).reshape(-1).numpy()
    l_max, l_rms = rel_err(got, ref)
    assert l_max < 0.05 and l_rms < 0.02
    got, _ = m.forward_step(7, 70)
    ref = o.forward_step([7], 70).reshape(-1).numpy()
    l_max, l_rms = rel_err(got, ref)
    assert l_max < 0.05 and l_rms < 0.02


def test_vl_token_count_mismatch_is_an_error(vl):
    """qwen3vl/model.rs:1158-1164: n_image_token != image_embed len -> Err."""
    from aha_amd._lib import AhaHipError
    from aha_amd.model import MultiModalData
    cfg, m, o = vl
    imgs, pv, grid, ids = make_request(cfg, [(64, 64)], 4, 17)
    ids.remove(cfg.image_token_id)
    m.clear_cache()
    with pytest.raises(AhaHipError, match="n_image_token"):
        m.forward_initial(ids, 0, MultiModalData(pv.to(torch.bfloat16), grid))
    m.clear_cache()


def test_vision_encode_then_precomputed_embeds(vl, gpu):
    """aha_hip_vision_encode + forward_initial(image_embeds=...) (the image-parallel path on one GPU) must reproduce the
    fused forward bit for bit."""
    from aha_amd.model import MultiModalData
    cfg, m, o = vl
    imgs, pv, grid, ids = make_request(cfg, [(160, 96), (64, 128)], 6, 23)
    m.clear_cache()
    a, _ = m.forward_initial(ids, 0, MultiModalData(pv.to(torch.bfloat16), grid))
    emb = m.vision_encode(MultiModalData(pv.to(torch.bfloat16), grid))
    assert emb.shape[0] == 1 + len(cfg.vision.deepstack_visual_indexes)
    m.clear_cache()
    b, _ = m.forward_initial(ids, 0, MultiModalData(image_grid_thw=grid, image_embeds=emb))
    assert np.array_equal(a, b)
    m.clear_cache()
