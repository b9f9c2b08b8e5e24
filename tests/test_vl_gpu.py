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


@pytest.mark.parametrize("thw", [(2, 64, 64), (5, 96, 64), (3, 32, 160), (1, 64, 32)])
def test_video_to_patches_bit_exact(gpu, thw):
    """process_videos normalises in the model dtype op by op (u8 -> bf16, * bf16(1/255), - mean, / std, each rounded), pairs
    consecutive frames and repeats an odd last frame: elementwise, so bit-identical to the restatement."""
    from aha_amd import ops
    g = np.random.default_rng(6)
    vid = g.integers(0, 256, size=(thw[0], thw[1], thw[2], 3), dtype=np.uint8)
    ref, grid = ov.process_videos(NM, [vid])
    got = ops.video_to_patches(torch.from_numpy(vid).to(gpu))
    assert tuple(grid[0]) == ((thw[0] + 1) // 2, thw[1] // 16, thw[2] // 16) and got.shape == ref.shape
    assert torch.equal(got.float().cpu(), ref)
    # not the image path's arithmetic: the f32 chain with one final rounding gives different bits somewhere
    img_like = ov.process_images(NM, [vid[0]])[0]
    if thw[0] == 1:
        assert not torch.equal(ref, img_like)


def make_video_request(cfg, img_sizes, vid_shapes, n_text, seed):
    """Prompt in the processor's layout (processor.rs:386-431): images, then per video and temporal patch some timestamp tokens,
    <|vision_start|>, h*w/4 video pads, <|vision_end|>."""
    from aha_amd.vision_host import image_prompt_ids, video_prompt_ids
    g = np.random.default_rng(seed)
    imgs = [g.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for (h, w) in img_sizes]
    vids = [g.integers(0, 256, size=(t, h, w, 3), dtype=np.uint8) for (t, h, w) in vid_shapes]
    pv, grid = ov.process_images(NM, imgs) if imgs else (None, None)
    pvv, vgrid = ov.process_videos(NM, vids)
    ids = [int(x) for x in g.integers(0, 1900, size=3)]
    if imgs:
        ids = image_prompt_ids(cfg, grid, ids, [int(x) for x in g.integers(0, 1900, size=2)])
    stamps = [[int(x) for x in g.integers(0, 1900, size=3)] for _ in range(int(vgrid[:, 0].sum()))]
    ids += video_prompt_ids(cfg, vgrid, stamps)
    ids += [int(x) for x in g.integers(0, 1900, size=n_text)]
    return imgs, vids, pv, grid, pvv, vgrid, ids


@pytest.mark.parametrize("img_sizes,vid_shapes", [([(64, 96)], [(5, 64, 64)]), ([], [(4, 96, 64)]), ([(64, 64), (32, 96)], [(2, 64, 64), (3, 32, 64)])])
def test_vl_video_prefill_and_decode(vl, gpu, img_sizes, vid_shapes):
    """Videos (model.rs:1169-1225): the frames' patch rows through the same tower, scattered at the <|video_pad|> rows, DeepStack
    rows of images and videos joined in position order, one (1, h, w) M-RoPE block per temporal patch.  The library encodes images
    and videos in one pass (row-wise ops + per-frame attention segments), the oracle in two like the reference."""
    from aha_amd.model import MultiModalData
    cfg, m, o = vl
    imgs, vids, pv, grid, pvv, vgrid, ids = make_video_request(cfg, img_sizes, vid_shapes, 7, 29)
    m.clear_cache(); o.clear_cache()
    data = MultiModalData(pv.to(torch.bfloat16) if pv is not None else None, grid, pixel_values_video=pvv.to(torch.bfloat16), video_grid_thw=vgrid)
    got, am = m.forward_initial(ids, 0, data)
    ref = o.forward_initial(ids, 0, (pv, grid, pvv, vgrid)).reshape(-1).numpy()
    ref_emb = [o.last_video_embeds] if pv is None else [o.last_image_embeds, o.last_video_embeds]
    ref_emb = torch.cat(ref_emb, 0).numpy()
    emb = m.debug_image_embeds(0, ref_emb.shape[0])
    e_max, e_rms = rel_err(emb, ref_emb)
    assert e_max < 0.08 and e_rms < 0.02, f"visual embeds off: max {e_max:.4f} rms {e_rms:.4f} (in std units)"
    for k in range(len(cfg.vision.deepstack_visual_indexes)):
        rd = o.last_video_deepstack[k] if pv is None else torch.cat([o.last_deepstack[k], o.last_video_deepstack[k]], 0)
        d_max, d_rms = rel_err(m.debug_image_embeds(k + 1, ref_emb.shape[0]), rd.numpy())
        assert d_max < 0.08 and d_rms < 0.02, f"deepstack {k} off: max {d_max:.4f} rms {d_rms:.4f}"
    l_max, l_rms = rel_err(got, ref)
    assert l_max < 0.05 and l_rms < 0.02, f"prefill logits off: max {l_max:.4f} rms {l_rms:.4f}"
    assert am == int(np.argmax(got))
    tok, off = int(np.argmax(ref)), len(ids)
    for step in range(4):
        got, _ = m.forward_step(tok, off)
        ref = o.forward_step([tok], off).reshape(-1).numpy()
        l_max, l_rms = rel_err(got, ref)
        assert l_max < 0.05 and l_rms < 0.02, f"decode step {step}: max {l_max:.4f} rms {l_rms:.4f}"
        tok, off = int(np.argmax(ref)), off + 1
    assert o.rope_delta < 0
    # the frames patchified on the GPU give the same logits as the host-side rows
    from aha_amd import vision_host
    m.clear_cache()
    a, _ = m.forward_initial(ids, 0, data)
    pvv_dev, vgrid_dev = vision_host.process_videos([torch.from_numpy(v).to(gpu) for v in vids], cfg)
    assert np.array_equal(vgrid_dev, vgrid)
    m.clear_cache()
    b, _ = m.forward_initial(ids, 0, MultiModalData(pv.to(torch.bfloat16) if pv is not None else None, grid, pixel_values_video=pvv_dev, video_grid_thw=vgrid))
    assert np.array_equal(a, b)
    # encode-only + precomputed rows (the image-parallel entry) with videos in the request
    emb_dev = m.vision_encode(data)
    m.clear_cache()
    c, _ = m.forward_initial(ids, 0, MultiModalData(image_grid_thw=grid, image_embeds=emb_dev, video_grid_thw=vgrid))
    assert np.array_equal(a, c)
    m.clear_cache()


def test_vl_video_token_count_mismatch_is_an_error(vl):
    """model.rs:1176-1183: the number of <|video_pad|> tokens must equal the video rows."""
    from aha_amd._lib import AhaHipError
    from aha_amd.model import MultiModalData
    cfg, m, o = vl
    imgs, vids, pv, grid, pvv, vgrid, ids = make_video_request(cfg, [], [(2, 64, 64)], 3, 31)
    ids.remove(cfg.video_token_id)
    m.clear_cache()
    with pytest.raises(AhaHipError, match="n_image_token"):
        m.forward_initial(ids, 0, MultiModalData(pixel_values_video=pvv.to(torch.bfloat16), video_grid_thw=vgrid))
    m.clear_cache()


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


def test_vit_attention_score_chain_variants_give_the_same_bits(vl, gpu):
    """head_dim 72 (96 / 80 padded) instantiations of the prefill attention: the matrix-pipe score chain (csrc/attn_common.h
    mfma_diag) against the vector-ALU one through the whole tower -- bit-identical merged embeddings and DeepStack taps."""
    from aha_amd import ops
    from aha_amd.model import MultiModalData
    cfg, m, o = vl
    imgs, pv, grid, ids = make_request(cfg, [(160, 96), (64, 128)], 6, 29)
    embs = {}
    try:
        for smx in (0, 1):
            ops.attn_variant(smx)
            embs[smx] = m.vision_encode(MultiModalData(pv.to(torch.bfloat16), grid))
    finally:
        ops.attn_variant(-1)
    assert torch.equal(embs[1], embs[0])


@pytest.mark.parametrize("smx", [0, 1, 3])
def test_vit_tower_on_every_score_chain_variant(vl, gpu, smx):
    """head_dim 72 (Q / K rows padded to 96, V block to 80) on every score chain of the prefill attention -- 0 / 1 = the reference's
    rounding chain (vector ALU / scale multiply on the matrix pipe), 3 = the f32 score chain (round 5 default) -- against the oracle's
    tower, segments of 60 / 32 / 196 patches, the same bound as every other tower test."""
    from aha_amd import ops
    from aha_amd.model import MultiModalData
    cfg, m, o = vl
    imgs, pv, grid, ids = make_request(cfg, [(160, 96), (64, 128), (224, 224)], 6, 31)
    o.clear_cache()
    o.forward_initial(ids, 0, (pv, grid))
    ref = o.last_image_embeds.numpy()
    try:
        ops.attn_variant(smx)
        emb = m.vision_encode(MultiModalData(pv.to(torch.bfloat16), grid))
    finally:
        ops.attn_variant(-1)
    e_max, e_rms = rel_err(emb[0].float().cpu().numpy(), ref)
    assert e_max < 0.08 and e_rms < 0.02, f"smx {smx}: visual embeds off: max {e_max:.4f} rms {e_rms:.4f} (in std units)"
    for k in range(len(cfg.vision.deepstack_visual_indexes)):
        d_max, d_rms = rel_err(emb[k + 1].float().cpu().numpy(), o.last_deepstack[k].numpy())
        assert d_max < 0.08 and d_rms < 0.02, f"smx {smx}: deepstack {k} off: max {d_max:.4f} rms {d_rms:.4f}"
