"""Worker of tests/test_model_gpu.py::test_norm_inside_the_split_k_reduce_is_bit_identical: a text prefill (hidden 1024, 2 + 1
layers) and a Qwen3-VL prefill with DeepStack adds (no fusion across them), each followed by a decode step, with every GEMM forced
onto the 256^2 tile in 2 K slices so that o_proj / down_proj end in the reduce pass; prints a digest of the logits.  Run with
AHA_GEMM_FUSE_NORM=0 (reduce pass, then rmsnorm_rows_kernel) and =1 (gemm_splitk_reduce_norm_kernel): the digests must be equal."""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from aha_amd import ops
    from aha_amd.configs import tiny_qwen3, tiny_qwen3vl
    from aha_amd.model import HipInferenceModel
    from aha_amd.weights import qwen3_text_weights, qwen3vl_weights
    from aha_amd.vision_host import synthetic_image_request
    h = hashlib.sha256()
    ops.gemm_plan(256, 2)
    cfg = tiny_qwen3(layers=3, hidden=1024, heads=8, kv_heads=2, inter=2048, vocab=2048)
    m = HipInferenceModel(cfg, qwen3_text_weights(cfg, seed=5))
    ids = [int(x) for x in np.random.default_rng(2).integers(0, cfg.vocab_size, size=300)]
    lg, tok = m.forward_initial(ids, 0)
    h.update(lg.tobytes())
    lg, _ = m.forward_step(tok, len(ids))
    h.update(lg.tobytes())
    m.close()
    vcfg = tiny_qwen3vl()
    vm = HipInferenceModel(vcfg, qwen3vl_weights(vcfg, seed=0))
    vids, data = synthetic_image_request(vcfg, 128, 230, torch.Generator().manual_seed(4))
    lg, tok = vm.forward_initial(vids, 0, data)
    h.update(lg.tobytes())
    lg, _ = vm.forward_step(tok, len(vids))
    h.update(lg.tobytes())
    vm.close()
    ops.gemm_plan(0, 0)
    # round 6: the ViT block's LayerNorms ride on its GEMM calls (norm2 on proj, norm1 of the next block on fc2: csrc/vision_tower.hip,
    # kernels_gemm.hip gemm_splitk_reduce_layernorm_kernel).  The vision tower at its REAL widths (1152 / 16 heads / 4304), 4 blocks with one
    # DeepStack merger in the middle (after a merger the next norm1 is its own launch), 1024 patches: once with every GEMM forced onto two K
    # slices (reduce pass everywhere), once with the automatic plans.
    from aha_amd.configs import Qwen3VLConfig, Qwen3VLVisionConfig
    wide = Qwen3VLConfig(text=vcfg.text, vision=Qwen3VLVisionConfig(depth=4, out_hidden_size=256, deepstack_visual_indexes=[1]),
                         image_token_id=2000, video_token_id=2001, vision_start_token_id=2002, vision_end_token_id=2003)
    ww = qwen3vl_weights(wide, seed=3)
    for plan in ((256, 2), (0, 0)):
        ops.gemm_plan(*plan)
        wm = HipInferenceModel(wide, ww)
        wids, wdata = synthetic_image_request(wide, 512, 40, torch.Generator().manual_seed(6))
        lg, tok = wm.forward_initial(wids, 0, wdata)
        h.update(lg.tobytes())
        h.update(np.asarray(wm.debug_image_embeds(0, 256)).tobytes())
        h.update(np.asarray(wm.debug_image_embeds(1, 256)).tobytes())
        wm.close()
    ops.gemm_plan(0, 0)
    # round 6: the audio tower's LayerNorm behind fc2 + residual (norm1 of the next layer, ln_post after the last) rides on the fc2 call
    # (csrc/audio_tower.hip; AHA_AUD_FUSE_LN=0: its own launch).  A Qwen3-ASR prefill with every GEMM forced onto two K slices (reduce pass
    # everywhere) and once more with the 128^2 ring kernel's K slices; audio embeddings and logits into the digest.
    from aha_amd.configs import tiny_qwen3_asr
    from aha_amd.model import MultiModalData
    from aha_amd.weights import qwen3_asr_weights
    from oracle import qwen3_asr as oa   # (host-side feature extraction for the request only: the digest compares HIP runs with each other)
    acfg = tiny_qwen3_asr(layers=3)
    aw = qwen3_asr_weights(acfg, seed=2)
    wave = np.clip(np.random.default_rng(11).normal(0, 0.1, 16000 * 4), -1, 1).astype(np.float32)
    feats = oa.log_mel(wave)
    n_tok = oa.get_feat_extract_output_lengths(feats.shape[1])
    aids = [5, 6, 7, acfg.audio_start_token_id] + [acfg.audio_token_id] * n_tok + [acfg.audio_end_token_id, 8, 9]
    for plan in ((256, 2), (128, 2), (0, 0)):
        ops.gemm_plan(*plan)
        am = HipInferenceModel(acfg, aw)
        lg, tok = am.forward_initial(aids, 0, MultiModalData(audio_features=feats))
        h.update(lg.tobytes())
        h.update(np.asarray(am.debug_audio_embeds(n_tok)).tobytes())
        am.close()
    ops.gemm_plan(0, 0)
    print("FUSE_NORM_DIGEST", h.hexdigest(), flush=True)


if __name__ == "__main__":
    main()
