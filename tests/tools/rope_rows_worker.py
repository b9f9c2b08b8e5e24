"""Worker of tests/test_model_gpu.py::test_row_vectorised_rope_kernel_is_bit_identical: a chunked prefill (second chunk starting in
the middle of a KV page), an M-RoPE VL prefill and a decode step; prints a digest of every logits vector.  Run twice, with
AHA_ROPE_ROWS=0 (per-element kernel) and =1 (row-vectorised kernel): the digests must be equal."""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from aha_amd.configs import tiny_qwen3, tiny_qwen3vl
    from aha_amd.model import HipInferenceModel, MultiModalData
    from aha_amd.weights import qwen3_text_weights, qwen3vl_weights
    from aha_amd.vision_host import synthetic_image_request
    h = hashlib.sha256()
    cfg = tiny_qwen3(layers=2, hidden=512, heads=6, kv_heads=2, inter=1024, vocab=2048)   # 6 + 2 head slots: a ragged slot chunk
    m = HipInferenceModel(cfg, qwen3_text_weights(cfg, seed=3))
    ids = [int(x) for x in np.random.default_rng(1).integers(0, cfg.vocab_size, size=277)]
    lg, _ = m.forward_initial(ids[:200], 0)
    h.update(lg.tobytes())
    lg, tok = m.forward_initial(ids[200:], 200)      # kv_start = 200: page 3 is entered at slot 8
    h.update(lg.tobytes())
    lg, _ = m.forward_step(tok, 277)
    h.update(lg.tobytes())
    m.close()
    vcfg = tiny_qwen3vl()
    vm = HipInferenceModel(vcfg, qwen3vl_weights(vcfg, seed=0))
    vids, data = synthetic_image_request(vcfg, 96, 40, torch.Generator().manual_seed(2))
    lg, tok = vm.forward_initial(vids, 0, data)
    h.update(lg.tobytes())
    lg, _ = vm.forward_step(tok, len(vids))
    h.update(lg.tobytes())
    vm.close()
    print("ROPE_ROWS_DIGEST", h.hexdigest(), flush=True)


if __name__ == "__main__":
    main()
