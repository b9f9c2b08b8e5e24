"""Host side of the Qwen3-VL request path: what Qwen3VLProcessor::process_info does around the tensors
(/root/reference/src/models/qwen3vl/processor.rs:310-444): smart-resize arithmetic, placeholder expansion, and the
hand-off of the image to the V0 kernel.  No tensor math happens here."""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np
import torch

from . import ops
from .configs import Qwen3VLConfig
from .model import MultiModalData


def img_smart_resize(h: int, w: int, factor: int = 32, min_pixels: int = 65536, max_pixels: int = 16777216) -> Tuple[int, int]:
    """img_smart_resize, /root/reference/src/utils/img_utils.rs:295-331 (f32 beta, round/floor/ceil by factor)."""
    if max(h, w) // min(h, w) > 200:
        raise ValueError(f"absolute aspect ratio mush be smaller than 200, got {max(h, w) // min(h, w)}")
    def rnd(v):  # round_by_factor (utils/mod.rs:392-395): f32 quotient, f32::round = half AWAY from zero (Python's round is half-even)
        q = np.float32(v) / np.float32(factor)
        fl = np.floor(q)
        return int(fl + (q - fl >= np.float32(0.5))) * factor
    h_bar, w_bar = max(factor, rnd(h)), max(factor, rnd(w))
    if h_bar * w_bar > max_pixels:
        beta = np.sqrt(np.float32(h * w) / np.float32(max_pixels), dtype=np.float32)
        h_bar = max(factor, int(math.floor(np.float32(h) / beta / factor)) * factor)
        w_bar = max(factor, int(math.floor(np.float32(w) / beta / factor)) * factor)
    elif h_bar * w_bar < min_pixels:
        beta = np.sqrt(np.float32(min_pixels) / np.float32(h * w), dtype=np.float32)
        h_bar = int(math.ceil(np.float32(h) * beta / factor)) * factor
        w_bar = int(math.ceil(np.float32(w) * beta / factor)) * factor
    return h_bar, w_bar


def process_images(imgs_u8: List[torch.Tensor], cfg: Qwen3VLConfig) -> MultiModalData:
    """process_images (processor.rs:229-251) on the GPU: process_img's smart-resize + CatmullRom resize_exact (processor.rs:159-166,
    `ops.image_resize`) when the image is not already at its target size, then the V0 patchify kernel."""
    v = cfg.vision
    pv, grids = [], []
    for im in imgs_u8:
        H, W = int(im.shape[0]), int(im.shape[1])
        th, tw = img_smart_resize(H, W, v.patch_size * v.spatial_merge_size)
        if (H, W) != (th, tw):
            im = ops.image_resize(im, th, tw)
            H, W = th, tw
        pv.append(ops.image_to_patches(im, v.patch_size, v.spatial_merge_size))
        grids.append([1, H // v.patch_size, W // v.patch_size])
    return MultiModalData(torch.cat(pv, 0), np.asarray(grids, dtype=np.uint32))


def image_prompt_ids(cfg: Qwen3VLConfig, grids: np.ndarray, prefix: List[int], suffix: List[int]) -> List[int]:
    """prefix + for each image: <|vision_start|> + N/4 x <|image_pad|> + <|vision_end|> (processor.rs:386-399) + suffix."""
    m2 = cfg.vision.spatial_merge_size ** 2
    ids = list(prefix)
    for g in np.asarray(grids).tolist():
        ids += [cfg.vision_start_token_id] + [cfg.image_token_id] * (g[0] * g[1] * g[2] // m2) + [cfg.vision_end_token_id]
    return ids + list(suffix)


def synthetic_image_request(cfg: Qwen3VLConfig, image_px: int, prompt_tokens: int, gen: torch.Generator, device="cuda",
                            n_images: int = 1):
    """BASELINE.md section 4 cfg 3 (one image) / cfg 5 (n_images = 8 at 2048^2): uniform-random uint8 images + 4 template
    tokens + prompt_tokens random text ids."""
    imgs = [torch.randint(0, 256, (image_px, image_px, 3), generator=gen, dtype=torch.uint8).to(device) for _ in range(n_images)]
    data = process_images(imgs, cfg)
    hi = min(cfg.text.vocab_size, 151643)
    prefix = torch.randint(0, hi, (4,), generator=gen).tolist()
    suffix = torch.randint(0, hi, (prompt_tokens,), generator=gen).tolist()
    return image_prompt_ids(cfg, data.image_grid_thw, prefix, suffix), data


def get_rope_index(cfg: Qwen3VLConfig, input_ids, image_grid_thw):
    """Qwen3VLModel::get_rope_index through the library's host code (csrc/vision.hip rope_index_core): (3, S) int32
    positions and rope_delta.  No GPU involved."""
    import ctypes as C
    from ._lib import check, lib
    from .model import make_desc
    ids = np.ascontiguousarray(np.asarray(input_ids, dtype=np.uint32).reshape(-1))
    grid = np.ascontiguousarray(np.asarray(image_grid_thw, dtype=np.uint32).reshape(-1, 3))
    pos = np.empty((3, ids.size), dtype=np.int32)
    delta = C.c_int64()
    desc = make_desc(cfg)
    check(lib().aha_hip_get_rope_index(C.byref(desc), ids.ctypes.data_as(C.c_void_p), ids.size, grid.ctypes.data_as(C.c_void_p),
                                       grid.shape[0], pos.ctypes.data_as(C.c_void_p), C.byref(delta)))
    return pos, int(delta.value)
