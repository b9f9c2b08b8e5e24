"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy f32) of the image resize in front of the Qwen3-VL patchifier.  Nothing
under aha_amd/ may import this module.

Path: `Qwen3VLProcessor::process_img` (reference src/models/qwen3vl/processor.rs:150-171): `img_smart_resize`
(src/utils/img_utils.rs:294-331) then `DynamicImage::resize_exact(w, h, FilterType::CatmullRom)`.

PARITY UNPINNED, and the restatement itself is [unverified]: the arithmetic lives in the third-party crate `image` 0.25.10
(Cargo.lock:2224-2227; `imageops::sample::{resize, vertical_sample, horizontal_sample, bc_cubic_spline}`), which is not under
/root/reference and cannot be built here.  What follows restates the crate's published algorithm from its documentation and
source as known to the author: vertical pass first into an f32 image, then the horizontal pass; per output sample the taps
`left..right` around the source coordinate, kernel argument scaled by max(ratio, 1), weights normalised by their sum, f32
accumulation in tap order, final clamp to [0, 255] and round-half-away-from-zero.  tests/test_image_pre_cpu.py checks it against
Pillow's bicubic (the same a = -0.5 kernel with the same support scaling, different pass order and 8-bit intermediate): a sanity
anchor of +-2 grey levels, not a pin.
"""
from __future__ import annotations

import numpy as np

F = np.float32


def bc_cubic_spline(x: np.float32, b: float = 0.0, c: float = 0.5) -> np.float32:
    """CatmullRom = B-C spline with B = 0, C = 0.5, support 2 (f32 arithmetic, powers as repeated products)."""
    a = F(abs(F(x)))
    b, c = F(b), F(c)
    if a < F(1.0):
        k = (F(12.0) - F(9.0) * b - F(6.0) * c) * (a * a * a) + (F(-18.0) + F(12.0) * b + F(6.0) * c) * (a * a) + (F(6.0) - F(2.0) * b)
    elif a < F(2.0):
        k = (-b - F(6.0) * c) * (a * a * a) + (F(6.0) * b + F(30.0) * c) * (a * a) + (F(-12.0) * b - F(48.0) * c) * a + (F(8.0) * b + F(24.0) * c)
    else:
        k = F(0.0)
    return F(k / F(6.0))


def sample_taps(n_in: int, n_out: int, support: float = 2.0):
    """Per output index: (left, normalised f32 weights) -- the loop head shared by vertical_sample / horizontal_sample."""
    ratio = F(n_in) / F(n_out)
    sratio = F(1.0) if ratio < F(1.0) else ratio
    src_support = F(support) * sratio
    taps = []
    for o in range(n_out):
        inp = (F(o) + F(0.5)) * ratio
        left = int(np.floor(inp - src_support))
        left = min(max(left, 0), n_in - 1)
        right = int(np.ceil(inp + src_support))
        right = min(max(right, left + 1), n_in)
        inp = inp - F(0.5)
        ws = [bc_cubic_spline((F(i) - inp) / sratio) for i in range(left, right)]
        s = F(0.0)
        for w in ws:
            s = F(s + w)
        taps.append((left, np.asarray([F(w / s) for w in ws], dtype=F)))
    return taps


def resize_exact_catmullrom(img_u8_hwc: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    """imageops::resize(image, new_w, new_h, CatmullRom) on an RGB8 image (H, W, 3)."""
    img = np.asarray(img_u8_hwc, dtype=np.uint8)
    H, W, C = img.shape
    if (new_w, new_h) == (W, H):
        return img.copy()
    src = img.astype(F)
    tmp = np.zeros((new_h, W, C), dtype=F)                      # vertical_sample -> f32 image
    for oy, (left, ws) in enumerate(sample_taps(H, new_h)):
        t = np.zeros((W, C), dtype=F)
        for i, w in enumerate(ws):
            t = (t + (src[left + i] * w).astype(F)).astype(F)  # t += p * w, one rounding per product and per sum
        tmp[oy] = t
    out = np.zeros((new_h, new_w, C), dtype=np.uint8)           # horizontal_sample -> clamp, round, u8
    for ox, (left, ws) in enumerate(sample_taps(W, new_w)):
        t = np.zeros((new_h, C), dtype=F)
        for i, w in enumerate(ws):
            t = (t + (tmp[:, left + i] * w).astype(F)).astype(F)
        t = np.clip(t, F(0.0), F(255.0))
        fl = np.floor(t)                                        # f32::round = half away from zero; t >= 0 here, t - floor(t) is exact
        out[:, ox] = (fl + ((t - fl) >= F(0.5))).astype(np.uint8)
    return out
