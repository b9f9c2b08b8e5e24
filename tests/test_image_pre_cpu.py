"""Image resize in front of the Qwen3-VL patchifier (reference qwen3vl/processor.rs:150-171) on the CPU: the oracle restatement of
image 0.25.10's CatmullRom resize against Pillow's bicubic as a sanity anchor (same kernel family, different pass order and an
8-bit intermediate: not a pin), tap-table properties, and the C-ABI smart-resize arithmetic against the Python mirror."""
import ctypes as C

import numpy as np
import pytest

from aha_amd import _lib
from aha_amd.vision_host import img_smart_resize
from oracle import image_pre as ip


def test_catmullrom_kernel_values():
    assert ip.bc_cubic_spline(0.0) == np.float32(1.0)
    assert ip.bc_cubic_spline(1.0) == 0 and ip.bc_cubic_spline(2.0) == 0 and ip.bc_cubic_spline(2.5) == 0
    assert abs(ip.bc_cubic_spline(0.5) - 0.5625) < 1e-7 and abs(ip.bc_cubic_spline(1.5) + 0.0625) < 1e-7   # Catmull-Rom
    assert ip.bc_cubic_spline(-0.5) == ip.bc_cubic_spline(0.5)


@pytest.mark.parametrize("n_in,n_out", [(100, 32), (32, 100), (640, 480), (37, 64), (64, 64), (5, 1), (1, 7)])
def test_tap_tables(n_in, n_out):
    taps = ip.sample_taps(n_in, n_out)
    assert len(taps) == n_out
    sr = max(n_in / n_out, 1.0)
    for o, (left, ws) in enumerate(taps):
        assert 0 <= left and left + len(ws) <= n_in and len(ws) >= 1
        assert len(ws) <= int(np.ceil(4 * sr)) + 2
        assert abs(float(ws.sum(dtype=np.float64)) - 1.0) < 1e-5          # normalised
        centre = (o + 0.5) * n_in / n_out - 0.5
        assert left <= max(centre, 0) + 1e-3 and left + len(ws) - 1 >= min(centre, n_in - 1) - 1e-3


def test_constant_and_identity():
    img = np.full((40, 56, 3), 137, np.uint8)
    np.testing.assert_array_equal(ip.resize_exact_catmullrom(img, 64, 96), np.full((64, 96, 3), 137, np.uint8))
    np.testing.assert_array_equal(ip.resize_exact_catmullrom(img, 16, 24), np.full((16, 24, 3), 137, np.uint8))
    g = np.random.default_rng(0).integers(0, 256, (33, 47, 3), dtype=np.uint8)
    np.testing.assert_array_equal(ip.resize_exact_catmullrom(g, 33, 47), g)           # unchanged size: a copy


@pytest.mark.parametrize("shape,new", [((96, 128), (64, 96)), ((50, 70), (96, 128)), ((120, 90), (64, 64)), ((64, 64), (160, 96))])
def test_close_to_pillow_bicubic(shape, new):
    PIL = pytest.importorskip("PIL.Image")
    g = np.random.default_rng(sum(shape))
    # blocky content plus an edge, kept inside [40, 220]: Pillow clips its 8-bit intermediate to [0, 255] between the passes,
    # the crate keeps an unclamped f32 intermediate, so overshoot beyond the range is a genuine difference of the two algorithms
    base = np.clip(g.normal(120, 25, (shape[0] // 4 + 2, shape[1] // 4 + 2, 3)), 40, 180)
    img = np.kron(base, np.ones((4, 4, 1)))[: shape[0], : shape[1]]
    img[:, shape[1] // 2:] += 40
    img = np.clip(img, 0, 255).astype(np.uint8)
    got = ip.resize_exact_catmullrom(img, new[0], new[1]).astype(np.int32)
    ref = np.asarray(PIL.fromarray(img).resize((new[1], new[0]), PIL.BICUBIC)).astype(np.int32)
    d = np.abs(got - ref)
    assert d.max() <= 2 and d.mean() < 0.35, (d.max(), d.mean())


@pytest.mark.parametrize("h,w", [(1024, 1024), (100, 100), (37, 4000), (4000, 6000), (31, 33), (768, 1365), (1, 150)])
def test_smart_resize_c_abi_matches_python(h, w):
    l = _lib.lib()
    ho, wo = C.c_uint32(), C.c_uint32()
    for (mn, mx) in ((65536, 16777216), (3136, 1003520)):
        rc = l.aha_hip_img_smart_resize(h, w, 32, mn, mx, C.byref(ho), C.byref(wo))
        assert rc == 0
        assert (ho.value, wo.value) == img_smart_resize(h, w, 32, mn, mx)
        assert ho.value % 32 == 0 and wo.value % 32 == 0
    assert l.aha_hip_img_smart_resize(1, 500, 32, 65536, 16777216, C.byref(ho), C.byref(wo)) < 0      # aspect ratio > 200
    with pytest.raises(ValueError):
        img_smart_resize(1, 500)


@pytest.mark.parametrize("n_in,n_out", [(64, 160), (64, 96), (100, 32), (75, 96), (1024, 672), (37, 64), (5, 1), (1, 7), (3000, 2048)])
def test_library_tap_tables_are_the_restatements_bit_for_bit(n_in, n_out):
    """The product's host-side tap builder (csrc/image_pre.hip build_taps) against oracle.sample_taps: same left / count and the
    same f32 weight bits, so the GPU passes (plain f32 multiply + add in tap order) reproduce the restatement exactly."""
    l = _lib.lib()
    left = (C.c_int32 * n_out)()
    count = (C.c_int32 * n_out)()
    cap = n_out * (int(np.ceil(4 * max(n_in / n_out, 1.0))) + 4)
    w = (C.c_float * cap)()
    n = l.aha_hip_debug_resize_taps(n_in, n_out, left, count, w, cap)
    assert n > 0
    ws = np.frombuffer(w, dtype=np.float32)[:n]
    off = 0
    for o, (rl, rw) in enumerate(ip.sample_taps(n_in, n_out)):
        assert (left[o], count[o]) == (rl, len(rw))
        np.testing.assert_array_equal(ws[off: off + len(rw)].view(np.uint32), rw.view(np.uint32))
        off += len(rw)
    assert off == n
