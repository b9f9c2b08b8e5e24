"""-m gpu: aha_hip_image_resize (CatmullRom resize_exact on the device) against the oracle restatement (oracle/image_pre.py).
Integer output through a fixed sequence of f32 roundings (un-fused multiply and add in tap order, taps from the same f32
arithmetic): bit-exact."""
import numpy as np
import pytest
import torch

from aha_amd import ops
from aha_amd.configs import tiny_qwen3vl
from aha_amd.vision_host import img_smart_resize, process_images
from oracle import image_pre as ip

pytestmark = pytest.mark.gpu


def rand_img(h, w, seed):
    g = np.random.default_rng(seed)
    img = g.integers(0, 256, (h, w, 3), dtype=np.uint8)
    img[: h // 3] = np.clip(g.normal(128, 30, (h // 3, w, 3)), 0, 255).astype(np.uint8)   # a smoother band
    img[h // 2, :, :] = 255                                                                # and hard lines: overshoot + clamp
    img[:, w // 2, :] = 0
    return img


@pytest.mark.parametrize("shape,new", [((96, 128), (64, 96)), ((50, 70), (96, 128)), ((120, 90), (64, 64)), ((64, 64), (160, 96)),
                                       ((1, 1), (32, 32)), ((33, 65), (32, 64)), ((200, 300), (32, 32)), ((31, 500), (64, 1024))])
def test_resize_bit_exact(gpu, shape, new):
    img = rand_img(shape[0], shape[1], sum(shape) + sum(new))
    got = ops.image_resize(torch.from_numpy(img).to(gpu), new[0], new[1]).cpu().numpy()
    ref = ip.resize_exact_catmullrom(img, new[0], new[1])
    assert got.shape == ref.shape
    np.testing.assert_array_equal(got, ref)


def test_same_size_is_a_copy(gpu):
    img = rand_img(64, 96, 1)
    np.testing.assert_array_equal(ops.image_resize(torch.from_numpy(img).to(gpu), 64, 96).cpu().numpy(), img)


def test_process_images_resizes_then_patchifies(gpu):
    """process_img end to end (processor.rs:150-171): a 100 x 75 image goes to its smart-resize size, then through the V0 kernel;
    the result equals patchifying the oracle-resized image."""
    cfg = tiny_qwen3vl()
    img = rand_img(100, 75, 5)
    th, tw = img_smart_resize(100, 75, cfg.vision.patch_size * cfg.vision.spatial_merge_size)
    assert (th, tw) != (100, 75)
    data = process_images([torch.from_numpy(img).to(gpu)], cfg)
    ref_img = ip.resize_exact_catmullrom(img, th, tw)
    want = process_images([torch.from_numpy(ref_img).to(gpu)], cfg)
    assert data.image_grid_thw.tolist() == [[1, th // cfg.vision.patch_size, tw // cfg.vision.patch_size]]
    assert torch.equal(data.pixel_values, want.pixel_values)
