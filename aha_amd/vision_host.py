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


# ---- video path (the part of get_video_data / process_info that is the reference's own arithmetic; decoding the container and
# the swscale BILINEAR|ACCURATE_RND resize are ffmpeg's and stay with the caller) ------------------------------------------------
def video_smart_resize(num_frames: int, height: int, width: int, temporal_factor: int = 2, factor: int = 32, min_pixels: int = 4096,
                       max_pixels: int = 25165824, video_ratio: int | None = 16) -> Tuple[int, int]:
    """video_smart_resize, /root/reference/src/utils/video_utils.rs:9-59, through the library's host code (csrc/image_pre.hip): the
    frame size get_video_data scales to (processor.rs:496-505: factor = patch_size * merge_size, min / max_pixels = the video
    preprocessor's shortest / longest edge, video_ratio = 16 so that swscale gets a multiple of 16)."""
    import ctypes as C
    from ._lib import AhaHipError, check, lib
    h, w = C.c_uint32(), C.c_uint32()
    try:
        check(lib().aha_hip_video_smart_resize(num_frames, height, width, temporal_factor, factor, min_pixels, max_pixels, video_ratio or 0,
                                               C.byref(h), C.byref(w)))
    except AhaHipError as e:   # the reference returns Err(anyhow!(..)) with these messages
        raise ValueError(str(e)) from e
    return int(h.value), int(w.value)


def sample_video_frames(total_frames: int, rate: float, fps: int = 2, min_frames: int = 4, max_frames: int = 768) -> Tuple[int, int, List[int]]:
    """Which decoded frames get_video_data keeps (processor.rs:481-489,518-535): (nframes, sample_interval, frame_indices).
    nframes only sizes the resize; the kept frames are the multiples of round(frames / nframes)."""
    import ctypes as C
    from ._lib import check, lib
    n, iv = C.c_uint32(), C.c_uint32()
    check(lib().aha_hip_video_sample_frames(total_frames, float(rate), fps, min_frames, max_frames, C.byref(n), C.byref(iv)))
    return int(n.value), int(iv.value), list(range(0, total_frames, int(iv.value)))


def calculate_timestamps(frame_indices: List[int], fps: float, t_merge_size: int = 2) -> List[float]:
    """calculate_timestamps (processor.rs:283-307): seconds of every temporal patch = mean of its first and last frame time (f32)."""
    import ctypes as C
    from ._lib import check, lib
    idx = np.ascontiguousarray(np.asarray(frame_indices, dtype=np.uint32).reshape(-1))
    out = np.empty((idx.size + t_merge_size - 1) // t_merge_size, dtype=np.float32)
    n = check(lib().aha_hip_video_timestamps(idx.ctypes.data_as(C.c_void_p), idx.size, float(fps), t_merge_size,
                                             out.ctypes.data_as(C.c_void_p), out.size))
    assert n == out.size
    return [float(x) for x in out]


def process_videos(videos_u8: List[torch.Tensor], cfg: Qwen3VLConfig) -> Tuple[torch.Tensor, np.ndarray]:
    """process_videos (processor.rs:253-281) on the GPU: per video (T, H, W, 3) uint8 frames (already sampled and resized) ->
    pixel_values_video rows and the (t, h, w) grids."""
    v = cfg.vision
    pv, grids = [], []
    for fr in videos_u8:
        T, H, W = (int(x) for x in fr.shape[:3])
        pv.append(ops.video_to_patches(fr, v.patch_size, v.spatial_merge_size))
        grids.append([(T + v.temporal_patch_size - 1) // v.temporal_patch_size, H // v.patch_size, W // v.patch_size])
    return torch.cat(pv, 0), np.asarray(grids, dtype=np.uint32)


def expand_vision_placeholders(text: str, image_grid_thw, video_grid_thw, video_meta, merge_size: int = 2,
                               image_token: str = "<|image_pad|>", video_token: str = "<|video_pad|>",
                               vision_start: str = "<|vision_start|>", vision_end: str = "<|vision_end|>") -> str:
    """process_info's rewrite of the rendered chat text (processor.rs:386-431): every <|image_pad|> becomes t*h*w/merge^2 of
    them; every <|vision_start|><|video_pad|><|vision_end|> (or bare <|video_pad|>) becomes, per temporal patch,
    "<{:.1} seconds>" + <|vision_start|> + h*w/merge^2 x <|video_pad|> + <|vision_end|>.  video_meta[k] = (frame_indices, fps) of
    video k (VideoMetadata, processor.rs:39-46)."""
    m2 = merge_size * merge_size
    if image_grid_thw is not None:
        k = 0
        while image_token in text:
            t, h, w = (int(x) for x in image_grid_thw[k])
            text = text.replace(image_token, "<|placeholder|>" * (t * h * w // m2), 1)
            k += 1
        text = text.replace("<|placeholder|>", image_token)
    if video_grid_thw is not None:
        k = 0
        while video_token in text:
            t, h, w = (int(x) for x in video_grid_thw[k])
            stamps = calculate_timestamps(video_meta[k][0], video_meta[k][1], merge_size)
            block = "".join(f"<{stamps[f]:.1f} seconds>{vision_start}{'<|placeholder|>' * (h * w // m2)}{vision_end}" for f in range(t))
            three = vision_start + video_token + vision_end
            text = text.replace(three, block, 1) if three in text else text.replace(video_token, block, 1)
            k += 1
        text = text.replace("<|placeholder|>", video_token)
    return text


def video_prompt_ids(cfg: Qwen3VLConfig, video_grids: np.ndarray, stamp_ids: List[List[int]] | None = None) -> List[int]:
    """The token layout expand_vision_placeholders produces for the videos, on ids: per temporal patch the (tokenised) timestamp,
    <|vision_start|>, h*w/merge^2 x <|video_pad|>, <|vision_end|>.  stamp_ids[j] = ids of the j-th timestamp text."""
    m2 = cfg.vision.spatial_merge_size ** 2
    ids, j = [], 0
    for g in np.asarray(video_grids).tolist():
        for _ in range(g[0]):
            ids += list(stamp_ids[j]) if stamp_ids is not None else []
            ids += [cfg.vision_start_token_id] + [cfg.video_token_id] * (g[1] * g[2] // m2) + [cfg.vision_end_token_id]
            j += 1
    return ids


# ---- process_info / get_data: the request-level orchestration ---------------------------------------------------------------------
def extract_vision_info(messages) -> dict:
    """extract_vision_info (processor.rs:126-149): the image / video URLs of the USER messages whose content is a list of parts, in
    message order.  Parts are an untagged serde enum (params/chat.rs:608-615): the first variant whose fields are present wins --
    Text (type + text), Image (type + image_url), Audio (type + audio_url), Video (type + video_url) -- the `type` string itself is
    not looked at."""
    out = {"image": [], "video": []}
    for m in messages:
        if m.get("role") != "user" or not isinstance(m.get("content"), list):
            continue
        for part in m["content"]:
            if not isinstance(part, dict) or "type" not in part:
                continue
            if "text" in part:
                continue
            if isinstance(part.get("image_url"), dict) and "url" in part["image_url"]:
                out["image"].append(part["image_url"]["url"])
            elif "audio_url" in part:
                continue
            elif isinstance(part.get("video_url"), dict) and "url" in part["video_url"]:
                out["video"].append(part["video_url"]["url"])
    return out


def plan_video(total_frames: int, rate: float, height: int, width: int, cfg: Qwen3VLConfig, fps: int = 2, min_frames: int = 4,
               max_frames: int = 768, min_pixels: int = 4096, max_pixels: int = 25165824) -> dict:
    """What get_video_data decides before it decodes (processor.rs:481-505): how many frames size the resize, which decoded frames
    are kept, and the size swscale scales them to."""
    nframes, interval, idx = sample_video_frames(total_frames, rate, fps, min_frames, max_frames)
    v = cfg.vision
    rh, rw = video_smart_resize(nframes, height, width, v.temporal_patch_size, v.patch_size * v.spatial_merge_size, min_pixels, max_pixels, 16)
    return {"nframes": nframes, "sample_interval": interval, "frame_indices": idx, "resize_hw": (rh, rw)}


class Qwen3VLProcessor:
    """Qwen3VLProcessor::process_info (processor.rs:310-444) for the Python host: collect the request's images and videos, turn them
    into patch rows + grids, rewrite the rendered chat text.  `image_loader(url) -> (H, W, 3) uint8` defaults to media_host.get_image;
    `video_loader(url, plan) -> (frames (T, H, W, 3) uint8 at plan(...)["resize_hw"], frame_indices, fps)` has no default: decoding
    the container and the swscale resize are ffmpeg's (plan = lambda total_frames, rate, height, width: plan_video(...)).
    `image_fn` / `video_fn` turn the decoded arrays into (rows, grids); the defaults run on the GPU (process_images /
    process_videos).  As in the reference a source that fails to load is reported and skipped (processor.rs:330-336,359-366)."""

    def __init__(self, cfg: Qwen3VLConfig, image_loader=None, video_loader=None, image_fn=None, video_fn=None, device: str = "cuda",
                 fps: int = 2, min_frames: int = 4, max_frames: int = 768, video_min_pixels: int = 4096, video_max_pixels: int = 25165824):
        from . import media_host
        self.cfg, self.device = cfg, device
        self.image_loader = image_loader or media_host.get_image
        self.video_loader = video_loader
        self.image_fn = image_fn or (lambda imgs: (lambda d: (d.pixel_values, d.image_grid_thw))(
            process_images([torch.from_numpy(np.ascontiguousarray(i)).to(device) for i in imgs], cfg)))
        self.video_fn = video_fn or (lambda vids: process_videos([torch.as_tensor(np.ascontiguousarray(v)).to(device) for v in vids], cfg))
        self.fps, self.min_frames, self.max_frames = fps, min_frames, max_frames
        self.video_min_pixels, self.video_max_pixels = video_min_pixels, video_max_pixels
        self.warnings: List[str] = []

    def plan(self, total_frames: int, rate: float, height: int, width: int) -> dict:
        return plan_video(total_frames, rate, height, width, self.cfg, self.fps, self.min_frames, self.max_frames,
                          self.video_min_pixels, self.video_max_pixels)

    def process_info(self, messages, text: str) -> dict:
        """-> GeneralInput as a dict: replace_text, pixel_values, image_grid_thw, pixel_values_video, video_grid_thw."""
        info = extract_vision_info(messages)
        out = {"replace_text": text, "pixel_values": None, "image_grid_thw": None, "pixel_values_video": None, "video_grid_thw": None}
        imgs = []
        for url in info["image"]:
            try:
                imgs.append(self.image_loader(url))
            except Exception as e:  # noqa: BLE001  (println!("get_image err: {e:?}") and carry on)
                self.warnings.append(f"get_image err: {e}")
        if imgs:
            out["pixel_values"], out["image_grid_thw"] = self.image_fn(imgs)
        vids, meta = [], []
        for url in info["video"]:
            try:
                if self.video_loader is None:
                    raise RuntimeError("no video loader: decoding is the caller's (ffmpeg)")
                frames, idx, fps = self.video_loader(url, self.plan)
                vids.append(frames)
                meta.append((list(idx), float(fps)))
            except Exception as e:  # noqa: BLE001
                self.warnings.append(f"get_video_data err: {e}")
        if vids:
            out["pixel_values_video"], out["video_grid_thw"] = self.video_fn(vids)
        out["replace_text"] = expand_vision_placeholders(text, out["image_grid_thw"], out["video_grid_thw"], meta,
                                                         self.cfg.vision.spatial_merge_size)
        return out


def get_data(messages, chat_template, tokenizer, processor: Qwen3VLProcessor, tools=None, enable_thinking=None):
    """Qwen3VLGenerateModel::get_data (qwen3vl/generate.rs:79-101): render the chat template, run the processor, tokenise the
    rewritten text -> (input_ids, MultiModalData) for forward_initial.  (cache_position is arange(len) and implicit here.)"""
    rendered = chat_template.apply_chat_template(messages, tools, enable_thinking)
    inp = processor.process_info(messages, rendered)
    ids = tokenizer.text_encode(inp["replace_text"])
    data = MultiModalData(inp["pixel_values"], inp["image_grid_thw"], pixel_values_video=inp["pixel_values_video"],
                          video_grid_thw=inp["video_grid_thw"])
    return ids, data


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


def get_rope_index(cfg: Qwen3VLConfig, input_ids, image_grid_thw, video_grid_thw=None):
    """Qwen3VLModel::get_rope_index through the library's host code (csrc/vision.hip rope_index_core): (3, S) int32
    positions and rope_delta.  No GPU involved."""
    import ctypes as C
    from ._lib import check, lib
    from .model import make_desc
    ids = np.ascontiguousarray(np.asarray(input_ids, dtype=np.uint32).reshape(-1))
    grid = np.ascontiguousarray(np.asarray(image_grid_thw if image_grid_thw is not None else [], dtype=np.uint32).reshape(-1, 3))
    vgrid = np.ascontiguousarray(np.asarray(video_grid_thw if video_grid_thw is not None else [], dtype=np.uint32).reshape(-1, 3))
    pos = np.empty((3, ids.size), dtype=np.int32)
    delta = C.c_int64()
    desc = make_desc(cfg)
    check(lib().aha_hip_get_rope_index_mm(C.byref(desc), ids.ctypes.data_as(C.c_void_p), ids.size,
                                          grid.ctypes.data_as(C.c_void_p) if grid.shape[0] else None, grid.shape[0],
                                          vgrid.ctypes.data_as(C.c_void_p) if vgrid.shape[0] else None, vgrid.shape[0],
                                          pos.ctypes.data_as(C.c_void_p), C.byref(delta)))
    return pos, int(delta.value)
