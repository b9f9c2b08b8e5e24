"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's Qwen3-VL path.

PARITY UNPINNED (see oracle/numerics.py).  Independent cross-check against HF transformers' Qwen3-VL in
tests/test_oracle_vs_hf.py.  Reference files restated:
  src/models/qwen3vl/model.rs:32-104 (patch embed), 106-185 (merger), 232-278 (vision attention), 346-370 (block),
      512-639 (fast_pos_embed_interpolate), 641-690 (rot_pos_emb), 692-740 (vision forward),
      775-828 (text model + DeepStack), 901-1133 (get_rope_index), 1135-1277 (forward)
  src/models/qwen3vl/processor.rs:174-251 (process_vision_tensor / process_images), 253-281 (process_videos),
      283-307 (calculate_timestamps), 386-431 (placeholder expansion), 481-535 (frame sampling of get_video_data; the ffmpeg
      decode + swscale resize around it is third-party and not restated)
  src/utils/video_utils.rs:9-59 (video_smart_resize)
  src/utils/img_utils.rs:272-293 (img_transform)
  src/position_embed/rope.rs:75-94 (apply_rotary_pos_emb_vision), 423-476, 541-580 (vision rotary, interleaved M-RoPE)
  src/utils/tensor_utils.rs:294-321 (masked_scatter_dim0), 354-365 (linspace), 466-470 (mask_index_add)
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .numerics import Numerics
from . import qwen3 as oq


# ---- V0: processor ------------------------------------------------------------------------------------------------
def img_transform(nm: Numerics, img_u8_hwc: np.ndarray, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)) -> torch.Tensor:
    """img_utils.rs:272-293: u8 HWC -> CHW f32, * (1/255), (x - mean) / std in f32, cast to T."""
    x = torch.from_numpy(np.ascontiguousarray(img_u8_hwc)).permute(2, 0, 1).to(torch.float32)
    x = x * torch.tensor(1.0 / 255.0, dtype=torch.float32)
    m = torch.tensor(mean, dtype=torch.float32).reshape(3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).reshape(3, 1, 1)
    return nm.r((x - m) / s)


def process_vision_tensor(img_tchw: torch.Tensor, patch=16, tps=2, merge=2) -> Tuple[torch.Tensor, np.ndarray]:
    """processor.rs:174-227: (t,c,h,w) -> (N, c*tps*patch*patch) rows in merge-window order, grid_thw (1,3)."""
    t = img_tchw.shape[0]
    if t % tps:
        img_tchw = torch.cat([img_tchw, img_tchw[-1:].repeat(tps - t % tps, 1, 1, 1)], 0)
    c = img_tchw.shape[1]
    gt, gh, gw = img_tchw.shape[0] // tps, img_tchw.shape[2] // patch, img_tchw.shape[3] // patch
    x = img_tchw.reshape(gt, tps, c, gh // merge, merge, patch, gw // merge, merge, patch)
    x = x.permute(0, 3, 6, 4, 7, 2, 1, 5, 8).reshape(gt * gh * gw, c * tps * patch * patch).contiguous()
    return x, np.array([[gt, gh, gw]], dtype=np.uint32)


def process_images(nm: Numerics, imgs_u8: List[np.ndarray], patch=16, tps=2, merge=2):
    """processor.rs:229-251: each image (already at its smart-resize size) is duplicated to 2 frames."""
    pv, grids = [], []
    for im in imgs_u8:
        x = img_transform(nm, im)[None]
        x = torch.cat([x, x], 0)
        p, g = process_vision_tensor(x, patch, tps, merge)
        pv.append(p)
        grids.append(g)
    return torch.cat(pv, 0), np.concatenate(grids, 0)


def process_videos(nm: Numerics, videos_u8: List[np.ndarray], patch=16, tps=2, merge=2, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
    """processor.rs:253-281: each video (T, H, W, 3) u8 (get_video_data's frames, already at the resize size) -> model dtype,
    affine(1/255, 0), broadcast_sub(mean), broadcast_div(std) -- every op in T, scalars / constants cast to T first (the image
    path normalises in f32 instead) -- then process_vision_tensor with the real frame pairs."""
    pv, grids = [], []
    for v in videos_u8:
        x = torch.from_numpy(np.ascontiguousarray(v)).permute(0, 3, 1, 2).to(torch.float32)   # (t, c, h, w); u8 is exact in T
        x = nm.r(nm.r(x * nm.r(torch.tensor(1.0 / 255.0))) + 0.0)
        m = nm.r(torch.tensor(mean, dtype=torch.float32)).reshape(1, 3, 1, 1)
        sd = nm.r(torch.tensor(std, dtype=torch.float32)).reshape(1, 3, 1, 1)
        x = nm.r(nm.r(x - m) / sd)
        p, g = process_vision_tensor(x, patch, tps, merge)
        pv.append(p)
        grids.append(g)
    return torch.cat(pv, 0), np.concatenate(grids, 0)


def _round_half_away(q) -> int:
    """f32::round."""
    q = np.float32(q)
    fl = np.floor(q)
    return int(fl + (np.float32(q - fl) >= np.float32(0.5)))


def video_smart_resize(num_frames, height, width, temporal_factor=2, factor=32, min_pixels=0, max_pixels=0, video_ratio=16):
    """utils/video_utils.rs:9-59 (round/floor/ceil_by_factor: utils/mod.rs:392-405, f32 quotients; u32 products)."""
    if num_frames < temporal_factor:
        raise ValueError(f"{num_frames} must be larger than temporal_factor {temporal_factor}")
    if height < factor or width < factor:
        raise ValueError(f"height:{height} or width:{width} must be larger than factor:{factor}")
    if max(height, width) // min(height, width) > 200:
        raise ValueError("absolute aspect ratio mush be smaller than 200")
    f = factor if video_ratio is None else factor * video_ratio // math.gcd(factor, video_ratio)
    rnd = lambda v, k: _round_half_away(np.float32(v) / np.float32(k)) * k
    h_bar, w_bar, t_bar = rnd(height, f), rnd(width, f), rnd(num_frames, temporal_factor)
    if t_bar * h_bar * w_bar > max_pixels:
        beta = np.sqrt(np.float32(num_frames * height * width) / np.float32(max_pixels), dtype=np.float32)
        h_bar = max(f, int(np.floor(np.float32(height) / beta / np.float32(f))) * f)
        w_bar = max(f, int(np.floor(np.float32(width) / beta / np.float32(f))) * f)
    elif t_bar * h_bar * w_bar < min_pixels:
        beta = np.sqrt(np.float32(min_pixels) / np.float32(num_frames * height * width), dtype=np.float32)
        h_bar = int(np.ceil(np.float32(height) * beta / np.float32(f))) * f
        w_bar = int(np.ceil(np.float32(width) * beta / np.float32(f))) * f
    return h_bar, w_bar


def sample_frame_indices(total_frames: int, rate: float, fps=2, min_frames=4, max_frames=768):
    """processor.rs:481-489,518-535: nframes = round(frames / rate * fps) clamped to [min_frames, max_frames] and to the frame
    count; sample_interval = round(frames / nframes); every decoded frame whose id is a multiple of the interval is kept."""
    frames32, rate32 = np.float32(total_frames), np.float32(rate)
    nframes = _round_half_away(frames32 / rate32 * np.float32(fps))
    nframes = min(min(max(nframes, min_frames), max_frames), total_frames)
    interval = _round_half_away(frames32 / np.float32(nframes))
    return nframes, interval, [i for i in range(total_frames) if i % interval == 0]


def calculate_timestamps(frame_indices: List[int], fps: float, t_merge_size=2) -> List[float]:
    """processor.rs:283-307 (f32): pad to a multiple of t_merge_size with the last index, index / fps, mean of the first and last
    of every group."""
    idx = list(frame_indices)
    if len(idx) % t_merge_size:
        idx += [idx[-1]] * (t_merge_size - len(idx) % t_merge_size)
    ts = [np.float32(i) / np.float32(fps) for i in idx]
    return [float((ts[i] + ts[i + t_merge_size - 1]) / np.float32(2.0)) for i in range(0, len(ts), t_merge_size)]


def expand_placeholders(text: str, image_grids, video_grids, video_meta, merge=2, image_token="<|image_pad|>", video_token="<|video_pad|>",
                        vs="<|vision_start|>", ve="<|vision_end|>") -> str:
    """process_info's text rewrite (processor.rs:386-431).  video_meta: per video (frame_indices, fps)."""
    m2 = merge * merge
    if image_grids is not None:
        k = 0
        while image_token in text:
            n = int(np.prod(np.asarray(image_grids[k], dtype=np.int64))) // m2
            text = text.replace(image_token, "<|placeholder|>" * n, 1)
            k += 1
        text = text.replace("<|placeholder|>", image_token)
    if video_grids is not None:
        k = 0
        while video_token in text:
            t, h, w = (int(x) for x in video_grids[k])
            stamps = calculate_timestamps(video_meta[k][0], video_meta[k][1], merge)
            ph = ""
            for f in range(t):
                ph += "<%.1f seconds>" % stamps[f] + vs + "<|placeholder|>" * (h * w // m2) + ve
            three = vs + video_token + ve
            text = text.replace(three, ph, 1) if three in text else text.replace(video_token, ph, 1)
            k += 1
        text = text.replace("<|placeholder|>", video_token)
    return text


# ---- M2: get_rope_index -------------------------------------------------------------------------------------------
def get_rope_index(ids: List[int], grid_thw: Optional[np.ndarray], cfg, video_grid_thw: Optional[np.ndarray] = None) -> Tuple[np.ndarray, int]:
    """qwen3vl/model.rs:901-1133 for B=1, no attention mask.  Returns position_ids (3,S) int64 and rope_delta."""
    S = len(ids)
    has_img = grid_thw is not None and len(grid_thw) > 0
    has_vid = video_grid_thw is not None and len(video_grid_thw) > 0
    if not has_img and not has_vid:
        return np.tile(np.arange(S, dtype=np.int64), (3, 1)), 0
    # model.rs:908-925: every (t, h, w) video grid is replaced by t rows (1, h, w)
    vrows = [[1, int(g[1]), int(g[2])] for g in (video_grid_thw if has_vid else []) for _ in range(int(g[0]))]
    merge = cfg.vision.spatial_merge_size
    blocks: List[np.ndarray] = []
    text_start, image_index, video_index = 0, 0, 0
    ids_a = np.asarray(ids)

    def next_start():   # llm_pos_ids_list.last().max_all() + 1, or 0 for an empty list (model.rs:984-991)
        nonempty = [b for b in blocks if b.size]
        return int(nonempty[-1].max()) + 1 if nonempty else 0

    nxt = [j + 1 for j in range(S - 1) if ids_a[j] == cfg.vision_start_token_id]   # get_vision_next_indices
    for e in nxt:
        if ids_a[e] == cfg.image_token_id:
            thw = grid_thw[image_index]
            image_index += 1
        elif ids_a[e] == cfg.video_token_id:
            thw = vrows[video_index]
            video_index += 1
        else:
            continue  # (the reference reuses the previous grid here -- or panics on an empty one; never in a processor-made prompt)
        text_end = e
        t, gh, gw = int(thw[0]), int(thw[1]) // merge, int(thw[2]) // merge
        text_len = text_end - text_start
        start = next_start()
        blocks.append(np.tile(np.arange(start, start + text_len, dtype=np.int64), (3, 1)))
        base = start + text_len
        ti = np.repeat(np.arange(base, base + t), gh * gw)
        hi = np.tile(np.repeat(np.arange(base, base + gh), gw), t)
        wi = np.tile(np.arange(base, base + gw), t * gh)
        blocks.append(np.stack([ti, hi, wi]).astype(np.int64))
        text_start = text_end + t * gh * gw
    if text_start < S:
        start = next_start()
        blocks.append(np.tile(np.arange(start, start + S - text_start, dtype=np.int64), (3, 1)))
    pos = np.concatenate(blocks, axis=1)
    assert pos.shape[1] == S, (pos.shape, S)
    return pos, int(pos.max()) + 1 - S


def mrope_cos_sin(inv_freq: torch.Tensor, pos: np.ndarray, mrope_section) -> Tuple[torch.Tensor, torch.Tensor]:
    """Qwen3VLTextRotaryEmbedding::forward + apply_interleaved_mrope (rope.rs:454-476, 541-580): (1,S,128) f32."""
    p = torch.from_numpy(np.asarray(pos)).to(torch.float32)            # (3,S)
    freqs = p[:, :, None] * inv_freq[None, None, :]                    # (3,S,d/2), K=1 matmul == exact product
    g = freqs[0].clone()
    for dim in (1, 2):
        idx = torch.arange(dim, mrope_section[dim] * 3, 3)
        g[:, idx] = freqs[dim][:, idx]
    emb = torch.cat([g, g], -1)
    return emb.cos()[None], emb.sin()[None]


# ---- V1-V7: vision tower --------------------------------------------------------------------------------------------
def layer_norm(nm: Numerics, x, w, b, eps=1e-6):
    """candle_nn::LayerNorm (modules.rs:867-875): f32 internally, output in T."""
    x32 = x.float()
    mu = x32.mean(-1, keepdim=True)
    var = ((x32 - mu) ** 2).mean(-1, keepdim=True)
    return nm.r((x32 - mu) / torch.sqrt(var + eps) * w + b)


def linspace_f32(start: float, end: float, steps: int) -> np.ndarray:
    """tensor_utils.rs:354-365 in f32."""
    if steps == 1:
        return np.array([start], dtype=np.float32)
    step = np.float32((np.float32(end) - np.float32(start)) / np.float32(steps - 1))
    return (np.float32(start) + np.arange(steps, dtype=np.float32) * step).astype(np.float32)


def gelu_tanh(x):
    return torch.nn.functional.gelu(x, approximate="tanh")


def gelu_erf(x):
    return torch.nn.functional.gelu(x)


class OracleVision:
    """Qwen3VLVisionModel (qwen3vl/model.rs:372-741)."""

    def __init__(self, cfg, weights: Dict[str, torch.Tensor], nm: Numerics, prefix="model.visual.", consume: bool = False):
        self.cfg, self.v, self.nm, self.p = cfg, cfg.vision, nm, prefix
        self.w = {}
        for k in [k for k in weights if k.startswith(prefix)]:
            self.w[k] = nm.r((weights.pop(k) if consume else weights[k]).float())
        v = self.v
        self.G = int(math.sqrt(v.num_position_embeddings))
        self.inv_freq = oq.compute_default_rope_parameters(v.head_dim // 2, 10000.0)   # rope.rs:429-433
        self.scale = oq.attn_scale(nm, v.head_dim)

    def patch_embed(self, pv):
        """model.rs:96-103: matmul with the flattened+transposed conv3d weight, broadcast_add bias."""
        W = self.w[self.p + "patch_embed.proj.weight"].reshape(self.v.hidden_size, -1)
        return self.nm.linear(pv, W, self.w[self.p + "patch_embed.proj.bias"])

    def pos_embed_indices(self, grid_thw):
        """Host index/weight construction of fast_pos_embed_interpolate (model.rs:512-603), in merge-window order.
        Returns idx (4,N) int64 and weights (4,N) f32."""
        G, merge = self.G, self.v.spatial_merge_size
        idx_all, w_all = [], []
        for (t, h, w) in np.asarray(grid_thw).tolist():
            hi, wi = linspace_f32(0.0, G - 1, h), linspace_f32(0.0, G - 1, w)
            hf, wf = hi.astype(np.uint32), wi.astype(np.uint32)          # to_dtype(U32) truncates
            hc, wc = np.minimum(hf + 1, G - 1), np.minimum(wf + 1, G - 1)
            dh, dw = (hi - hf.astype(np.float32))[:, None], (wi - wf.astype(np.float32))[None, :]
            bh, bhc = (hf * G)[:, None], (hc * G)[:, None]
            idx = np.stack([bh + wf[None], bh + wc[None], bhc + wf[None], bhc + wc[None]]).reshape(4, -1)
            one = np.float32(1.0)
            wt = np.stack([(one - dh) * (one - dw), (one - dh) * dw, dh * (one - dw), dh * dw]).astype(np.float32).reshape(4, -1)
            # repeat t times, then (t, h/m, m, w/m, m) -> (t, h/m, w/m, m, m)  (model.rs:604-627)
            def perm(a):
                a = np.tile(a.reshape(4, 1, h, w), (1, t, 1, 1)).reshape(4, t, h // merge, merge, w // merge, merge)
                return a.transpose(0, 1, 2, 4, 3, 5).reshape(4, -1)
            idx_all.append(perm(idx))
            w_all.append(perm(wt))
        return np.concatenate(idx_all, 1).astype(np.int64), np.concatenate(w_all, 1)

    def pos_embed(self, grid_thw):
        nm = self.nm
        idx, wt = self.pos_embed_indices(grid_thw)
        E = self.w[self.p + "pos_embed.weight"]
        wt_t = nm.r(torch.from_numpy(wt))                               # weight_tensor.to_dtype(self.dtype)
        pe = nm.r(E[torch.from_numpy(idx)] * wt_t[..., None])           # (4,N,D) broadcast_mul
        return nm.r(nm.r(nm.r(pe[0] + pe[1]) + pe[2]) + pe[3])

    def rot_pos(self, grid_thw):
        """rot_pos_emb (model.rs:641-690): per patch (row, col) in merge-window order -> (N, head_dim/2) freqs f32."""
        merge = self.v.spatial_merge_size
        rows, cols = [], []
        for (t, h, w) in np.asarray(grid_thw).tolist():
            mh, mw = h // merge, w // merge
            r = (np.arange(mh)[:, None, None, None] * merge + np.arange(merge)[None, None, :, None])
            c = (np.arange(mw)[None, :, None, None] * merge + np.arange(merge)[None, None, None, :])
            r = np.broadcast_to(r, (mh, mw, merge, merge)).reshape(-1)
            c = np.broadcast_to(c, (mh, mw, merge, merge)).reshape(-1)
            rows.append(np.tile(r, t))
            cols.append(np.tile(c, t))
        rows, cols = np.concatenate(rows), np.concatenate(cols)
        fr = torch.from_numpy(rows).float()[:, None] * self.inv_freq[None]
        fc = torch.from_numpy(cols).float()[:, None] * self.inv_freq[None]
        return torch.cat([fr, fc], 1), rows, cols

    def attention(self, li, x, cos, sin, seg):
        """Qwen3VLVisionAttention::forward (model.rs:232-278)."""
        nm, v, p = self.nm, self.v, f"{self.p}blocks.{li}.attn."
        N = x.shape[0]
        qkv = nm.linear(x, self.w[p + "qkv.weight"], self.w[p + "qkv.bias"]).reshape(N, 3, v.num_heads, v.head_dim)
        q, k, val = qkv[:, 0], qkv[:, 1], qkv[:, 2]
        c, s = nm.r(cos)[:, None], nm.r(sin)[:, None]               # apply_rotary_pos_emb_vision, rope.rs:75-94
        rot = lambda u: nm.r(nm.r(u * c) + nm.r(oq.rotate_half(u) * s))
        q, k = rot(q), rot(k)
        outs = []
        for a, b in zip(seg[:-1], seg[1:]):
            qq = q[a:b].transpose(0, 1)[None]
            kk = k[a:b].transpose(0, 1)[None]
            vv = val[a:b].transpose(0, 1)[None]
            outs.append(oq.eager_attention_forward(nm, qq, kk, vv, 1, None, self.scale))   # (1, n, heads, d)
        o = torch.cat(outs, 1).reshape(N, -1)
        return nm.linear(o, self.w[p + "proj.weight"], self.w[p + "proj.bias"])

    def block(self, li, x, cos, sin, seg):
        nm, p = self.nm, f"{self.p}blocks.{li}."
        h = layer_norm(nm, x, self.w[p + "norm1.weight"], self.w[p + "norm1.bias"])
        x = nm.r(x + self.attention(li, h, cos, sin, seg))
        h = layer_norm(nm, x, self.w[p + "norm2.weight"], self.w[p + "norm2.bias"])
        h = nm.r(gelu_tanh(nm.linear(h, self.w[p + "mlp.linear_fc1.weight"], self.w[p + "mlp.linear_fc1.bias"])))
        h = nm.linear(h, self.w[p + "mlp.linear_fc2.weight"], self.w[p + "mlp.linear_fc2.bias"])
        return nm.r(x + h)

    def merger(self, p, x, postshuffle):
        """Qwen3VLVisionPatchMerger::forward (model.rs:167-184); act = erf GELU (model.rs:131)."""
        nm = self.nm
        M = self.v.hidden_size * self.v.spatial_merge_size ** 2
        if postshuffle:
            x = x.reshape(-1, M)
        x = layer_norm(nm, x, self.w[p + "norm.weight"], self.w[p + "norm.bias"]).reshape(-1, M)
        h = nm.r(gelu_erf(nm.linear(x, self.w[p + "linear_fc1.weight"], self.w[p + "linear_fc1.bias"])))
        return nm.linear(h, self.w[p + "linear_fc2.weight"], self.w[p + "linear_fc2.bias"])

    def forward(self, pixel_values, grid_thw):
        """Qwen3VLVisionModel::forward (model.rs:692-740) -> (merged (N/4, out), [deepstack k (N/4, out)])."""
        nm = self.nm
        x = self.patch_embed(nm.r(pixel_values.float()))
        x = nm.r(x + self.pos_embed(grid_thw))
        rp, _, _ = self.rot_pos(grid_thw)
        emb = torch.cat([rp, rp], -1)
        cos, sin = emb.cos(), emb.sin()
        seg = [0]
        for (t, h, w) in np.asarray(grid_thw).tolist():
            for _ in range(t):
                seg.append(seg[-1] + h * w)
        deep = []
        for li in range(self.v.depth):
            x = self.block(li, x, cos, sin, seg)
            if li in self.v.deepstack_visual_indexes:
                k = self.v.deepstack_visual_indexes.index(li)
                deep.append(self.merger(f"{self.p}deepstack_merger_list.{k}.", x, True))
        return self.merger(self.p + "merger.", x, False), deep


class OracleQwen3VL:
    """Qwen3VLModel (qwen3vl/model.rs:837-1316) with its InferenceModel impl (1279-1316)."""

    def __init__(self, cfg, weights, nm: Optional[Numerics] = None, consume: bool = False):
        self.cfg, self.nm = cfg, nm or Numerics()
        tcfg = cfg.text
        tcfg.tie_word_embeddings = cfg.tie_word_embeddings
        has_vision = any(k.startswith("model.visual.") for k in weights)
        self.text = oq.OracleQwen3(tcfg, weights, self.nm, prefix="model.language_model.", consume=consume)
        self.vision = OracleVision(cfg, weights, self.nm, consume=consume) if has_vision else None
        self.rope_delta: Optional[int] = None
        self.last_image_embeds = None
        self.last_deepstack = None

    def clear_cache(self):
        self.rope_delta = None
        self.text.clear_cache()

    def stop_token_ids(self):
        return self.text.stop_token_ids()

    def forward(self, input_ids, seqlen_offset, mm=None):
        nm, t = self.nm, self.text
        ids = list(input_ids)
        S = len(ids)
        x = t.embed_tokens(ids)
        vis_rows, deep = None, None
        grid = None
        vgrid = None
        if mm is not None:
            pv, grid, pvv, vgrid = (tuple(mm) + (None, None))[:4]
            x = x.clone()
            img_rows, vid_rows, deep_i, deep_v = [], [], None, None
            if pv is not None:                                    # model.rs:1150-1168
                img, deep_i = self.vision.forward(pv, grid)
                img_rows = [i for i, tok in enumerate(ids) if tok == self.cfg.image_token_id]
                if len(img_rows) != img.shape[0]:
                    raise ValueError(f"n_image_token num: {len(img_rows)} not equal to image_embed len: {img.shape[0]}")
                x[0, img_rows] = img                              # masked_scatter_dim0
                self.last_image_embeds, self.last_deepstack = img, deep_i
            if pvv is not None:                                   # model.rs:1169-1187: a second pass of the same tower
                vid, deep_v = self.vision.forward(pvv, vgrid)
                vid_rows = [i for i, tok in enumerate(ids) if tok == self.cfg.video_token_id]
                if len(vid_rows) != vid.shape[0]:
                    raise ValueError(f"n_image_token num: {len(vid_rows)} not equal to image_embed len: {vid.shape[0]}")
                x[0, vid_rows] = vid
                self.last_video_embeds, self.last_video_deepstack = vid, deep_v
            # model.rs:1188-1225: deepstack rows of images and videos joined in position order (index_add into zeros)
            vis_rows = sorted(img_rows + vid_rows)
            if deep_i is not None and deep_v is not None:
                deep = []
                for di, dv in zip(deep_i, deep_v):
                    j = torch.zeros(len(vis_rows), di.shape[-1])
                    j[[vis_rows.index(r) for r in img_rows]] = di
                    j[[vis_rows.index(r) for r in vid_rows]] = dv
                    deep.append(j)
            else:
                deep = deep_i if deep_i is not None else deep_v
        if self.rope_delta is None:                               # model.rs:1229-1236
            pos, self.rope_delta = get_rope_index(ids, grid, self.cfg, vgrid)
        else:
            pos = np.tile(np.arange(S, dtype=np.int64) + seqlen_offset + self.rope_delta, (3, 1))
        cs = mrope_cos_sin(t.inv_freq, pos, self.cfg.text.mrope_section)

        def after(li, h):
            if deep is not None and li < len(deep):               # mask_index_add (model.rs:812-822)
                h = h.clone()
                h[0, vis_rows] = nm.r(h[0, vis_rows] + deep[li])
            return h

        h = t.forward_hidden(None, x, seqlen_offset, cos_sin=cs, after_layer=after)
        return nm.linear(h, t.lm_head)

    def forward_initial(self, input_ids, seqlen_offset, mm=None):
        return self.forward(input_ids, seqlen_offset, mm)

    def forward_step(self, input_ids, seqlen_offset):
        return self.forward(input_ids, seqlen_offset, None)
