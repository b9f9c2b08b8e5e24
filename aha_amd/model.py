"""Host-side mirror of the reference's operator interface for this path, on top of the C ABI.

  HipInferenceModel.forward_initial / forward_step / clear_cache / stop_token_ids
      == trait InferenceModel                (/root/reference/src/models/common/mod.rs:25-45)
  generate_generic(...)                      == generate_generic with Sampling::ArgMax when temperature < 1e-7
                                             (/root/reference/src/models/common/generate.rs:70-159, sample.rs:7-37)
Same names, same argument meaning, errors raised as exceptions where the reference returns Err(anyhow!).
Everything numeric happens inside libaha_hip.so; this file only marshals buffers.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import MmInput, ModelDesc, TensorView, check, lib
from .configs import Qwen3ASRConfig, Qwen3Config, Qwen3VLConfig

_DT = {torch.bfloat16: _lib.AHA_BF16, torch.float16: _lib.AHA_F16, torch.float32: _lib.AHA_F32}


class HipContext:
    def __init__(self, device: int = 0):
        self.handle = C.c_void_p()
        check(lib().aha_hip_init(device, C.byref(self.handle)))
        self.device = device

    def close(self):
        if self.handle:
            lib().aha_hip_shutdown(self.handle)
            self.handle = C.c_void_p()


@dataclass
class MultiModalData:
    """data_vec = [pixel_values, image_grid_thw, pixel_values_video, video_grid_thw, cache_position]
    (qwen3vl/generate.rs:79-101, model.rs:1292-1308)."""
    pixel_values: Optional[torch.Tensor] = None    # (n_patches, C*T*P*P) bf16 or f32, processor row order
    image_grid_thw: Optional[np.ndarray] = None    # (n_images, 3) uint32
    pixel_values_video: Optional[torch.Tensor] = None   # the same for the sampled frames of the videos (process_videos)
    video_grid_thw: Optional[np.ndarray] = None    # (n_videos, 3) uint32, t = temporal patches (frame pairs)
    # Qwen3-ASR: data_vec = [input_features] (qwen3_asr/generate.rs:100-125): log-mel (128, F) f32, or raw 16 kHz samples
    audio_features: Optional[np.ndarray] = None
    audio_samples: Optional[np.ndarray] = None
    # image-parallel ViT: embeddings already computed (vision_encode, gathered over RCCL): (1+K, n_tokens, H) bf16 on the GPU
    image_embeds: Optional[torch.Tensor] = None


def make_desc(cfg, kv_reserve_tokens: int = 0) -> ModelDesc:
    d = ModelDesc()
    if isinstance(cfg, Qwen3VLConfig):
        t, v = cfg.text, cfg.vision
        d.arch = _lib.AHA_ARCH_QWEN3VL
        d.vis_depth, d.vis_hidden_size, d.vis_num_heads = v.depth, v.hidden_size, v.num_heads
        d.vis_intermediate_size, d.vis_in_channels, d.vis_patch_size = v.intermediate_size, v.in_channels, v.patch_size
        d.vis_temporal_patch_size, d.vis_spatial_merge_size = v.temporal_patch_size, v.spatial_merge_size
        d.vis_out_hidden_size, d.vis_num_position_embeddings = v.out_hidden_size, v.num_position_embeddings
        for i, x in enumerate(v.deepstack_visual_indexes):
            d.vis_deepstack_indexes[i] = x
        d.vis_num_deepstack = len(v.deepstack_visual_indexes)
        d.image_token_id, d.video_token_id = cfg.image_token_id, cfg.video_token_id
        d.vision_start_token_id, d.vision_end_token_id = cfg.vision_start_token_id, cfg.vision_end_token_id
        tie = cfg.tie_word_embeddings
    elif isinstance(cfg, Qwen3ASRConfig):
        t, a = cfg.text, cfg.audio
        d.arch = _lib.AHA_ARCH_QWEN3ASR
        d.aud_d_model, d.aud_encoder_layers, d.aud_attention_heads = a.d_model, a.encoder_layers, a.encoder_attention_heads
        d.aud_ffn_dim, d.aud_num_mel_bins, d.aud_downsample_hidden_size = a.encoder_ffn_dim, a.num_mel_bins, a.downsample_hidden_size
        d.aud_output_dim, d.aud_n_window = a.output_dim, a.n_window
        d.audio_token_id = cfg.audio_token_id
        tie = t.tie_word_embeddings
    else:
        t = cfg
        d.arch = _lib.AHA_ARCH_QWEN3
        tie = cfg.tie_word_embeddings
    d.hidden_size, d.intermediate_size, d.num_hidden_layers = t.hidden_size, t.intermediate_size, t.num_hidden_layers
    d.num_attention_heads, d.num_key_value_heads, d.head_dim = t.num_attention_heads, t.num_key_value_heads, t.head_dim
    d.vocab_size = t.vocab_size
    d.rms_norm_eps, d.rope_theta = t.rms_norm_eps, t.rope_theta
    d.tie_word_embeddings = int(tie)
    ms = t.mrope_section or [0, 0, 0]
    for i in range(3):
        d.mrope_section[i] = ms[i]
    d.kv_reserve_tokens = kv_reserve_tokens
    eos = list(t.eos_token_ids)[:8]
    d.n_stop_tokens = len(eos)
    for i, e in enumerate(eos):
        d.stop_tokens[i] = e
    return d


def tp_unique_id() -> bytes:
    """RCCL unique id for aha_hip_tp_init_rccl: rank 0 makes it, every rank of the TP group receives the same bytes."""
    buf = C.create_string_buffer(128)
    check(lib().aha_hip_tp_unique_id(buf))
    return buf.raw


class HipInferenceModel:
    """One model instance on one GPU (== XxxGenerateModel::init's model object, qwen3/generate.rs:22-50)."""

    def __init__(self, cfg, weights: Dict[str, torch.Tensor], ctx: Optional[HipContext] = None, device: int = 0,
                 kv_reserve_tokens: int = 0, tp_rank: int = 0, tp_size: int = 1, allreduce=None,
                 rccl_unique_id: Optional[bytes] = None, reduce_scatter=None, all_gather=None):
        """tp_size > 1 shards the decoder stack (heads / MLP columns) over ranks; every rank passes the FULL weights and
        the library slices them.  The all-reduce is either RCCL (rccl_unique_id: 128 bytes from tp_unique_id(), shared
        by all ranks) or a host callback allreduce(ptr: int, count: int) -> None that sums `count` f32 at device
        pointer `ptr` over ranks in place (the seam a gloo test or another transport plugs into).
        reduce_scatter(ptr, count_per_rank) / all_gather(ptr, bytes_per_rank) (both or neither; in place, semantics in
        include/aha_hip.h aha_hip_set_seq_parallel) switch the prefill to the sequence-parallel form; with RCCL it is on by
        itself."""
        self.cfg = cfg
        self.text_cfg: Qwen3Config = cfg.text if isinstance(cfg, (Qwen3VLConfig, Qwen3ASRConfig)) else cfg
        self._own_ctx = ctx is None
        self.ctx = ctx or HipContext(device)
        self.handle = C.c_void_p()
        desc = make_desc(cfg, kv_reserve_tokens)
        desc.tp_rank, desc.tp_size = tp_rank, tp_size
        views = (TensorView * len(weights))()
        keep = []
        for i, (name, t) in enumerate(weights.items()):
            t = t.detach().contiguous()
            if t.dtype not in _DT:
                raise TypeError(f"{name}: unsupported dtype {t.dtype}")
            keep.append(t)
            views[i].name = name.encode()
            views[i].data = t.data_ptr()
            views[i].dtype = _DT[t.dtype]
            views[i].ndim = t.dim()
            for j, s in enumerate(t.shape):
                views[i].shape[j] = s
            views[i].on_device = int(t.is_cuda)
        if any(t.is_cuda for t in keep):
            torch.cuda.synchronize()
        check(lib().aha_hip_model_create(self.ctx.handle, C.byref(desc), views, len(weights), C.byref(self.handle)))
        del keep
        self._allreduce_c = None
        if rccl_unique_id is not None:
            buf = C.create_string_buffer(bytes(rccl_unique_id), 128)
            check(lib().aha_hip_tp_init_rccl(self.handle, buf))
        elif tp_size > 1 and allreduce is not None:
            def _cb(ptr, count, _user):
                try:
                    allreduce(int(ptr), int(count))
                    return 0
                except Exception:  # noqa: BLE001 -- must not unwind through C frames
                    import traceback
                    traceback.print_exc()
                    return 1
            self._allreduce_c = _lib.ALLREDUCE_FN(_cb)
            check(lib().aha_hip_set_allreduce(self.handle, self._allreduce_c, None))
            if (reduce_scatter is None) != (all_gather is None):
                raise ValueError("pass both reduce_scatter and all_gather, or neither")
            if reduce_scatter is not None:
                def _wrap(fn):
                    def _c(ptr, n, _user):
                        try:
                            fn(int(ptr), int(n))
                            return 0
                        except Exception:  # noqa: BLE001
                            import traceback
                            traceback.print_exc()
                            return 1
                    return _c
                self._rs_c = _lib.REDUCE_SCATTER_FN(_wrap(reduce_scatter))
                self._ag_c = _lib.ALL_GATHER_FN(_wrap(all_gather))
                check(lib().aha_hip_set_seq_parallel(self.handle, self._rs_c, self._ag_c, None))
        self.vocab = self.text_cfg.vocab_size
        self._logits = np.empty(self.vocab, dtype=np.float32)

    def set_context_parallel(self, rank: int, world: int, all_gather=None, rccl_unique_id: Optional[bytes] = None) -> None:
        """Context-parallel prefill (include/aha_hip.h aha_hip_set_context_parallel): this model holds the FULL weights (tp_size 1) and
        owns two row chunks of every prompt of a fresh cache; per layer the ranks all-gather the layer's K / V pages -- over RCCL
        (rccl_unique_id: 128 bytes from tp_unique_id(), the same on every rank) or through the host callback
        all_gather(ptr: int, bytes_per_rank: int) (in place: rank r's slice is its contribution).  After forward_initial every rank holds
        the whole cache and the last position's logits.  world = 1 switches it off."""
        self._cp_ag_c = None
        if all_gather is not None:
            def _c(ptr, n, _user):
                try:
                    all_gather(int(ptr), int(n))
                    return 0
                except Exception:  # noqa: BLE001 -- must not unwind through C frames
                    import traceback
                    traceback.print_exc()
                    return 1
            self._cp_ag_c = _lib.ALL_GATHER_FN(_c)
        check(lib().aha_hip_set_context_parallel(self.handle, rank, world, self._cp_ag_c, None))
        if rccl_unique_id is not None:
            buf = C.create_string_buffer(bytes(rccl_unique_id), 128)
            check(lib().aha_hip_cp_init_rccl(self.handle, buf))

    @classmethod
    def from_pretrained(cls, path: str, ctx: Optional[HipContext] = None, device: int = 0, kv_reserve_tokens: int = 0):
        """== XxxGenerateModel::init(path, device, dtype) minus tokenizer / chat template (qwen3/generate.rs:22-50): the
        native loader reads config.json + generation_config.json, mmaps every *.safetensors file and builds the model."""
        from .checkpoint import config_from_desc, parse_config
        self = cls.__new__(cls)
        self.desc = parse_config(path)
        self.cfg = config_from_desc(self.desc)
        self.text_cfg = self.cfg.text if isinstance(self.cfg, (Qwen3VLConfig, Qwen3ASRConfig)) else self.cfg
        self._own_ctx = ctx is None
        self.ctx = ctx or HipContext(device)
        self.handle = C.c_void_p()
        self._allreduce_c = None
        check(lib().aha_hip_model_load(self.ctx.handle, os.fsencode(path), kv_reserve_tokens, C.byref(self.handle)))
        self.vocab = int(self.desc.vocab_size)
        self._logits = np.empty(self.vocab, dtype=np.float32)
        return self

    # -- InferenceModel ------------------------------------------------------------------------------------------
    def forward_initial(self, input_ids: Sequence[int], seqlen_offset: int, data: Optional[MultiModalData] = None,
                        want_logits: bool = True):
        ids = np.ascontiguousarray(np.asarray(input_ids, dtype=np.uint32).reshape(-1))
        am = C.c_uint32()
        mm_ref = None
        if data is not None:
            mm = MmInput()
            if data.pixel_values is not None:
                pv = data.pixel_values.detach().contiguous()
                if pv.is_cuda:  # produced on torch's stream; the library copies on its own stream
                    torch.cuda.current_stream(pv.device).synchronize()
                grid = np.ascontiguousarray(np.asarray(data.image_grid_thw, dtype=np.uint32).reshape(-1, 3))
                mm.pixel_values = pv.data_ptr()
                mm.pixel_dtype = _DT[pv.dtype]
                mm.n_patches = pv.shape[0]
                mm.image_grid_thw = grid.ctypes.data_as(C.POINTER(C.c_uint32))
                mm.n_images = grid.shape[0]
            if data.pixel_values_video is not None:
                pvv = data.pixel_values_video.detach().contiguous()
                if pvv.is_cuda:
                    torch.cuda.current_stream(pvv.device).synchronize()
                assert data.pixel_values is None or pvv.dtype == data.pixel_values.dtype, "image and video pixel values share a dtype"
                vgrid = np.ascontiguousarray(np.asarray(data.video_grid_thw, dtype=np.uint32).reshape(-1, 3))
                mm.pixel_values_video = pvv.data_ptr()
                mm.pixel_dtype = _DT[pvv.dtype]
                mm.n_patches_video = pvv.shape[0]
                mm.video_grid_thw = vgrid.ctypes.data_as(C.POINTER(C.c_uint32))
                mm.n_videos = vgrid.shape[0]
            if data.image_embeds is not None:
                ie = data.image_embeds.detach().contiguous()
                assert ie.is_cuda and ie.dtype == torch.bfloat16 and ie.dim() == 3
                torch.cuda.current_stream(ie.device).synchronize()
                grid = np.ascontiguousarray(np.asarray(data.image_grid_thw if data.image_grid_thw is not None else [], dtype=np.uint32).reshape(-1, 3))
                mm.image_embeds = ie.data_ptr()
                mm.n_image_tokens = ie.shape[1]
                mm.image_grid_thw = grid.ctypes.data_as(C.POINTER(C.c_uint32)) if grid.shape[0] else None
                mm.n_images = grid.shape[0]
                if data.video_grid_thw is not None and data.pixel_values_video is None:   # (the grids still drive get_rope_index)
                    vgrid = np.ascontiguousarray(np.asarray(data.video_grid_thw, dtype=np.uint32).reshape(-1, 3))
                    mm.video_grid_thw = vgrid.ctypes.data_as(C.POINTER(C.c_uint32))
                    mm.n_videos = vgrid.shape[0]
            if data.audio_features is not None:
                af = np.ascontiguousarray(np.asarray(data.audio_features, dtype=np.float32))
                mm.audio_features = af.ctypes.data_as(C.POINTER(C.c_float))
                mm.n_frames = af.shape[1]
            if data.audio_samples is not None:
                au = np.ascontiguousarray(np.asarray(data.audio_samples, dtype=np.float32).reshape(-1))
                mm.audio_samples = au.ctypes.data_as(C.POINTER(C.c_float))
                mm.n_samples = au.size
            mm_ref = C.byref(mm)
        lp = self._logits.ctypes.data_as(C.POINTER(C.c_float)) if want_logits else None
        check(lib().aha_hip_forward_initial(self.handle, ids.ctypes.data_as(C.POINTER(C.c_uint32)), ids.size,
                                            seqlen_offset, mm_ref, lp, C.byref(am)))
        return (self._logits.copy() if want_logits else None), int(am.value)

    def forward_step(self, token: int, seqlen_offset: int, want_logits: bool = True):
        am = C.c_uint32()
        lp = self._logits.ctypes.data_as(C.POINTER(C.c_float)) if want_logits else None
        check(lib().aha_hip_forward_step(self.handle, int(token), seqlen_offset, lp, C.byref(am)))
        return (self._logits.copy() if want_logits else None), int(am.value)

    def clear_cache(self):
        check(lib().aha_hip_clear_cache(self.handle))

    def stop_token_ids(self) -> List[int]:
        buf = (C.c_uint32 * 8)()
        n = check(lib().aha_hip_stop_token_ids(self.handle, buf, 8))
        return [int(buf[i]) for i in range(n)]

    # -- extensions ----------------------------------------------------------------------------------------------
    def decode_greedy(self, first_token: int, seqlen_offset: int, max_new: int) -> List[int]:
        buf = (C.c_uint32 * max(max_new, 1))()
        n = check(lib().aha_hip_decode_greedy(self.handle, int(first_token), seqlen_offset, max_new, buf))
        return [int(buf[i]) for i in range(n)]

    def sample_candidates(self, context: Sequence[int], repeat_penalty: float, temperature: float, k: int):
        """Device half of sample_and_push (common/generate.rs:70-86): repeat penalty over `context`, then the k largest
        logits of the last forward call -> (values f32[k], indices u32[k], max, sumexp over the whole vocabulary)."""
        ctx = np.ascontiguousarray(np.asarray(context, dtype=np.uint32).reshape(-1))
        n = max(int(k), 1)  # a bad k is reported by the library, not by numpy
        vals = np.empty(n, dtype=np.float32)
        idx = np.empty(n, dtype=np.uint32)
        mx, se = C.c_float(), C.c_float()
        check(lib().aha_hip_sample_candidates(self.handle, ctx.ctypes.data_as(C.POINTER(C.c_uint32)), ctx.size,
                                              float(repeat_penalty), float(temperature), int(k),
                                              vals.ctypes.data_as(C.POINTER(C.c_float)),
                                              idx.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(mx), C.byref(se)))
        return vals, idx, float(mx.value), float(se.value)

    def last_logits(self) -> np.ndarray:
        out = np.empty(self.text_cfg.vocab_size, dtype=np.float32)
        check(lib().aha_hip_last_logits(self.handle, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def debug_allreduce(self, t: torch.Tensor) -> None:
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        torch.cuda.current_stream(t.device).synchronize()
        check(lib().aha_hip_debug_allreduce(self.handle, t.data_ptr(), t.numel()))

    # -- TextEmbedding / TextRerank (common/embedding.rs:7-9, common/reranker.rs:5-7) ------------------------------------
    def embed_one(self, input_ids: Sequence[int]) -> np.ndarray:
        """Qwen3Embedding::embed_one after tokenisation (qwen3_embedding/mod.rs:50-64): L2-normalised last hidden state."""
        ids = np.ascontiguousarray(np.asarray(input_ids, dtype=np.uint32).reshape(-1))
        out = np.empty(self.text_cfg.hidden_size, dtype=np.float32)
        check(lib().aha_hip_embed(self.handle, ids.ctypes.data_as(C.POINTER(C.c_uint32)), ids.size,
                                  out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def embed_multi(self, inputs: Sequence[Sequence[int]]) -> np.ndarray:
        if len(inputs) == 0:
            raise ValueError("embedding input cannot be empty")  # qwen3_embedding/mod.rs:39-41
        return np.stack([self.embed_one(x) for x in inputs], 0)

    def rerank(self, query_ids: Sequence[int], documents_ids: Sequence[Sequence[int]]) -> np.ndarray:
        """Qwen3Reranker::rerank (qwen3_reranker/mod.rs:23-31): cosine_similarity_no_l2(query, docs) on normalised vectors."""
        q = self.embed_one(query_ids)[None, :]
        return (q @ self.embed_multi(documents_ids).T)[0]

    def cache_len(self) -> int:
        return int(lib().aha_hip_cache_len(self.handle))

    def kv_export(self):
        """aha_hip_kv_export: -> (uint8 tensor on this model's GPU holding [layer][page][local kv head][K | V] blocks, n_tokens,
        rope_delta).  The byte image of this rank's share of the cache -- what crosses the gather after a sharded prefill."""
        need, ntok, delta = C.c_size_t(), C.c_size_t(), C.c_int64()
        check(lib().aha_hip_kv_export(self.handle, None, 0, C.byref(need), C.byref(ntok), C.byref(delta)))
        buf = torch.empty(need.value, dtype=torch.uint8, device=f"cuda:{self.ctx.device}")
        torch.cuda.current_stream(buf.device).synchronize()
        check(lib().aha_hip_kv_export(self.handle, buf.data_ptr(), buf.numel(), None, None, None))
        return buf, int(ntok.value), int(delta.value)

    def kv_import(self, buf: torch.Tensor, src_heads: int, src_head0: int, dst_head0: int, n_heads: int, n_tokens: int, rope_delta: int):
        """aha_hip_kv_import: heads [src_head0, +n_heads) of a packed buffer (src_heads per page) -> this model's heads [dst_head0, ..)."""
        assert buf.is_cuda and buf.dtype == torch.uint8 and buf.is_contiguous()
        torch.cuda.current_stream(buf.device).synchronize()
        check(lib().aha_hip_kv_import(self.handle, buf.data_ptr(), buf.numel(), src_heads, src_head0, dst_head0, n_heads, n_tokens, rope_delta))

    def debug_graph_step(self, replays: int = 50):
        """(us per decode step enqueued launch by launch, us per step replayed as one hipGraph) at the current cache length; clears the cache."""
        a, b = C.c_double(), C.c_double()
        check(lib().aha_hip_debug_graph_step(self.handle, replays, C.byref(a), C.byref(b)))
        return a.value, b.value

    def debug_steps_executed(self) -> int:
        return int(lib().aha_hip_debug_steps_executed(self.handle))

    def set_profiling(self, on: bool):
        check(lib().aha_hip_set_profiling(self.handle, int(on)))

    def get_profile(self, kernel_class: str):
        ms, n, b, f = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        check(lib().aha_hip_get_profile(self.handle, kernel_class.encode(), C.byref(ms), C.byref(n), C.byref(b), C.byref(f)))
        return {"ms": ms.value, "launches": n.value, "bytes": b.value, "flops": f.value}

    def debug_scramble_pages(self, on: bool = True):
        check(lib().aha_hip_debug_scramble_pages(self.handle, int(on)))

    def debug_last_hidden(self) -> np.ndarray:
        out = np.empty(self.text_cfg.hidden_size, dtype=np.float32)
        check(lib().aha_hip_debug_last_hidden(self.handle, out.ctypes.data_as(C.POINTER(C.c_float)), out.size))
        return out

    def debug_image_embeds(self, which: int, rows: int) -> np.ndarray:
        out = np.empty((rows, self.text_cfg.hidden_size), dtype=np.float32)
        check(lib().aha_hip_debug_image_embeds(self.handle, which, out.ctypes.data_as(C.POINTER(C.c_float)), out.size))
        return out

    def vision_encode(self, data: "MultiModalData") -> torch.Tensor:
        """ViT only (aha_hip_vision_encode): -> (1 + n_deepstack, n_tokens, hidden) bf16 on this model's GPU; the images' tokens
        first, then the videos'."""
        mm = MmInput()
        m2 = self.cfg.vision.spatial_merge_size ** 2
        n_tok = 0
        keep = []
        if data.pixel_values is not None:
            pv = data.pixel_values.detach().contiguous()
            if pv.is_cuda:
                torch.cuda.current_stream(pv.device).synchronize()
            grid = np.ascontiguousarray(np.asarray(data.image_grid_thw, dtype=np.uint32).reshape(-1, 3))
            mm.pixel_values, mm.pixel_dtype, mm.n_patches = pv.data_ptr(), _DT[pv.dtype], pv.shape[0]
            mm.image_grid_thw, mm.n_images = grid.ctypes.data_as(C.POINTER(C.c_uint32)), grid.shape[0]
            n_tok += int(sum(int(g[0]) * int(g[1]) * int(g[2]) for g in grid) // m2)
            keep += [pv, grid]
        if data.pixel_values_video is not None:
            pvv = data.pixel_values_video.detach().contiguous()
            if pvv.is_cuda:
                torch.cuda.current_stream(pvv.device).synchronize()
            vgrid = np.ascontiguousarray(np.asarray(data.video_grid_thw, dtype=np.uint32).reshape(-1, 3))
            mm.pixel_values_video, mm.pixel_dtype, mm.n_patches_video = pvv.data_ptr(), _DT[pvv.dtype], pvv.shape[0]
            mm.video_grid_thw, mm.n_videos = vgrid.ctypes.data_as(C.POINTER(C.c_uint32)), vgrid.shape[0]
            n_tok += int(sum(int(g[0]) * int(g[1]) * int(g[2]) for g in vgrid) // m2)
            keep += [pvv, vgrid]
        k = 1 + len(self.cfg.vision.deepstack_visual_indexes)
        out = torch.empty(k, n_tok, self.text_cfg.hidden_size, dtype=torch.bfloat16, device=f"cuda:{self.ctx.device}")
        nt = C.c_int64()
        check(lib().aha_hip_vision_encode(self.handle, C.byref(mm), out.data_ptr(), C.byref(nt)))
        assert nt.value == n_tok
        return out

    def debug_audio_embeds(self, rows: int) -> np.ndarray:
        out = np.empty((rows, self.text_cfg.hidden_size), dtype=np.float32)
        check(lib().aha_hip_debug_audio_embeds(self.handle, out.ctypes.data_as(C.POINTER(C.c_float)), out.size))
        return out

    def close(self):
        if self.handle:
            lib().aha_hip_model_destroy(self.handle)
            self.handle = C.c_void_p()
        if self._own_ctx:
            self.ctx.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class Usage:
    """utils/response_utils.rs:225-255 / params/shared.rs:3-28 timing fields."""
    prompt_tokens: int
    prompt_secs: float
    completion_tokens: int
    completion_secs: float

    @property
    def completion_tps(self) -> float:
        return self.completion_tokens / self.completion_secs if self.completion_secs > 0 else float("nan")


def generate_generic(model: HipInferenceModel, input_ids: Sequence[int], max_tokens: int,
                     data: Optional[MultiModalData] = None, device_loop: bool = False):
    """generate_generic (common/generate.rs:115-159) with temperature 0 => Sampling::ArgMax (sample.rs:13-37).

    Returns (generated token ids, Usage).  ``device_loop`` uses the aha_hip_decode_greedy extension for the decode
    loop (no per-token host round trip); the token sequence is identical by construction.
    """
    eos = set(model.stop_token_ids())
    generated: List[int] = []
    seqlen_offset, seq_len = 0, len(input_ids)
    t0 = time.perf_counter()
    _, tok = model.forward_initial(input_ids, seqlen_offset, data, want_logits=False)
    generated.append(tok)
    prompt_secs = time.perf_counter() - t0
    t0 = time.perf_counter()
    if device_loop and max_tokens > 1:
        seqlen_offset += seq_len
        generated += model.decode_greedy(tok, seqlen_offset, max_tokens - 1)
    else:
        for _ in range(1, max_tokens):
            seqlen_offset += seq_len
            seq_len = 1
            _, tok = model.forward_step(tok, seqlen_offset, want_logits=False)
            generated.append(tok)
            if tok in eos:
                break
    completion_secs = time.perf_counter() - t0
    model.clear_cache()
    return generated, Usage(len(input_ids), prompt_secs, len(generated), completion_secs)
