"""Multi-GPU plumbing for the one part of the path that shards without a collective: independent requests.

BASELINE.json north_star: decode stays single-GPU (weights 15 GB << 288 GB), so N GPUs serve N independent requests
(replicas, "scaling": "weak"); only the timing is combined across ranks (max over ranks, as the bench contract asks).
The two parts that DO shard are driven from here as well: the ViT by images (independent units,
one all-gather of the embeddings: `encode_images_sharded`) and the long-context prefill by tensor parallelism inside the
library (`sharded_prefill`: one TP group over all ranks, all-reduce over RCCL).
One process per GPU; `torch.distributed` backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple


def shard_units(n_units: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block partition of n_units independent units (requests / images) over ranks: [start, end)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(n_units, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def init_process_group(backend: str = "nccl", device=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if dist.is_initialized():
        return dist
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, **kw)
    return dist


def aggregate_throughput(units_this_rank: float, secs_this_rank: float, device="cpu") -> Tuple[float, float]:
    """Whole-job throughput = (sum over ranks of units) / (max over ranks of seconds).  Returns (value, max_secs)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return units_this_rank / secs_this_rank, secs_this_rank
    t = torch.tensor([secs_this_rank], dtype=torch.float64, device=device)
    u = torch.tensor([units_this_rank], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()) / float(t.item()), float(t.item())


def _sync_clock(device=None) -> float:
    """perf_counter after the device's queues have drained (phase boundaries of the diagnostic pass; never inside a timed region)."""
    import time
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize(device)
    return time.perf_counter()


def encode_images_sharded(encode_fn, per_image_inputs: List, tokens_per_image: List[int], world: int, rank: int, meta=None,
                          timings: Optional[dict] = None):
    """Image-parallel ViT: images are independent units (block-diagonal attention per image,
    /root/reference/src/models/qwen3vl/model.rs:258-273), so rank r encodes images shard_units(n, world, r) and ONE
    all_gather moves the embeddings.  ``encode_fn(list_of_inputs) -> (K, n_local_tokens, H)`` tensor (device for nccl,
    CPU for gloo); returns (K, total_tokens, H) in image order on every rank.  Ragged shards are padded to the largest
    shard for the collective and trimmed afterwards.  ``meta = (K, H, dtype, device)`` of the encoder's output, known from the
    config: a rank without images then builds its padding on ITS OWN device and the timed path has no pickle collective
    (without it the shapes are agreed with one all_gather_object and the padding goes to this rank's current device).
    ``timings`` (diagnostic pass only: it synchronises the device at the phase boundaries) receives ``vit_s`` = this rank's encoder
    time and ``embeds_all_gather_s`` = padding + collective + trim."""
    import torch
    import torch.distributed as dist
    n = len(per_image_inputs)
    a, b = shard_units(n, world, rank)
    t0 = _sync_clock() if timings is not None else 0.0
    local = encode_fn(per_image_inputs[a:b]) if b > a else None
    if timings is not None:
        t1 = _sync_clock()
        timings["vit_s"] = timings.get("vit_s", 0.0) + (t1 - t0)
        timings["vit_images_this_rank"] = b - a
    if world == 1 or not dist.is_initialized():
        return local
    counts = [sum(tokens_per_image[slice(*shard_units(n, world, r))]) for r in range(world)]
    mx = max(counts)
    if local is not None:
        K, H, dtype, device = local.shape[0], local.shape[2], local.dtype, local.device
    elif meta is not None:
        K, H, dtype, device = meta
    if meta is None:
        shapes = [None] * world
        dist.all_gather_object(shapes, None if local is None else (local.shape[0], local.shape[2], local.dtype))
        if local is None:   # this rank has no image: it still takes part in the collective, with a buffer on its OWN device
            K, H, dtype = next(s for s in shapes if s is not None)
            device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    buf = torch.zeros(K, mx, H, dtype=dtype, device=device)
    if local is not None:
        buf[:, : local.shape[1]] = local
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    res = torch.cat([o[:, :c] for o, c in zip(outs, counts) if c > 0], dim=1)
    if timings is not None:
        timings["embeds_all_gather_s"] = timings.get("embeds_all_gather_s", 0.0) + (_sync_clock() - t1)
        timings["embeds_all_gather_bytes"] = int(buf.numel() * buf.element_size() * world)
    return res


def broadcast_bytes(payload, src: int = 0):
    """Rank `src`'s bytes object on every rank (the 128-byte RCCL unique id travels this way)."""
    import torch.distributed as dist
    box = [payload]
    dist.broadcast_object_list(box, src=src)
    return box[0]


# library profiler classes (csrc/model.hip ProfScope: HIP events on the model's stream around each launch group) -> phase names
PREFILL_PHASE_CLASSES = (("gemm", "gemm_s"), ("attn_prefill", "attn_s"), ("attn_vit", "vit_attn_s"), ("elem", "rowwise_s"),
                         ("reduce_scatter", "reduce_scatter_s"), ("rs_wait", "reduce_scatter_wait_s"), ("all_gather", "all_gather_s"),
                         ("ag_wait", "all_gather_wait_s"), ("cp_kv_gather", "kv_all_gather_s"), ("allreduce", "allreduce_s"), ("gemv", "lm_head_s"))


def read_prefill_phases(model, ph: dict) -> dict:
    """Seconds per profiler class of the forward calls since model.set_profiling(True), under the phase names of the bench line."""
    for cls, name in PREFILL_PHASE_CLASSES:
        pr = model.get_profile(cls)
        if pr["launches"]:
            ph[name] = round(pr["ms"] * 1e-3, 5)
            ph[name[:-2] + "_launches"] = int(pr["launches"])
    return ph


def sharded_prefill(cfg, weights, input_ids, data, rank: int, world: int, device_index: int, kv_reserve_tokens: int = 0,
                    repeats: int = 1, phases_out: Optional[dict] = None, mode: str = "tp"):
    """BASELINE cfg 5's sharded path on `world` GPUs of one node, one process per GPU:
      * mode "tp": the ranks form ONE tensor-parallel group: every rank passes the full checkpoint, the library keeps its q/k/v heads,
        gate/up rows and o/down columns, and all-reduces the row-parallel partial sums over RCCL (aha_hip_tp_init_rccl);
        mode "cp" (context-parallel, include/aha_hip.h aha_hip_set_context_parallel): every rank keeps the FULL weights and owns two
        row chunks of the prompt; the only exchange in the decoder stack is one all-gather of the layer's K / V pages per layer, every
        rank ends with the whole cache (no hand-back) and the lm_head runs replicated;
      * the ViT runs image-parallel: rank r encodes images shard_units(n, world, r), one all_gather moves the merged +
        DeepStack embeddings (encode_images_sharded), and every rank scatters all of them into its prompt;
      * lm_head is vocabulary-parallel inside the library (arg-max pair exchange).
    Returns (first greedy token, seconds of the slowest of `repeats` timed prefills on this rank, model).  world == 1 is the
    plain single-GPU call, so the N = 1 value of a scaling curve is the single-GPU cfg 5 prefill.
    ``phases_out`` (a dict): ONE more, untimed prefill runs with the library profiler on and the device synchronised at the host-side
    phase boundaries, and the dict receives this rank's seconds per phase -- ViT encode, embeddings all-gather, and inside the decoder
    stack (HIP events on the model's stream): GEMMs, attention, row-wise kernels, blocking reduce-scatters, the wait for the
    reduce-scatters overlapped on the communication stream, all-gathers of the normalised rows, all-reduces, lm_head -- plus
    ``stack_s`` (wall clock of forward_initial) so that what the events do not cover (host gaps, page mapping) shows as the
    difference.  A first multi-GPU run then says WHERE the time went, not only how long it took."""
    import time
    import numpy as np
    import torch
    from .model import HipInferenceModel, MultiModalData, tp_unique_id
    uid = None
    if world > 1:
        uid = broadcast_bytes(tp_unique_id() if rank == 0 else None)
    if mode not in ("tp", "cp"):
        raise ValueError("mode must be 'tp' or 'cp'")
    if mode == "cp":
        model = HipInferenceModel(cfg, weights, device=device_index, kv_reserve_tokens=kv_reserve_tokens)
        if world > 1:
            model.set_context_parallel(rank, world, rccl_unique_id=uid)
    else:
        model = HipInferenceModel(cfg, weights, device=device_index, kv_reserve_tokens=kv_reserve_tokens,
                                  tp_rank=rank if world > 1 else 0, tp_size=world, rccl_unique_id=uid)
    grid = None if data is None else np.asarray(data.image_grid_thw, dtype=np.uint32).reshape(-1, 3)

    def one_prefill(ph=None):
        model.clear_cache()
        mm = data
        if data is not None and world > 1:
            m2 = cfg.vision.spatial_merge_size ** 2
            toks = [int(g[0]) * int(g[1]) * int(g[2]) // m2 for g in grid]
            patches = np.cumsum([0] + [int(g[0]) * int(g[1]) * int(g[2]) for g in grid])

            def enc(idx_range):
                a, b = idx_range[0], idx_range[-1] + 1
                return model.vision_encode(MultiModalData(data.pixel_values[patches[a]:patches[b]], grid[a:b]))
            meta = (1 + len(cfg.vision.deepstack_visual_indexes), cfg.text.hidden_size, torch.bfloat16, torch.device("cuda", device_index))
            emb = encode_images_sharded(lambda idx: enc(idx), list(range(len(grid))), toks, world, rank, meta=meta, timings=ph)
            mm = MultiModalData(image_grid_thw=grid, image_embeds=emb.contiguous())
        if ph is None:
            _, tok = model.forward_initial(input_ids, 0, mm, want_logits=False)
            return tok
        t0 = _sync_clock()
        model.set_profiling(True)
        _, tok = model.forward_initial(input_ids, 0, mm, want_logits=False)
        read_prefill_phases(model, ph)
        model.set_profiling(False)
        ph["stack_s"] = round(_sync_clock() - t0, 5)
        for k in ("vit_s", "embeds_all_gather_s"):
            if k in ph:
                ph[k] = round(ph[k], 5)
        return tok

    tok = one_prefill()   # warm (page allocation, RCCL channels)
    worst = 0.0
    for _ in range(repeats):
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tok = one_prefill()
        torch.cuda.synchronize()
        worst = max(worst, time.perf_counter() - t0)
    if phases_out is not None:
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        tok_d = one_prefill(phases_out)
        phases_out["first_token_equal_to_timed_run"] = bool(tok_d == tok)
    return tok, worst, model


def gather_kv_to_rank0(tp_model, full_model, rank: int, world: int):
    """KV hand-back after a sharded prefill (north_star: decode stays single-GPU): every rank packs the
    K / V of its kv heads (aha_hip_kv_export: [layer][page][head][K | V] blocks, the byte image of its pages), ONE gather moves the
    packs to rank 0 (RCCL over xGMI under the nccl backend: 738 MB per rank at 41 k tokens of the 8B model; gloo moves host copies),
    and rank 0's un-sharded model (`full_model`: all heads, all weights -- 15 GB next to 288 GB) scatters rank r's heads into its own
    pages at head offset r * kv_heads / world (aha_hip_kv_import), taking over the cache length and the rope_delta.  Rank 0 then
    decodes alone, exactly as after a single-GPU prefill.  Returns the number of cached tokens (rank 0) or None."""
    import torch
    import torch.distributed as dist
    buf, n_tokens, delta = tp_model.kv_export()
    if world == 1 or not dist.is_initialized():
        if full_model is not None and full_model is not tp_model:
            kvh = full_model.text_cfg.num_key_value_heads
            full_model.kv_import(buf, kvh, 0, 0, kvh, n_tokens, delta)
        return n_tokens
    on_host = dist.get_backend() != "nccl"
    send = buf.cpu() if on_host else buf
    parts = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
    dist.gather(send, parts, dst=0)
    if rank != 0:
        return None
    kvh = full_model.text_cfg.num_key_value_heads
    per = kvh // world
    for r, part in enumerate(parts):
        part = part.to(buf.device) if on_host else part
        full_model.kv_import(part, per, 0, r * per, per, n_tokens, delta)
    return n_tokens
