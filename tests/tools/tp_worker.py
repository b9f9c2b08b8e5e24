"""Worker of tests/test_tp_gpu.py::test_tp2_two_processes_gloo: one of WORLD_SIZE processes that share the test box's single
GPU, each holding one tensor-parallel rank of a tiny Qwen3-VL model.  The all-reduce seam (aha_hip_set_allreduce) is a
callback that stages the f32 partial sums through host memory and sums them with torch.distributed (gloo); the ViT runs
image-parallel with one gloo all-gather (aha_amd.parallel.encode_images_sharded).  Rank 0 also holds the unsharded model and
checks that the sharded stack reproduces its logits (f32 summation order apart) and its greedy tokens."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    from aha_amd import parallel
    from aha_amd.configs import tiny_qwen3vl
    from aha_amd.model import HipInferenceModel, MultiModalData
    from aha_amd.weights import qwen3vl_weights
    from oracle.numerics import Numerics
    from oracle import qwen3vl as ov

    cfg = tiny_qwen3vl()
    w = qwen3vl_weights(cfg, seed=0)
    calls = [0]

    def allreduce(ptr, count):
        iface = {"shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 2}
        dev = torch.as_tensor(type("H", (), {"__cuda_array_interface__": iface})(), device="cuda:0")
        host = dev.cpu()
        dist.all_reduce(host)
        dev.copy_(host)
        torch.cuda.synchronize()
        calls[0] += 1

    sp_calls = [0, 0]

    def dev_view(ptr, n, typestr):
        iface = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}
        return torch.as_tensor(type("H", (), {"__cuda_array_interface__": iface})(), device="cuda:0")

    def reduce_scatter(ptr, count_per_rank):   # in place: this rank's slice receives the sum, staged through host memory
        dev = dev_view(ptr, count_per_rank * world, "<f4")
        host = dev.cpu()
        dist.all_reduce(host)                  # gloo has no reduce_scatter: all_reduce on the host copy, keep this rank's slice
        dev.fill_(float("nan"))                # the other slices are undefined by contract
        dev[rank * count_per_rank:(rank + 1) * count_per_rank].copy_(host[rank * count_per_rank:(rank + 1) * count_per_rank])
        torch.cuda.synchronize()
        sp_calls[0] += 1

    def all_gather(ptr, bytes_per_rank):
        dev = dev_view(ptr, bytes_per_rank * world, "|u1")
        mine = dev[rank * bytes_per_rank:(rank + 1) * bytes_per_rank].cpu()
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        dev.copy_(torch.cat(parts))
        torch.cuda.synchronize()
        sp_calls[1] += 1

    sp = os.environ.get("TP_WORKER_SP", "1") != "0"
    m = HipInferenceModel(cfg, w, tp_rank=rank, tp_size=world, allreduce=allreduce,
                          reduce_scatter=reduce_scatter if sp else None, all_gather=all_gather if sp else None)
    g = np.random.default_rng(7)
    imgs = [g.integers(0, 256, size=(h, wd, 3), dtype=np.uint8) for (h, wd) in [(96, 160), (64, 64), (128, 96)]]
    pv, grid = ov.process_images(Numerics("bf16"), imgs)
    ids = [int(x) for x in g.integers(0, 1900, size=3)]
    for gi in grid.tolist():
        ids += [cfg.vision_start_token_id] + [cfg.image_token_id] * (gi[0] * gi[1] * gi[2] // 4) + [cfg.vision_end_token_id]
    ids += [int(x) for x in g.integers(0, 1900, size=40)]
    pvb = pv.to(torch.bfloat16)
    toks = [int(a * b * c) // 4 for a, b, c in grid.tolist()]
    patches = np.cumsum([0] + [int(a * b * c) for a, b, c in grid.tolist()])

    def enc(idx):
        a, b = idx[0], idx[-1] + 1
        return m.vision_encode(MultiModalData(pvb[patches[a]:patches[b]].cuda(), grid[a:b])).cpu()   # gloo gathers host tensors

    emb = parallel.encode_images_sharded(enc, list(range(len(imgs))), toks, world, rank).cuda().contiguous()
    got, tok = m.forward_initial(ids, 0, MultiModalData(image_grid_thw=grid, image_embeds=emb))
    # KV hand-back: the head-sharded cache of the TP prefill gathered into an UN-sharded model on rank 0,
    # which then decodes alone (north_star: decode stays single-GPU); checked below against a model prefilled on one GPU
    handback = HipInferenceModel(cfg, w) if rank == 0 else None
    n_cached = parallel.gather_kv_to_rank0(m, handback, rank, world)
    assert (n_cached == len(ids)) if rank == 0 else (n_cached is None)
    dec = [tok]
    off = len(ids)
    for _ in range(6):
        lg, t = m.forward_step(dec[-1], off)
        dec.append(t)
        off += 1
    assert calls[0] > 0, "the all-reduce seam was never used"
    if sp:
        assert sp_calls[0] > 0 and sp_calls[1] > 0, "the sequence-parallel seam was never used"
    # the per-phase diagnosis bench.py attaches to its sharded-prefill object (parallel.sharded_prefill phases_out): one more prefill
    # with the library profiler on; every phase of THIS path must show up with a time and a launch count, and the token must not move
    ph = {}
    emb2 = parallel.encode_images_sharded(enc, list(range(len(imgs))), toks, world, rank, timings=ph).cuda().contiguous()
    assert torch.equal(emb2, emb)
    m.clear_cache()
    m.set_profiling(True)
    _, tok2 = m.forward_initial(ids, 0, MultiModalData(image_grid_thw=grid, image_embeds=emb2), want_logits=False)
    parallel.read_prefill_phases(m, ph)
    m.set_profiling(False)
    assert tok2 == tok
    want_ph = ["vit_s", "embeds_all_gather_s", "gemm_s", "attn_s", "rowwise_s", "lm_head_s"] + (["reduce_scatter_s", "all_gather_s"] if sp else ["allreduce_s"])
    assert all(k in ph and ph[k] >= 0.0 for k in want_ph), (want_ph, ph)
    L = cfg.text.num_hidden_layers
    if sp:
        assert ph["reduce_scatter_launches"] == 2 * L and ph["all_gather_launches"] >= 2 * L, ph
    m.clear_cache()
    m.forward_initial(ids, 0, MultiModalData(image_grid_thw=grid, image_embeds=emb), want_logits=False)   # the cache the checks below continue from
    all_dec = [None] * world
    dist.all_gather_object(all_dec, dec)
    assert all(d == all_dec[0] for d in all_dec), f"ranks disagree on the greedy tokens: {all_dec}"
    if rank == 0:
        single = HipInferenceModel(cfg, w)
        ref, rtok = single.forward_initial(ids, 0, MultiModalData(pvb, grid))
        s = float(ref.std())
        assert float(np.abs(got - ref).max()) <= 0.02 * s, f"TP prefill logits: {np.abs(got - ref).max() / s:.4f} std"
        rdec = [rtok]
        o2 = len(ids)
        for _ in range(6):
            rl, t = single.forward_step(rdec[-1], o2)
            rdec.append(t)
            o2 += 1
        s2 = float(rl.std())
        assert float(np.abs(lg - rl).max()) <= 0.03 * s2 or dec != rdec, "TP decode logits drifted"
        margin_ok = dec == rdec
        # single-GPU decode on the gathered cache: position / rope_delta / cache length taken over by aha_hip_kv_import
        assert handback.cache_len() == len(ids)
        hdec, o3 = [rtok], len(ids)
        for i in range(6):                     # teacher-forced with the single-GPU run's tokens: same inputs on both sides
            hl, t = handback.forward_step(rdec[i], o3)
            hdec.append(t)
            o3 += 1
        assert float(np.abs(hl - rl).max()) <= 0.03 * s2, "decode on the gathered KV cache drifted from the single-GPU run"
        print(f"TP_WORKER_OK tokens_equal={margin_ok} handback_tokens_equal={hdec == rdec} allreduce_calls={calls[0]} "
              f"reduce_scatter_calls={sp_calls[0]} all_gather_calls={sp_calls[1]} phases={sorted(k for k in ph if k.endswith('_s'))}", flush=True)
        handback.close()
        single.close()
    m.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
