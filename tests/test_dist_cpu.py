"""CPU (-m "not gpu"): the N > 1 path with world_size 2 over gloo -- request sharding and the bench aggregation
(sum of units / max of seconds), exactly the code bench.py runs over RCCL."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from aha_amd import parallel
    dist = parallel.init_process_group("gloo")
    a, b = parallel.shard_units(5, world, rank)
    units = float(b - a) * 10.0            # e.g. tokens produced by this rank's requests
    secs = 1.0 + rank                      # rank 1 is the slow one
    value, tmax = parallel.aggregate_throughput(units, secs)
    out.put((rank, a, b, value, tmax))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_replicas_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, a0, b0, v0, t0), (r1, a1, b1, v1, t1) = res
    assert (a0, b0, a1, b1) == (0, 3, 3, 5)
    assert t0 == t1 == 2.0                       # max over ranks
    assert abs(v0 - 50.0 / 2.0) < 1e-9 and v0 == v1   # whole-job units / slowest rank's time


def _gather_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from aha_amd import parallel
    dist = parallel.init_process_group("gloo")
    toks = [3, 5, 2]                                   # three images with different token counts, two ranks
    def encode(imgs):                                  # stand-in for HipInferenceModel.vision_encode
        return torch.cat([torch.full((2, toks[i], 4), float(i)) + torch.arange(toks[i]).float()[None, :, None] / 10 for i in imgs], 1)
    res = parallel.encode_images_sharded(encode, [0, 1, 2], toks, world, rank)
    # fewer images than ranks: rank 1 has nothing to encode and pads on its OWN device; with `meta` (K, H, dtype, device known from
    # the config) there is no pickle collective at all, without it the shapes are agreed first -- same result either way
    one_a = parallel.encode_images_sharded(encode, [1], [toks[1]], world, rank, meta=(2, 4, torch.float32, torch.device("cpu")))
    one_b = parallel.encode_images_sharded(encode, [1], [toks[1]], world, rank)
    assert torch.equal(one_a, one_b) and tuple(one_a.shape) == (2, 5, 4) and one_a.device.type == "cpu"
    assert abs(float(one_a[0, 3, 0]) - 1.3) < 1e-6
    # the diagnostic pass (bench.py sharded_prefill "phases"): encoder seconds and gather seconds of THIS rank, same result
    tm = {}
    timed = parallel.encode_images_sharded(encode, [0, 1, 2], toks, world, rank, timings=tm)
    assert torch.equal(timed, res) and tm["vit_s"] >= 0.0 and tm["embeds_all_gather_s"] >= 0.0
    assert tm["vit_images_this_rank"] == (2 if rank == 0 else 1) and tm["embeds_all_gather_bytes"] == 2 * 8 * 4 * 4 * world
    out.put((rank, res.shape, res[0, :, 0].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_image_parallel_gather_gloo():
    """Image-parallel ViT plumbing: ragged shards, one all_gather, image order preserved on every rank."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [0.0, 0.1, 0.2, 1.0, 1.1, 1.2, 1.3, 1.4, 2.0, 2.1]
    for rank, shape, col in res:
        assert tuple(shape) == (2, 10, 4)
        assert all(abs(a - b) < 1e-6 for a, b in zip(col, want))


def _uid_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from aha_amd import parallel
    dist = parallel.init_process_group("gloo")
    # the RCCL unique id of the TP group: made on rank 0 (a host-only call of the library), identical bytes on every rank
    from aha_amd.model import tp_unique_id
    uid = parallel.broadcast_bytes(tp_unique_id() if rank == 0 else None)
    # the strong-scaling aggregation of bench.py's sharded prefill: every rank contributes tokens / world, slowest rank's time
    value, tmax = parallel.aggregate_throughput(40980.0 / world, 1.5 + 0.25 * rank)
    out.put((rank, uid, value, tmax))
    dist.barrier()
    dist.destroy_process_group()


def test_tp_group_setup_and_strong_scaling_aggregate_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_uid_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, u0, v0, t0), (_, u1, v1, t1) = res
    assert isinstance(u0, bytes) and len(u0) == 128 and u0 == u1 and any(u0)
    assert t0 == t1 == 1.75 and abs(v0 - 40980.0 / 1.75) < 1e-6 and v0 == v1
