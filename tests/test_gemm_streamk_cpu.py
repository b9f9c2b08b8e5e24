"""CPU tier: the host planner of the persistent GEMM kernel (csrc/kernels_gemm_sk.hip).  The kernel only walks the segment lists it is
handed, so the plan IS the correctness contract: every (tile, K tile) of the GEMM must be covered exactly once, the pieces of a cut tile
must carry consistent piece counts / slots in K order / chunk ranges that do not overlap another tile's, and the whole tiles of the
full rounds must follow the XCD-aware order (worker b serves XCD b & 7).  Host-only query: aha_hip_debug_streamk_plan."""
import ctypes

import numpy as np
import pytest

from aha_amd import _lib


def plan(M, N, K, tile_n=256, workers=256, ws=1 << 30):
    lib = _lib.lib()
    cap = 1 << 17
    out = (ctypes.c_int32 * (8 * cap))()
    off = (ctypes.c_int32 * (workers + 1))()
    info = (ctypes.c_int32 * 7)()
    n = lib.aha_hip_debug_streamk_plan(M, N, K, tile_n, workers, ws, out, cap, off, info)
    assert 0 <= n <= cap, (n, _lib.last_error() if hasattr(_lib, "last_error") else "")
    segs = np.frombuffer(out, dtype=np.int32)[: 8 * n].reshape(n, 8).copy()
    return segs, np.array(off[:], dtype=np.int64), dict(zip(("workers", "chunks", "counters", "split", "style", "cuts", "makespan_x1000"), info))


def check(M, N, K, tile_n, workers, ws=1 << 30):
    segs, off, info = plan(M, N, K, tile_n, workers, ws)
    nk = K // 64
    ntm, ntn = -(-M // 256), -(-N // tile_n)
    assert off[0] == 0 and off[-1] == len(segs) and np.all(np.diff(off) >= 0) and info["workers"] == workers
    cover = np.zeros((ntm, ntn, nk), dtype=np.int32)
    tiles = {}
    for b in range(workers):
        for m0, n0, kt0, kt1, nparts, slot, chunk, ctr in segs[off[b]: off[b + 1]]:
            assert m0 % 256 == 0 and n0 % tile_n == 0 and 0 <= m0 < M and 0 <= n0 < N and 0 <= kt0 < kt1 <= nk
            cover[m0 // 256, n0 // tile_n, kt0:kt1] += 1
            assert 1 <= nparts <= 4 and 0 <= slot < nparts
            if nparts == 1:
                assert (kt0, kt1) == (0, nk), "an uncut tile runs its whole K range"
            tiles.setdefault((m0, n0), []).append((slot, kt0, kt1, nparts, chunk, ctr, b))
    assert np.all(cover == 1), f"{int((cover != 1).sum())} (tile, K tile) cells not covered exactly once"
    used_chunks, used_ctrs = set(), set()
    n_split = 0
    for (m0, n0), ps in tiles.items():
        ps.sort()
        nparts = ps[0][3]
        assert len(ps) == nparts and [p[0] for p in ps] == list(range(nparts)), "slots 0..P-1, one piece each"
        assert all(p[3] == nparts and p[4] == ps[0][4] and p[5] == ps[0][5] for p in ps)
        assert ps[0][1] == 0 and ps[-1][2] == nk and all(a[2] == b[1] for a, b in zip(ps, ps[1:])), "slots are in K order and abut"
        if nparts > 1:
            n_split += 1
            rng = set(range(ps[0][4], ps[0][4] + nparts))
            assert not (rng & used_chunks) and ps[0][5] not in used_ctrs
            used_chunks |= rng
            used_ctrs.add(ps[0][5])
            assert len({p[6] & 7 for p in ps}) == 1, "the pieces of a tile stay on one XCD's workers"
            assert min(p[2] - p[1] for p in ps) >= 1
    assert n_split == info["split"] and len(used_chunks) == info["chunks"] and len(used_ctrs) == info["counters"]
    assert info["chunks"] * 256 * 1024 <= ws and info["counters"] <= 4096
    # load balance: no worker runs more than the reported makespan (K steps incl. the model's overheads) and the plan is not absurd
    def weight(m0):   # the planner's model of a ragged row tile: a K step of its staging skeleton costs 0.23 of a full one
        rows = min(256, M - m0)
        return 1.0 if rows > 96 else 0.23 + 0.77 * (-(-rows // 32)) / 4.0
    steps = [sum(float(s[3] - s[2]) * weight(int(s[0])) for s in segs[off[b]: off[b + 1]]) for b in range(workers)]
    assert max(steps) <= info["makespan_x1000"] / 1000.0 + 1e-3
    return segs, off, info, steps


@pytest.mark.parametrize("M,N,K,tile_n", [(1542, 6144, 4096, 256), (1542, 6144, 4096, 192), (1542, 4096, 4096, 256), (1542, 24576, 4096, 256),
                                          (1542, 24576, 4096, 192), (1542, 4096, 12288, 256), (4096, 3456, 1152, 256), (4096, 4304, 1152, 256),
                                          (4096, 1152, 4288, 256), (8192, 8192, 8192, 256), (256, 256, 64, 256), (300, 520, 1024, 256),
                                          (2048, 1024, 1024, 256), (40980, 4096, 4096, 256)])
def test_every_k_tile_of_every_tile_is_covered_exactly_once(M, N, K, tile_n):
    check(M, N, K, tile_n, 256)


@pytest.mark.parametrize("workers", [8, 64, 224, 240, 248, 256, 304])
def test_any_worker_count_that_is_a_multiple_of_eight(workers):
    """The CU reservation knob (tensor-parallel prefill: RCCL's kernels need CUs beside the GEMM): the same shapes planned for fewer
    workgroups are still complete."""
    check(1542, 4096, 4096, 256, workers)
    check(1542, 24576, 4096, 256, workers)
    check(5120, 4096, 12288, 256, workers)


def test_cfg3_plans_balance_the_last_round():
    """BASELINE cfg 3 (M = 1542): what the planner buys, in K steps of the slowest worker against the one-tile-per-block launch.
    gate+up: 672 tiles = 3 rounds x 64 = 192 steps -> two whole rounds + a last round cut in two (<= 165 steps of work per worker);
    down_proj: 2 slices x 96 = 96 steps + a reduce pass -> (2/5, 2/5, 1/5) pieces, <= 80; o_proj: as before 32 (two equal pieces) or
    better."""
    _, _, info, steps = check(1542, 24576, 4096, 256, 256)
    assert info["split"] > 0 and max(steps) <= 165, (info, max(steps))
    _, _, info, steps = check(1542, 4096, 12288, 256, 256)
    assert max(steps) <= 82, (info, max(steps))      # (a whole ragged tile: 192 x 0.42)
    _, _, info, steps = check(1542, 4096, 4096, 256, 256)
    assert max(steps) <= 32, (info, max(steps))
    # a shape whose tiles fill whole rounds is not cut at all
    _, _, info, steps = check(8192, 8192, 8192, 256, 256)
    assert info["split"] == 0 and max(steps) == 4 * 128


def test_a_small_workspace_limits_the_cuts():
    """Chunks live in the caller's workspace: with room for few of them the planner falls back to fewer / no cuts, never past the end."""
    _, _, big, _ = check(1542, 4096, 12288, 256, 256)
    _, _, small, _ = check(1542, 4096, 12288, 256, 256, ws=100 * 256 * 1024)
    assert small["chunks"] <= 100 < big["chunks"]
    _, _, none, steps = check(1542, 4096, 12288, 256, 256, ws=0)
    assert none["chunks"] == 0 and none["split"] == 0 and max(steps) == 192


def test_bad_arguments_are_refused():
    lib = _lib.lib()
    out = (ctypes.c_int32 * 8)()
    assert lib.aha_hip_debug_streamk_plan(512, 512, 100, 256, 256, 1 << 30, out, 1, None, None) < 0     # K not a multiple of 64
    assert lib.aha_hip_debug_streamk_plan(512, 512, 128, 256, 250, 1 << 30, out, 1, None, None) < 0     # workers not a multiple of 8
    assert lib.aha_hip_debug_streamk_plan(512, 512, 128, 128, 256, 1 << 30, out, 1, None, None) < 0     # no 128-column tile here
