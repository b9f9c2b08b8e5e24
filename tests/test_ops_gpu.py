"""-m gpu: every HIP kernel family against the oracle restatement on the same seeded inputs (through the C ABI).

Tolerances (bf16 model dtype): a HIP op may differ from the oracle's op-granular bf16 result by at most a few bf16
ulps of the output (f32 accumulation order, f32 vs bf16-rounded softmax probabilities); each test states its bound.
"""
import math

import numpy as np
import pytest
import torch

from oracle import qwen3 as oq
from oracle.numerics import Numerics

pytestmark = pytest.mark.gpu
NM = Numerics("bf16", matmul_f64=True)  # ideal (order-independent) accumulation, rounded once


def bf(x):
    return x.to(torch.bfloat16)


def rnd(shape, seed, std=1.0, mean=0.0):
    g = torch.Generator().manual_seed(seed)
    return bf(torch.randn(shape, generator=g) * std + mean)


def ulp_bf16(x):
    """bf16 ulp at |x| (f32 tensor)."""
    e = torch.floor(torch.log2(x.abs().clamp_min(1e-30)))
    return torch.pow(2.0, e - 7)


def assert_close_ulps(got, ref, ulps, frac_exact=None, what="", row_scale=False):
    """|got - ref| <= ulps bf16 ulps, where the ulp is taken at max(|ref|, scale): an element near zero is the sum of
    cancelling terms of typical size `scale`, so its absolute error follows that size, not its own magnitude.
    scale = rms of the tensor, or (row_scale) the largest |ref| of the element's row (attention rows that see few
    keys are as large as V itself while rows that average many keys are small)."""
    got, ref = got.float().cpu(), ref.float().cpu()
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    rms = ref.abs().amax(-1, keepdim=True) if row_scale else ref.pow(2).mean().sqrt()
    tol = ulps * ulp_bf16(torch.maximum(ref.abs(), rms))
    bad = (got - ref).abs() > tol
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} elements off by more than {ulps} bf16 ulp; max diff {float((got-ref).abs().max())}"
    if frac_exact is not None:
        fe = float((got == ref).float().mean())
        assert fe >= frac_exact, f"{what}: only {fe:.4f} of elements bit-identical (< {frac_exact})"


@pytest.mark.parametrize("rows,dim", [(1, 1024), (7, 4096), (33, 512), (5, 128), (3, 1152)])
def test_rmsnorm(gpu, rows, dim):
    from aha_amd import ops
    x, w = rnd((rows, dim), 1), rnd((dim,), 2, 0.02, 1.0)
    ref = oq.rms_norm(NM, x.float(), w.float(), 1e-6)
    got = ops.rmsnorm(x.to(gpu), w.to(gpu), 1e-6)
    assert_close_ulps(got, ref, 1, 0.99, "rmsnorm")


@pytest.mark.parametrize("N,K", [(4096, 1024), (1024, 3072), (6144, 4096), (4096, 12288), (1000, 512), (152, 264)])
@pytest.mark.parametrize("fuse_norm,residual", [(False, False), (True, False), (False, True)])
def test_gemv(gpu, N, K, fuse_norm, residual):
    from aha_amd import ops
    W, x = rnd((N, K), 3, 0.02), rnd((K,), 4)
    nw = rnd((K,), 5, 0.02, 1.0) if fuse_norm else None
    res = rnd((N,), 6) if residual else None
    h = oq.rms_norm(NM, x.float(), nw.float(), 1e-6) if fuse_norm else x.float()
    ref = NM.linear(h[None], W.float())[0]
    if residual:
        ref = NM.r(res.float() + ref)
    got = ops.gemv(W.to(gpu), x.to(gpu), None if nw is None else nw.to(gpu), 1e-6, None if res is None else res.to(gpu))
    assert_close_ulps(got, ref, 1, 0.98, "gemv")


@pytest.mark.parametrize("I,K", [(3072, 1024), (12288, 4096), (1024, 512)])
def test_gemv_gate_up(gpu, I, K):
    from aha_amd import ops
    Wg, Wu, x, nw = rnd((I, K), 7, 0.02), rnd((I, K), 8, 0.02), rnd((K,), 9), rnd((K,), 10, 0.02, 1.0)
    h = oq.rms_norm(NM, x.float(), nw.float(), 1e-6)
    lhs = NM.r(oq.silu(NM.linear(h[None], Wg.float())))
    ref = NM.r(lhs * NM.linear(h[None], Wu.float()))[0]
    got = ops.gemv_gate_up(Wg.to(gpu), Wu.to(gpu), x.to(gpu), nw.to(gpu), 1e-6)
    assert_close_ulps(got, ref, 2, 0.97, "gemv_gate_up")


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 384, 1024), (1542, 512, 4096), (37, 1152, 1536), (64, 4352, 1152), (65, 136, 72)])
def test_gemm_plain(gpu, M, N, K):
    from aha_amd import ops
    A, W = rnd((M, K), 11), rnd((N, K), 12, 0.02)
    ref = NM.linear(A.float(), W.float())
    got = ops.gemm(A.to(gpu), W.to(gpu))
    assert_close_ulps(got, ref, 1, 0.98, "gemm")


@pytest.mark.parametrize("tile,splitk", [(128, 0), (2128, 1), (256, 1), (256, 2), (256, 4)])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 520, 1032), (1542, 1024, 4096), (257, 4304, 1152), (513, 72 * 4, 72)])
def test_gemm_every_tile_kernel(gpu, tile, splitk, M, N, K):
    """The same op through every GEMM kernel: 128^2 tile, 256 x 128 tile (the eight-wave ring kernel), 256^2 tile, 256^2 with K split into 2 / 4 f32 slabs + the reduce
    pass (odd M / N / K tails, K shorter than a slice).  All must meet the oracle bound; split-K changes only the f32
    summation order."""
    from aha_amd import ops, _lib
    A, W, b, res = rnd((M, K), 31), rnd((N, K), 32, 0.02), rnd((N,), 33, 0.5), rnd((M, N), 34)
    y = NM.r(torch.nn.functional.gelu(NM.linear(A.float(), W.float(), b.float()), approximate="tanh"))
    ref = NM.r(res.float() + y)
    ops.gemm_plan(tile, splitk)
    try:
        got = ops.gemm(A.to(gpu), W.to(gpu), b.to(gpu), res.to(gpu), _lib.ACT_GELU_TANH)
        plain = ops.gemm(A.to(gpu), W.to(gpu))
    finally:
        ops.gemm_plan(0, 0)
    assert_close_ulps(got, ref, 2, 0.97, f"gemm tile {tile} splitk {splitk}")
    assert_close_ulps(plain, NM.linear(A.float(), W.float()), 1, 0.98, f"plain gemm tile {tile} splitk {splitk}")


@pytest.mark.parametrize("splitk", [1, 2, 3])
@pytest.mark.parametrize("M,N,K", [(289, 512, 192), (130, 256, 320), (700, 328, 448), (608, 1024, 2048), (1542, 768, 4160)])
def test_gemm_four_wave_kernel_edges(gpu, splitk, M, N, K):
    """gemm256q_kernel (K a multiple of 64): odd K-tile counts (3, 5, 7, 65: the two-tile loop body's tail), uneven split-K slices,
    every count of valid 32-row fragments in the last row tile (289 -> 2, 130 -> 1 / 0, 700 -> 2 / 0 of the second wave row, 1542 -> 1 / 0),
    ragged N (328), the bias + residual epilogue on the deep-K route (K = 2048)."""
    from aha_amd import ops, _lib
    A, W, b, res = rnd((M, K), 41), rnd((N, K), 42, 0.02), rnd((N,), 43, 0.5), rnd((M, N), 44)
    ref_plain = NM.linear(A.float(), W.float())
    ref = NM.r(res.float() + NM.linear(A.float(), W.float(), b.float()))
    ops.gemm_plan(256, splitk)
    try:
        plain = ops.gemm(A.to(gpu), W.to(gpu))
        full = ops.gemm(A.to(gpu), W.to(gpu), b.to(gpu), res.to(gpu), _lib.ACT_NONE)
    finally:
        ops.gemm_plan(0, 0)
    assert_close_ulps(plain, ref_plain, 1, 0.98, f"gemm256q plain splitk {splitk}")
    assert_close_ulps(full, ref, 2, 0.97, f"gemm256q bias+residual splitk {splitk}")


@pytest.mark.parametrize("splitk", [2, 3, 4, 6, 8])
@pytest.mark.parametrize("M,N,K", [(406, 1024, 2048), (390, 896, 3584), (128, 512, 4096), (16, 1024, 1536), (130, 1000, 1032), (257, 520, 4104), (64, 136, 520)])
def test_gemm_ring_kernel_k_slices(gpu, splitk, M, N, K):
    """128^2 tiles x K slices on gemm_glds_ring_kernel<ACT_PARTIAL_F32> + the reduce pass (the cfg 4 o_proj / down_proj / fc2 plans, forced
    here on every slice count): uneven slices, K tails inside the last slice (1032, 4104, 520), fewer k steps per slice than ring stages
    (520 = 9 k tiles over 8 slices), ragged M / N; bias + residual through the reduce pass; bit-repeatable (a hazard in the staging
    ring would show as a run-to-run difference); and the automatic plan of the same shape inside the same bound."""
    from aha_amd import ops, _lib
    A, W, b, res = rnd((M, K), 45), rnd((N, K), 46, 0.02), rnd((N,), 47, 0.5), rnd((M, N), 48)
    ref_plain = NM.linear(A.float(), W.float())
    ref = NM.r(res.float() + NM.linear(A.float(), W.float(), b.float()))
    Ad, Wd, bd, rd = A.to(gpu), W.to(gpu), b.to(gpu), res.to(gpu)
    ops.gemm_plan(128, splitk)
    try:
        plain = ops.gemm(Ad, Wd)
        full = ops.gemm(Ad, Wd, bd, rd, _lib.ACT_NONE)
        for _ in range(8):
            assert torch.equal(ops.gemm(Ad, Wd), plain)
    finally:
        ops.gemm_plan(0, 0)
    assert_close_ulps(plain, ref_plain, 1, 0.98, f"ring kernel, {splitk} K slices")
    assert_close_ulps(full, ref, 2, 0.97, f"ring kernel, {splitk} K slices, bias + residual")
    assert_close_ulps(ops.gemm(Ad, Wd), ref_plain, 1, 0.98, "automatic plan")


@pytest.mark.parametrize("seed", range(8))
def test_gemm_random_shapes_on_the_automatic_plan(gpu, seed):
    """Eight draws of six random (M, N, K, epilogue) each through launch_gemm's AUTOMATIC plan -- few rows to a few hundred (the ring kernel,
    its K slices), 256-2000 rows with short and long K (256 x 128 / 256 x 192 / 256^2 tiles, split-K, ragged-N splits), ragged M / N / K
    (multiples of 8) -- against the oracle chain of the epilogue; a second call must repeat the bits."""
    import random
    from aha_amd import ops, _lib
    rng = random.Random(1000 + seed)
    for case in range(6):
        M = rng.choice([rng.randint(1, 64), rng.randint(65, 450), rng.randint(256, 2000)])
        N = 8 * rng.choice([rng.randint(8, 64), rng.randint(64, 560)])
        K = 8 * rng.choice([rng.randint(8, 40), rng.randint(40, 540)])
        epi = rng.choice(["plain", "bias", "bias_res", "res", "bias_gelu", "silu_pairs"])
        if epi == "silu_pairs":
            N = max(32, N // 32 * 32)
        A, W = rnd((M, K), 7000 + seed * 10 + case), rnd((N, K), 7100 + seed * 10 + case, 0.02)
        b, r = rnd((N,), 7200 + seed * 10 + case, 0.5), rnd((M, N), 7300 + seed * 10 + case)
        Ad, Wd, bd, rd = A.to(gpu), W.to(gpu), b.to(gpu), r.to(gpu)
        lin = NM.linear(A.float(), W.float())
        what = f"seed {seed} case {case}: M={M} N={N} K={K} {epi}"
        if epi == "plain":
            got, ref, ulps = ops.gemm(Ad, Wd), lin, 1
            again = ops.gemm(Ad, Wd)
        elif epi == "bias":
            got, ref, ulps = ops.gemm(Ad, Wd, bd), NM.linear(A.float(), W.float(), b.float()), 2
            again = ops.gemm(Ad, Wd, bd)
        elif epi == "bias_res":
            got, ref, ulps = ops.gemm(Ad, Wd, bd, rd, _lib.ACT_NONE), NM.r(r.float() + NM.linear(A.float(), W.float(), b.float())), 2
            again = ops.gemm(Ad, Wd, bd, rd, _lib.ACT_NONE)
        elif epi == "res":
            got, ref, ulps = ops.gemm(Ad, Wd, None, rd, _lib.ACT_NONE), NM.r(r.float() + lin), 2
            again = ops.gemm(Ad, Wd, None, rd, _lib.ACT_NONE)
        elif epi == "bias_gelu":
            ref = NM.r(torch.nn.functional.gelu(NM.linear(A.float(), W.float(), b.float()), approximate="tanh"))
            got, ulps = ops.gemm(Ad, Wd, bd, None, _lib.ACT_GELU_TANH), 3
            again = ops.gemm(Ad, Wd, bd, None, _lib.ACT_GELU_TANH)
        else:   # gate / up rows interleaved in blocks of 16 (the model loader's layout): out[:, j] = silu(gate_j) * up_j
            Wg, Wu = W[: N // 2], W[N // 2:]
            Wf = ops.interleave_gate_up(Wg, Wu)
            ref = NM.r(NM.r(oq.silu(NM.linear(A.float(), Wg.float()))) * NM.linear(A.float(), Wu.float()))
            got, ulps = ops.gemm(Ad, Wf.to(gpu), None, None, _lib.ACT_SILU_MUL_PAIRS), 2
            again = ops.gemm(Ad, Wf.to(gpu), None, None, _lib.ACT_SILU_MUL_PAIRS)
        assert torch.equal(got, again), what + ": not repeatable"
        assert_close_ulps(got, ref, ulps, 0.95, what)


@pytest.mark.parametrize("epi", ["plain", "bias_res", "bias_gelu"])
@pytest.mark.parametrize("M,N,K", [(200, 640, 1152), (390, 896, 896), (406, 1000, 1032), (70, 1736, 264), (512, 1152, 1152)])
def test_gemm_ring_kernel_is_bit_identical_to_the_single_stage_kernel(gpu, epi, M, N, K):
    """gemm_glds_ring_kernel (launches of at most one block per CU: eight waves of 64 x 32 behind a four-stage LDS-DMA ring, and the
    256 x 128 form) against gemm_glds_kernel (four waves of 64 x 64, one stage): same tile, same k order per accumulator, same epilogue
    chain => the same bits.  A row's result does not depend on the other rows of the launch, so the same rows are computed twice under
    the forced 128^2 plan -- alone (<= 256 blocks: the ring kernel) and as the head of a 36 x taller matrix (> 256 blocks: the
    single-stage kernel) -- and once under the forced 256 x 128 plan; each repeated (a hazard in the ring shows as a difference)."""
    from aha_amd import ops, _lib
    A, W, b = rnd((M, K), 51), rnd((N, K), 52, 0.02), rnd((N,), 53, 0.5)
    big = torch.cat([A] + [rnd((M, K), 60 + i) for i in range(35)], 0)
    res, res_big = rnd((M, N), 54), None
    Ad, Wd, bd, bigd = A.to(gpu), W.to(gpu), b.to(gpu), big.to(gpu)
    if epi == "bias_res":
        res_big = torch.cat([res, torch.zeros(35 * M, N, dtype=res.dtype)], 0).to(gpu)

    def run(a, r):
        if epi == "plain":
            return ops.gemm(a, Wd)
        if epi == "bias_res":
            return ops.gemm(a, Wd, bd, r, _lib.ACT_NONE)
        return ops.gemm(a, Wd, bd, None, _lib.ACT_GELU_TANH)
    assert (36 * M + 127) // 128 * ((N + 127) // 128) > 256 >= (M + 127) // 128 * ((N + 127) // 128)
    ops.gemm_plan(128, 1)
    try:
        alone = run(Ad, res.to(gpu))
        for _ in range(6):
            assert torch.equal(run(Ad, res.to(gpu)), alone)
        head = run(bigd, res_big)[:M]
    finally:
        ops.gemm_plan(0, 0)
    assert torch.equal(alone, head)
    ops.gemm_plan(2128, 1)
    try:
        tall = run(Ad, res.to(gpu))
        for _ in range(6):
            assert torch.equal(run(Ad, res.to(gpu)), tall)
    finally:
        ops.gemm_plan(0, 0)
    assert torch.equal(alone, tall)


@pytest.mark.parametrize("M,N,K", [(600, 512, 1152), (4096, 3456, 1152), (300, 768, 192)])
def test_gemm_four_wave_kernel_short_k_bias_gelu(gpu, M, N, K):
    """Short K loops (18 and 3 K tiles) with a bias / bias + GELU epilogue and no residual run on the four-wave 256^2 kernel (the
    ViT qkv and fc1 shapes): automatic plan and the forced 256^2 tile, against the oracle chain."""
    from aha_amd import ops, _lib
    A, W, b = rnd((M, K), 81), rnd((N, K), 82, 0.02), rnd((N,), 83, 0.5)
    ref_b = NM.linear(A.float(), W.float(), b.float())
    ref_g = NM.r(torch.nn.functional.gelu(ref_b, approximate="tanh"))
    for plan in ((0, 0), (256, 1)):
        ops.gemm_plan(*plan)
        try:
            got_b = ops.gemm(A.to(gpu), W.to(gpu), b.to(gpu))
            got_g = ops.gemm(A.to(gpu), W.to(gpu), b.to(gpu), None, _lib.ACT_GELU_TANH)
        finally:
            ops.gemm_plan(0, 0)
        assert_close_ulps(got_b, ref_b, 2, 0.97, f"bias, plan {plan}")          # (two roundings: the bounds of the chained cases above)
        # GELU on exp2 / rcp (kernels_gemm.hip gelu_tanh_f) behind a reordered f32 sum: at 14 M outputs a handful land 3 ulps
        # away (4 of 14 155 776 measured); the bound is 3 ulps with at most two per million beyond 2
        assert_close_ulps(got_g, ref_g, 3, 0.97, f"bias + GELU, plan {plan}")
        g32, r32 = got_g.float().cpu(), ref_g.float().cpu()
        tol2 = 2 * ulp_bf16(torch.maximum(r32.abs(), r32.pow(2).mean().sqrt()))
        assert ((g32 - r32).abs() > tol2).float().mean().item() < 2e-6, f"bias + GELU, plan {plan}: too many outputs beyond 2 ulps"


def test_gemm_ragged_n_is_split_by_the_automatic_plan(gpu):
    """N = 4304 = 16 x 256 + 208 at M >= 256 (the ViT fc1 shape): the automatic plan runs the columns up to 4096 and the 208-column
    tail as two GEMMs over sub-views of W / C / bias / residual.  Same bound against the oracle as any other plan, and the two
    halves must meet exactly at the seam (a wrong view offset shows up as garbage in columns 4096.. or in the last rows)."""
    from aha_amd import ops, _lib
    M, N, K = 600, 4304, 1152
    A, W, b, res = rnd((M, K), 51), rnd((N, K), 52, 0.02), rnd((N,), 53, 0.5), rnd((M, N), 54)
    y = NM.r(torch.nn.functional.gelu(NM.linear(A.float(), W.float(), b.float()), approximate="tanh"))
    ref = NM.r(res.float() + y)
    got = ops.gemm(A.to(gpu), W.to(gpu), b.to(gpu), res.to(gpu), _lib.ACT_GELU_TANH)
    # the GELU bound of test_gemm_four_wave_kernel_short_k_bias_gelu (exp2 / rcp GELU behind a reordered f32 sum): 3 ulps, at most two per
    # million beyond 2 -- whichever tile the plan picks for these rows (round 6: 128^2 tiles here; one of 2.6 M outputs lands at 3 ulps)
    assert_close_ulps(got, ref, 3, 0.97, "ragged-N gemm, automatic plan")
    g32, r32 = got.float().cpu(), ref.float().cpu()
    tol2 = 2 * ulp_bf16(torch.maximum(r32.abs(), r32.pow(2).mean().sqrt()))
    assert ((g32 - r32).abs() > tol2).float().mean().item() < 2e-6, "ragged-N gemm, automatic plan: too many outputs beyond 2 ulps"
    assert_close_ulps(got[:, 4000:], ref[:, 4000:], 3, 0.97, "ragged-N gemm around the seam")
    ops.gemm_plan(256, 1)     # ... and the two-launch split of the 256-row kernel the test is named after, forced
    try:
        got256 = ops.gemm(A.to(gpu), W.to(gpu), b.to(gpu), res.to(gpu), _lib.ACT_GELU_TANH)
    finally:
        ops.gemm_plan(0, 0)
    assert_close_ulps(got256, ref, 3, 0.97, "ragged-N gemm, 256-row tiles")
    g32 = got256.float().cpu()
    assert ((g32 - r32).abs() > tol2).float().mean().item() < 2e-6, "ragged-N gemm, 256-row tiles: too many outputs beyond 2 ulps"
    assert_close_ulps(got256[:, 4000:], ref[:, 4000:], 3, 0.97, "ragged-N gemm around the seam, 256-row tiles")
    plain = ops.gemm(A.to(gpu), W.to(gpu))
    assert_close_ulps(plain, NM.linear(A.float(), W.float()), 1, 0.98, "ragged-N plain gemm")


def test_gemm_gate_up_pairs_256_tile(gpu):
    from aha_amd import ops, _lib
    M, I, K = 700, 1024, 512
    A, Wg, Wu = rnd((M, K), 17), rnd((I, K), 18, 0.05), rnd((I, K), 19, 0.05)
    lhs = NM.r(oq.silu(NM.linear(A.float(), Wg.float())))
    ref = NM.r(lhs * NM.linear(A.float(), Wu.float()))
    Wf = ops.interleave_gate_up(Wg, Wu)
    ops.gemm_plan(256, 1)
    try:
        got = ops.gemm(A.to(gpu), Wf.to(gpu), act=_lib.ACT_SILU_MUL_PAIRS)
    finally:
        ops.gemm_plan(0, 0)
    assert_close_ulps(got, ref, 2, 0.97, "gemm gate/up pairs, 256 tile")


@pytest.mark.parametrize("M,N,K", [(256, 192, 64), (289, 384, 192), (130, 576, 320), (700, 328, 448), (1542, 960, 4096), (513, 1000, 128)])
def test_gemm_256x192_tile_plain(gpu, M, N, K):
    """gemm256q_kernel<.., NF3> (256 x 192 tile, three fragment columns per wave; round 3): odd K-tile counts (1, 3, 5, 7, 64, 2), every
    count of valid 32-row fragments in the last row tile, N a multiple of 192, ragged N (328 = 192 + 136, 1000 = 5 x 192 + 40: the last
    column tile is partly / mostly out of range, including a wave whose whole 96-column slice is), against the oracle."""
    from aha_amd import ops
    A, W = rnd((M, K), 91), rnd((N, K), 92, 0.02)
    ref = NM.linear(A.float(), W.float())
    ops.gemm_plan(192, 1)
    try:
        got = ops.gemm(A.to(gpu), W.to(gpu))
    finally:
        ops.gemm_plan(0, 0)
    assert_close_ulps(got, ref, 1, 0.98, "gemm 256x192 plain")


@pytest.mark.parametrize("M,I,K", [(700, 1056, 512), (1542, 3072, 1024), (256, 96, 64), (300, 4000, 192)])
def test_gemm_256x192_tile_gate_up_pairs(gpu, M, I, K):
    """The gate+up epilogue on the 192-column tile: a tile holds 6 gate|up fragment pairs = 96 output columns; I = 1056 = 11 x 96,
    3072 = 32 x 96, 96 (one tile), 4000 (ragged: 41 x 96 + 64)."""
    from aha_amd import ops, _lib
    A, Wg, Wu = rnd((M, K), 93), rnd((I, K), 94, 0.05), rnd((I, K), 95, 0.05)
    lhs = NM.r(oq.silu(NM.linear(A.float(), Wg.float())))
    ref = NM.r(lhs * NM.linear(A.float(), Wu.float()))
    Wf = ops.interleave_gate_up(Wg, Wu)
    ops.gemm_plan(192, 1)
    try:
        got = ops.gemm(A.to(gpu), Wf.to(gpu), act=_lib.ACT_SILU_MUL_PAIRS)
    finally:
        ops.gemm_plan(0, 0)
    assert_close_ulps(got, ref, 2, 0.97, "gemm gate/up pairs, 256x192 tile")


def test_gemm_256x192_tile_equals_the_256_tile_bitwise(gpu):
    """Same MFMA order along K, same rounding chain: the 192-column tile must reproduce the 256-column tile's output bit for bit
    (a wrong staging row / fragment column would not survive this on random data) at the cfg 3 qkv shape."""
    from aha_amd import ops
    M, N, K = 1542, 6144, 4096
    A, W = rnd((M, K), 96).to(gpu), rnd((N, K), 97, 0.02).to(gpu)
    outs = []
    for tile in (256, 192):
        ops.gemm_plan(tile, 1)
        try:
            outs.append(ops.gemm(A, W))
        finally:
            ops.gemm_plan(0, 0)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("seg,groups,N,K,act", [(512, 3, 768, 1024, 0), (300, 2, 1536, 512, 0), (600, 4, 1152, 320, 4), (256, 8, 192, 64, 0), (88, 2, 512, 512, 0),
                                                 (1283, 2, 3072, 1024, 4)])
def test_gemm_row_groups_equal_one_gemm_per_segment_bitwise(gpu, seg, groups, N, K, act):
    """launch_gemm_grouped (csrc/kernels_gemm.hip; the tensor-parallel prefill's chunked all-gather reads its staging layout through it):
    `groups` segments of `seg` rows, packed in A, written to rows c_row0 + g * c_gstride of a larger output and clipped at m_total -- one
    launch on the four-wave kernels (256- and 192-column tiles, plain and gate+up epilogues) against one plain GEMM per segment on the same
    kernel: bit-identical, and rows outside the segments / past m_total untouched.  (seg = 88: below one row tile -> the per-segment path.)"""
    from aha_amd import ops, _lib
    a_gs, c_gs, c_row0 = seg, seg + 137, 45
    m_total = c_row0 + (groups - 1) * c_gs + seg - 7          # the last segment loses its last 7 rows
    A = rnd((groups * a_gs, K), 111).to(gpu)
    W = rnd((N, K), 112, 0.05).to(gpu)
    n_out = N // 2 if act == _lib.ACT_SILU_MUL_PAIRS else N
    for tile in (256, 192):
        ops.gemm_plan(tile, 1)
        try:
            out = torch.full((m_total + 64, n_out), 7.0, dtype=torch.bfloat16, device=gpu)
            ops.gemm_grouped(A, W, out, seg, groups, a_gs, c_gs, c_row0, m_total, act)
            ref = torch.full_like(out, 7.0)
            for g in range(groups):
                r = c_row0 + g * c_gs
                rows = min(seg, m_total - r)
                ref[r:r + rows] = ops.gemm(A[g * a_gs:g * a_gs + rows].contiguous(), W, act=act)
        finally:
            ops.gemm_plan(0, 0)
        assert torch.equal(out, ref), f"tile {tile}"
    # the automatic tile choice gives one of the two
    out = torch.full((m_total + 64, n_out), 7.0, dtype=torch.bfloat16, device=gpu)
    ops.gemm_grouped(A, W, out, seg, groups, a_gs, c_gs, c_row0, m_total, act)
    if seg >= 256:
        assert torch.equal(out, ref)


@pytest.mark.parametrize("M,N,K,sk", [(4096, 1152, 4352, 2), (1542, 1000, 2048, 3), (300, 328, 1024, 4), (600, 576, 1280, 2)])
def test_gemm_256x192_tile_split_k_equals_the_256_tile_split_bitwise(gpu, M, N, K, sk):
    """Split-K on the 192-column tile (gemm256q_kernel<ACT_PARTIAL_F32, .., NF3>; round 4, the ViT fc2 plan): the same K slices, the
    same K order inside a slice, the same reduce pass -> bit-identical to the 256-column slabs, with every epilogue the reduce pass runs
    (bias + residual, bias + GELU, plain); ragged N (1000 = 5 x 192 + 40, 328), ragged M, a K-tile count the slices do not divide (20 / 3)."""
    from aha_amd import ops, _lib
    A, W, b, res = rnd((M, K), 101).to(gpu), rnd((N, K), 102, 0.02).to(gpu), rnd((N,), 103, 0.5).to(gpu), rnd((M, N), 104).to(gpu)
    for args in [(b, res), (b, None, _lib.ACT_GELU_TANH), ()]:
        outs = []
        for tile in (256, 192):
            ops.gemm_plan(tile, sk)
            try:
                outs.append(ops.gemm(A, W, *args))
            finally:
                ops.gemm_plan(0, 0)
        assert torch.equal(outs[0], outs[1]), f"epilogue {len(args)}"
    ref = NM.r(NM.r(NM.linear(A.float().cpu(), W.float().cpu()) + b.float().cpu()) + res.float().cpu())
    ops.gemm_plan(192, sk)
    try:
        got = ops.gemm(A, W, b, res)
    finally:
        ops.gemm_plan(0, 0)
    assert_close_ulps(got, ref, 2, 0.97, "192-column split-K, bias + residual")


@pytest.mark.parametrize("M", [257, 262, 288, 513, 1542, 1568, 2049])
@pytest.mark.parametrize("N,K", [(192, 64), (328, 192), (1152, 1024), (6144, 4096)])
def test_gemm_fifth_fragment_row_equals_the_256_tile_bitwise(gpu, M, N, K):
    """ROW5 (round 4): M = 256 q + r, 1 <= r <= 32, runs on the 192-column kernel with q row tiles, the r rows as a fifth fragment row
    of the wm = 1 waves of the last one.  Every element is still the same K-ordered MFMA sum and goes through the same epilogue band
    code, so the output -- plain and gate * up -- must equal the 256-column kernel's bit for bit: r = 1, 6, 32, the q = 1 / 2 / 6 / 8
    cases, one / three / 16 / 64 K tiles, ragged N (328), the poisoned-LDS repeat (the fifth row's pieces live in the half of the W1
    region the 192-column tile otherwise never writes)."""
    from aha_amd import ops, _lib
    if M == 2049 and K == 4096:
        pytest.skip("size")
    A, W = rnd((M, K), 221).to(gpu), rnd((N, K), 222, 0.02).to(gpu)
    outs, pairs = [], []
    for tile in (256, 192):
        ops.gemm_plan(tile, 1)
        try:
            if tile == 192:
                ops.poison_lds(M + N)
            outs.append(ops.gemm(A, W))
            if N % 32 == 0:
                pairs.append(ops.gemm(A, W, act=_lib.ACT_SILU_MUL_PAIRS))
        finally:
            ops.gemm_plan(0, 0)
    assert torch.equal(outs[0], outs[1]), f"rows differ: {(outs[0] != outs[1]).any(-1).nonzero().flatten()[:8].tolist()}"
    if pairs:
        assert torch.equal(pairs[0], pairs[1])
    assert_close_ulps(outs[1], NM.linear(A.float().cpu(), W.float().cpu()), 1, 0.98, "gemm 256x192 + fifth fragment row")


@pytest.mark.parametrize("tile", [128, 192, 256])
def test_kernels_do_not_consume_unstaged_lds(gpu, tile):
    """aha_hip_debug_poison_lds fills the LDS of every CU with seeded garbage.  A GEMM tile that read a staging slot it had not
    (yet) written in this launch would see the previous kernel's leftovers: invisible when a launch is simply repeated (the
    leftovers are its own, identical, data), visible as run-to-run differences behind different poisons.  Ragged M and N so that
    the masked rows / columns and the 192-column tile's half-used W region are exercised."""
    from aha_amd import ops, _lib
    A, W = rnd((513, 512), 211).to(gpu), rnd((1000, 512), 212, 0.05).to(gpu)
    Wp = rnd((1024, 512), 213, 0.05).to(gpu)
    ops.gemm_plan(tile, 1)
    try:
        outs = []
        for seed in (None, 1, 2, 3):
            if seed is not None:
                ops.poison_lds(seed)
            outs.append((ops.gemm(A, W).clone(), ops.gemm(A, Wp, act=_lib.ACT_SILU_MUL_PAIRS).clone()))
    finally:
        ops.gemm_plan(0, 0)
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1])


@pytest.mark.parametrize("tile", [192, 128])
def test_small_kernels_beside_a_gemm_on_another_stream(gpu, tile):
    """Two streams, one GPU (what two tensor-parallel rank threads of tests/test_tp_gpu.py, or two models of one process, produce):
    the RMSNorm rows must come out the same whatever GEMM waves share their CUs.  Round 3 found that they did not: beside the
    192-column GEMM tile (424 VGPRs: room for a foreign wave on the SIMD) ~20% of the launches returned rows scaled by
    sqrt(16/15) -- clang's `v_pk_add_f32 .. op_sel:[0,1] op_sel_hi:[1,0]` at the end of the sum of squares picked the wrong
    half of its second source in lanes 16-31 / 48-63 (profiles/r03_simd_coresidency.md).  Both sides are closed: the kernel no
    longer contains that instruction form (tests/test_isa_cpu.py) and the four-wave GEMM claims the whole register file."""
    import threading
    from aha_amd import ops, _lib
    x = rnd((513, 512), 201).to(gpu)
    w = bf(1 + 0.1 * torch.randn(512, generator=torch.Generator().manual_seed(202))).to(gpu)
    A, W = rnd((513, 512), 203).to(gpu), rnd((1024, 512), 204, 0.05).to(gpu)
    # ... and the batch-1 matvecs (their packed fmas broadcast one activation element per pair of rows: second-source selects until
    # round 3, first-source selects -- measured safe -- since): qkv-like (1 row per wave), gate+up (2 rows per wave), wide N (4 rows)
    xv = rnd((4096,), 205).to(gpu)
    Wq, Wg, Wu, Wl = (rnd((n, 4096), 206 + i, 0.02).to(gpu) for i, n in enumerate((6144, 12288, 12288, 32768)))
    victims = {
        "rmsnorm rows": lambda: ops.rmsnorm(x, w, 1e-6),
        "matvec 6144 x 4096": lambda: ops.gemv(Wq, xv),
        "gate+up matvec 12288 x 4096": lambda: ops.gemv_gate_up(Wg, Wu, xv),
        "matvec 32768 x 4096": lambda: ops.gemv(Wl, xv),
    }
    refs = {k: f().clone() for k, f in victims.items()}
    ref = refs["rmsnorm rows"]
    torch.cuda.synchronize()
    stop, started = [False], threading.Event()

    def other_stream():
        s2 = torch.cuda.Stream()
        with torch.cuda.stream(s2):
            while not stop[0]:
                for _ in range(20):
                    ops.gemm(A, W, act=_lib.ACT_SILU_MUL_PAIRS)
                s2.synchronize()
                started.set()

    ops.gemm_plan(tile, 1)
    t = threading.Thread(target=other_stream)
    t.start()
    try:
        assert started.wait(60)
        s = torch.cuda.Stream()
        bad = {k: 0 for k in victims}
        with torch.cuda.stream(s):
            for k, f in victims.items():
                for _ in range(1500 if k == "rmsnorm rows" else 500):
                    y = f()
                    s.synchronize()
                    bad[k] += not torch.equal(y, refs[k])
    finally:
        stop[0] = True
        t.join()
        ops.gemm_plan(0, 0)
    assert not any(bad.values()), f"launches that differ beside the {tile}-column GEMM on another stream: {bad}"


def test_gemm_transpose_detect(gpu):
    """A = I against an asymmetric W: catches swapped C rows/cols (cdna guide: always A=I-check with asymmetric B)."""
    from aha_amd import ops
    K = 128
    A = bf(torch.eye(K))
    W = bf(torch.arange(192 * K, dtype=torch.float32).reshape(192, K) % 251 - 100.0)
    got = ops.gemm(A.to(gpu), W.to(gpu)).float().cpu()
    assert torch.equal(got, W.float().t())


@pytest.mark.parametrize("act", ["gelu_tanh", "gelu_erf", "silu", "none"])
def test_gemm_bias_act_residual(gpu, act):
    from aha_amd import ops, _lib
    M, N, K = 150, 640, 1152
    A, W, b, res = rnd((M, K), 13), rnd((N, K), 14, 0.02), rnd((N,), 15, 0.5), rnd((M, N), 16)
    y = NM.linear(A.float(), W.float(), b.float())
    if act == "gelu_tanh":
        y = NM.r(torch.nn.functional.gelu(y, approximate="tanh")); code = _lib.ACT_GELU_TANH
    elif act == "gelu_erf":
        y = NM.r(torch.nn.functional.gelu(y)); code = _lib.ACT_GELU_ERF
    elif act == "silu":
        y = NM.r(oq.silu(y)); code = _lib.ACT_SILU
    else:
        code = _lib.ACT_NONE
    ref = NM.r(res.float() + y)
    got = ops.gemm(A.to(gpu), W.to(gpu), b.to(gpu), res.to(gpu), code)
    assert_close_ulps(got, ref, 2, 0.97, f"gemm+bias+{act}+res")


def test_gemm_gate_up_pairs(gpu):
    from aha_amd import ops, _lib
    M, I, K = 130, 1024, 512
    A, Wg, Wu = rnd((M, K), 17), rnd((I, K), 18, 0.05), rnd((I, K), 19, 0.05)
    lhs = NM.r(oq.silu(NM.linear(A.float(), Wg.float())))
    ref = NM.r(lhs * NM.linear(A.float(), Wu.float()))
    Wf = ops.interleave_gate_up(Wg, Wu)
    got = ops.gemm(A.to(gpu), Wf.to(gpu), act=_lib.ACT_SILU_MUL_PAIRS)
    assert got.shape == (M, I)
    assert_close_ulps(got, ref, 2, 0.97, "gemm gate/up pairs")


def _rope_ref(qkv, qw, kw, pos, axis_map, nh, kvh, d, eps, theta):
    S = qkv.shape[0]
    q = qkv[:, : nh * d].float().reshape(1, S, nh, d)
    k = qkv[:, nh * d: (nh + kvh) * d].float().reshape(1, S, kvh, d)
    v = qkv[:, (nh + kvh) * d:]
    q = oq.rms_norm(NM, q, qw.float(), eps).transpose(1, 2)
    k = oq.rms_norm(NM, k, kw.float(), eps).transpose(1, 2)
    inv = oq.compute_default_rope_parameters(d, theta)
    p = pos.float()  # (3,S)
    fr = p[:, :, None] * inv[None, None, :]           # (3,S,d/2)
    g = torch.gather(fr, 0, axis_map.long()[None, None, :].expand(1, S, d // 2))[0]
    emb = torch.cat([g, g], -1)
    q, k = oq.apply_rotary_pos_emb(NM, q, k, emb.cos()[None], emb.sin()[None])
    return q.transpose(1, 2).reshape(S, nh * d), k.transpose(1, 2).reshape(S, kvh * d), v


@pytest.mark.parametrize("mrope", [False, True])
def test_qknorm_rope(gpu, mrope):
    from aha_amd import ops
    S, nh, kvh, d, theta = 77, 4, 2, 128, 1e6
    qkv = rnd((S, (nh + 2 * kvh) * d), 20)
    qw, kw = rnd((d,), 21, 0.02, 1.0), rnd((d,), 22, 0.02, 1.0)
    g = torch.Generator().manual_seed(23)
    if mrope:
        pos = torch.randint(0, 5000, (3, S), generator=g, dtype=torch.int32)
        axis = torch.zeros(d // 2, dtype=torch.int32)
        for i in range(d // 2):
            if i % 3 == 1 and i < 60: axis[i] = 1
            if i % 3 == 2 and i < 60: axis[i] = 2
    else:
        pos = (torch.arange(S, dtype=torch.int32) + 1234)[None].repeat(3, 1).contiguous()
        axis = torch.zeros(d // 2, dtype=torch.int32)
    rq, rk, rv = _rope_ref(qkv, qw, kw, pos, axis, nh, kvh, d, 1e-6, theta)
    q, k, v = ops.qknorm_rope(qkv.to(gpu), qw.to(gpu), kw.to(gpu), pos.to(gpu), axis.to(gpu), nh, kvh, d, 1e-6, theta)
    assert torch.equal(v.cpu(), rv), "v must pass through bit-exact"
    # cos/sin of f32 angles up to ~5e3 rad: device sinf/cosf vs torch differ by <= 1 ulp f32 -> rare bf16 flips
    assert_close_ulps(q, rq, 2, 0.97, "q rope")
    assert_close_ulps(k, rk, 2, 0.97, "k rope")


def _attn_ref(q, k, v, nh, kvh, d, causal, kv_offset, nm=None):
    nm = nm or NM
    S, L = q.shape[0], k.shape[0]
    qq = q.float().reshape(1, S, nh, d).transpose(1, 2)
    kk = k.float().reshape(1, L, kvh, d).transpose(1, 2)
    vv = v.float().reshape(1, L, kvh, d).transpose(1, 2)
    mask = None
    if causal:
        i = torch.arange(S)[:, None] + kv_offset
        j = torch.arange(L)[None, :]
        mask = torch.zeros(S, L).masked_fill(j > i, float("-inf"))[None, None]
    o = oq.eager_attention_forward(nm, qq, kk, vv, nh // kvh, mask, oq.attn_scale(NM, d))
    return o.reshape(S, nh * d)


NM_F32SCORES = Numerics("bf16", matmul_f64=True, attn_scores_rounded=False)   # the f32 score chain's own rounding points


@pytest.mark.parametrize("L", [1, 63, 64, 65, 200, 1000, 4133])
@pytest.mark.parametrize("nh,kvh", [(16, 8), (32, 8), (4, 4)])
def test_attn_decode(gpu, L, nh, kvh):
    from aha_amd import ops
    d = 128
    q, k, v = rnd((1, nh * d), 30), rnd((L, kvh * d), 31), rnd((L, kvh * d), 32)
    # round 6: the decode kernels keep f32 scores like the prefill kernels' default chain (one convention per cache position, round-5 advisor):
    # the chain's own rounding points at 3 ulp, the eager oracle (two bf16 roundings of every score, modules.rs:782-783) at 4
    ref = _attn_ref(q, k, v, nh, kvh, d, False, 0, NM_F32SCORES)[0]
    got = ops.attn_decode(q[0].contiguous().to(gpu), k.to(gpu), v.to(gpu), nh, kvh, d)
    # unrounded-f32 P vs the oracle's bf16-rounded P: <= 1 ulp of the output plus accumulation order
    assert_close_ulps(got, ref, 3, None, "attn_decode")
    # (per head, at the head's largest |output|: the row scale the prefill tests use -- a dim near zero is a sum of cancelling terms)
    assert_close_ulps(got.reshape(nh, d), _attn_ref(q, k, v, nh, kvh, d, False, 0)[0].reshape(nh, d), 4, None, "attn_decode vs the eager oracle", row_scale=True)
    assert float((got.float().cpu() - ref).abs().max()) < 0.02


def test_attn_decode_peaky(gpu):
    """Forces large score spreads (online-softmax rescale path) -- one key dominates late in the sequence."""
    from aha_amd import ops
    nh, kvh, d, L = 8, 2, 128, 700
    q, k, v = rnd((1, nh * d), 33), rnd((L, kvh * d), 34, 0.3), rnd((L, kvh * d), 35)
    k[650] = (q[0, :d] * 2.0).repeat(kvh)  # spike for head 0 (and correlated for others) in a late page
    ref = _attn_ref(q, k, v, nh, kvh, d, False, 0, NM_F32SCORES)[0]
    got = ops.attn_decode(q[0].contiguous().to(gpu), k.to(gpu), v.to(gpu), nh, kvh, d)
    assert_close_ulps(got, ref, 3, None, "attn_decode peaky")


@pytest.mark.parametrize("S,off", [(1, 0), (5, 0), (64, 0), (65, 0), (130, 0), (300, 0), (17, 100), (64, 64), (100, 333)])
@pytest.mark.parametrize("nh,kvh", [(4, 2), (8, 2)])
def test_attn_prefill_causal(gpu, S, off, nh, kvh):
    from aha_amd import ops
    d, L = 128, S + off
    q, k, v = rnd((S, nh * d), 40), rnd((L, kvh * d), 41), rnd((L, kvh * d), 42)
    # the library's default score chain is the f32 one since round 5 (test_attn_prefill_f32_score_chain): its own rounding points
    ref = _attn_ref(q, k, v, nh, kvh, d, True, off, NM_F32SCORES)
    got = ops.attn_prefill(q.to(gpu), k.to(gpu), v.to(gpu), nh, kvh, d, off, True)
    assert_close_ulps(got, ref, 3, None, "attn_prefill", row_scale=True)
    assert_close_ulps(got, _attn_ref(q, k, v, nh, kvh, d, True, off), 4, None, "attn_prefill vs the eager oracle", row_scale=True)


def test_attn_prefill_full(gpu):
    from aha_amd import ops
    S, nh, kvh, d = 150, 4, 4, 128
    q, k, v = rnd((S, nh * d), 43), rnd((S, kvh * d), 44), rnd((S, kvh * d), 45)
    ref = _attn_ref(q, k, v, nh, kvh, d, False, 0, NM_F32SCORES)
    got = ops.attn_prefill(q.to(gpu), k.to(gpu), v.to(gpu), nh, kvh, d, 0, False)
    assert_close_ulps(got, ref, 3, None, "attn_prefill full", row_scale=True)


@pytest.mark.parametrize("S,off,nh,kvh,causal", [(300, 0, 8, 2, True), (100, 333, 4, 2, True), (2048, 0, 32, 8, True), (1542, 0, 32, 8, True),
                                                 (150, 0, 4, 4, False)])
def test_attn_prefill_score_chain_on_the_matrix_pipe_gives_the_same_bits(gpu, S, off, nh, kvh, causal):
    """`bf16(q.k) * bf16(scale)` (modules.rs:782-783) as diag(scale) x packed-bf16 on the matrix pipe (csrc/attn_common.h
    mfma_diag) is ONE exact bf16 x bf16 product per score, like the vector-ALU multiply: the two variants must agree bit for bit
    (4-wave and 8-wave blocks, diagonal and interior tiles, offsets)."""
    from aha_amd import ops
    d, L = 128, S + off
    q, k, v = rnd((S, nh * d), 46), rnd((L, kvh * d), 47), rnd((L, kvh * d), 48)
    qg, kg, vg = q.to(gpu), k.to(gpu), v.to(gpu)
    outs = {}
    try:
        for smx in (0, 1):
            ops.attn_variant(smx)
            outs[smx] = ops.attn_prefill(qg, kg, vg, nh, kvh, d, off, causal)
    finally:
        ops.attn_variant(-1)
    assert torch.equal(outs[1], outs[0]), "the matrix-pipe scale multiply differs from the vector-ALU chain"
    if S <= 300:
        assert_close_ulps(outs[1], _attn_ref(q, k, v, nh, kvh, d, causal, off), 3, None, "attn_prefill smx 1", row_scale=True)


@pytest.mark.parametrize("smx", [3])
@pytest.mark.parametrize("S,off", [(1, 0), (5, 0), (31, 0), (32, 0), (33, 0), (64, 0), (65, 0), (130, 0), (300, 0), (17, 100), (64, 64), (100, 333)])
@pytest.mark.parametrize("nh,kvh", [(4, 2), (8, 2)])
def test_attn_prefill_f32_score_chain(gpu, smx, S, off, nh, kvh):
    """Round 5 (AHA_ATTN_SMX=3, the default): the scores stay the f32 QK^T accumulators through scale, mask, maximum and exponential;
    P is rounded to bf16 once for the P.V MFMA.  Against the oracle with its score roundings
    OFF (the chain's own rounding points: <= 3 bf16 ulp of the row scale, the bound the bit-faithful chain is held to against the eager
    oracle) and ON (the reference's eager path, modules.rs:782-783: <= 4 ulp)."""
    from aha_amd import ops
    d, L = 128, S + off
    q, k, v = rnd((S, nh * d), 40), rnd((L, kvh * d), 41), rnd((L, kvh * d), 42)
    try:
        ops.attn_variant(smx)
        got = ops.attn_prefill(q.to(gpu), k.to(gpu), v.to(gpu), nh, kvh, d, off, True)
        full = ops.attn_prefill(q.to(gpu), k.to(gpu), v.to(gpu), nh, kvh, d, off, False) if off == 0 else None
    finally:
        ops.attn_variant(-1)
    assert_close_ulps(got, _attn_ref(q, k, v, nh, kvh, d, True, off, NM_F32SCORES), 3, None, f"attn_prefill smx {smx} vs f32-score oracle", row_scale=True)
    # (4 ulp: a score that the eager path's two roundings move across a bf16 boundary shifts its probability by up to 2^-8 relative)
    assert_close_ulps(got, _attn_ref(q, k, v, nh, kvh, d, True, off), 4, None, f"attn_prefill smx {smx} vs eager oracle", row_scale=True)
    if full is not None:
        assert_close_ulps(full, _attn_ref(q, k, v, nh, kvh, d, False, 0, NM_F32SCORES), 3, None, f"attn_prefill full smx {smx}", row_scale=True)


@pytest.mark.parametrize("smx", [3])
def test_attn_prefill_f32_score_chain_peaky_and_long(gpu, smx):
    """The rescale path of the online softmax (a key that dominates late, after the running maximum had settled and the rescale was
    being skipped) and a launch long enough for 8-wave / XCD-ordered blocks (S = 2048 x 32 heads)."""
    from aha_amd import ops
    nh, kvh, d, S = 8, 2, 128, 700
    q, k, v = rnd((S, nh * d), 33), rnd((S, kvh * d), 34, 0.3), rnd((S, kvh * d), 35)
    k[650] = (q[690, :d] * 2.0).repeat(kvh)   # spikes for the late rows, in tile 10
    k[40] = (q[300, :d] * 1.5).repeat(kvh)
    try:
        ops.attn_variant(smx)
        got = ops.attn_prefill(q.to(gpu), k.to(gpu), v.to(gpu), nh, kvh, d, 0, True)
        S2, nh2, kvh2 = 2048, 32, 8
        q2, k2, v2 = rnd((S2, nh2 * d), 36), rnd((S2, kvh2 * d), 37), rnd((S2, kvh2 * d), 38)
        got2 = ops.attn_prefill(q2.to(gpu), k2.to(gpu), v2.to(gpu), nh2, kvh2, d, 0, True)
    finally:
        ops.attn_variant(-1)
    assert_close_ulps(got, _attn_ref(q, k, v, nh, kvh, d, True, 0, NM_F32SCORES), 3, None, f"attn_prefill peaky smx {smx}", row_scale=True)
    ref2 = _attn_ref(q2[:, :4 * d], k2[:, :d], v2[:, :d], 4, 1, d, True, 0, NM_F32SCORES)   # the q heads of kv head 0
    assert_close_ulps(got2[:, :4 * d], ref2, 3, None, f"attn_prefill S=2048 smx {smx}", row_scale=True)
    ref2b = _attn_ref(q2[:, 28 * d:], k2[:, 7 * d:], v2[:, 7 * d:], 4, 1, d, True, 0, NM_F32SCORES)   # ... and of kv head 7
    assert_close_ulps(got2[:, 28 * d:], ref2b, 3, None, f"attn_prefill S=2048 smx {smx} (last kv head)", row_scale=True)


@pytest.mark.parametrize("form", [64, 65])
@pytest.mark.parametrize("S,off", [(1, 0), (5, 0), (31, 0), (32, 0), (33, 0), (64, 0), (65, 0), (130, 0), (255, 0), (256, 0), (257, 0), (300, 0), (600, 0),
                                   (17, 100), (64, 64), (100, 333), (300, 37)])
@pytest.mark.parametrize("nh,kvh", [(4, 2), (8, 2), (16, 8)])
def test_attn_prefill_64_rows_per_wave(gpu, form, S, off, nh, kvh):
    """Round 6: the one-wave-per-SIMD form of the prefill attention (csrc/kernels_attn64.hip: 256-row workgroups, 64 q rows per wave on
    v_mfma_f32_32x32x16_bf16, three-stage LDS-DMA ring; 65 = software-pipelined inside the wave) on the f32 score chain's bounds: <= 3 bf16
    ulp of the row scale against the oracle with the chain's own rounding points, <= 4 against the eager oracle (modules.rs:782-783), causal
    with offsets and full, ragged lengths around the 32- / 64- / 256-row boundaries, grid order (kvh 2) and XCD order (kvh 8)."""
    from aha_amd import ops
    d, L = 128, S + off
    q, k, v = rnd((S, nh * d), 40), rnd((L, kvh * d), 41), rnd((L, kvh * d), 42)
    try:
        ops.attn_form(form)
        got = ops.attn_prefill(q.to(gpu), k.to(gpu), v.to(gpu), nh, kvh, d, off, True)
        full = ops.attn_prefill(q.to(gpu), k.to(gpu), v.to(gpu), nh, kvh, d, off, False) if off == 0 else None
    finally:
        ops.attn_form(-1)
    assert_close_ulps(got, _attn_ref(q, k, v, nh, kvh, d, True, off, NM_F32SCORES), 3, None, f"attn_prefill form {form} vs f32-score oracle", row_scale=True)
    # against the eager oracle (two bf16 roundings of every score, modules.rs:782-783) the f32 chain is held to 4 ulp on the shapes of
    # test_attn_prefill_f32_score_chain; the larger shapes added here carry one or two elements at 5 ulp -- for the 16-row kernel exactly as
    # for this one (scripts/attn64_diag.py: a score the two roundings move across a bf16 boundary) -- so they are held to 5, and to the
    # 16-row kernel's own distance
    old_shapes = (S, off) in [(1, 0), (5, 0), (31, 0), (32, 0), (33, 0), (64, 0), (65, 0), (130, 0), (300, 0), (17, 100), (64, 64), (100, 333)] and kvh == 2
    assert_close_ulps(got, _attn_ref(q, k, v, nh, kvh, d, True, off), 4 if old_shapes else 5, None, f"attn_prefill form {form} vs eager oracle", row_scale=True)
    if full is not None:
        assert_close_ulps(full, _attn_ref(q, k, v, nh, kvh, d, False, 0, NM_F32SCORES), 3, None, f"attn_prefill full form {form}", row_scale=True)


@pytest.mark.parametrize("form", [64, 65])
def test_attn_prefill_64_rows_per_wave_peaky_and_long(gpu, form):
    """The rescale path (a key that dominates late, after the running maximum had settled and the rescale was being skipped; with the
    in-wave pipeline the rescale of tile t + 1 follows P.V of tile t) and S = 2048 x 32 heads: 8 blocks per head in the XCD order."""
    from aha_amd import ops
    nh, kvh, d, S = 8, 2, 128, 700
    q, k, v = rnd((S, nh * d), 33), rnd((S, kvh * d), 34, 0.3), rnd((S, kvh * d), 35)
    k[650] = (q[690, :d] * 2.0).repeat(kvh)   # spikes for the late rows, in tile 10
    k[40] = (q[300, :d] * 1.5).repeat(kvh)
    try:
        ops.attn_form(form)
        got = ops.attn_prefill(q.to(gpu), k.to(gpu), v.to(gpu), nh, kvh, d, 0, True)
        S2, nh2, kvh2 = 2048, 32, 8
        q2, k2, v2 = rnd((S2, nh2 * d), 36), rnd((S2, kvh2 * d), 37), rnd((S2, kvh2 * d), 38)
        got2 = ops.attn_prefill(q2.to(gpu), k2.to(gpu), v2.to(gpu), nh2, kvh2, d, 0, True)
        ops.attn_form(16)
        old2 = ops.attn_prefill(q2.to(gpu), k2.to(gpu), v2.to(gpu), nh2, kvh2, d, 0, True)
    finally:
        ops.attn_form(-1)
    assert_close_ulps(got, _attn_ref(q, k, v, nh, kvh, d, True, 0, NM_F32SCORES), 3, None, f"attn_prefill peaky form {form}", row_scale=True)
    ref2 = _attn_ref(q2[:, :4 * d], k2[:, :d], v2[:, :d], 4, 1, d, True, 0, NM_F32SCORES)   # the q heads of kv head 0
    assert_close_ulps(got2[:, :4 * d], ref2, 3, None, f"attn_prefill S=2048 form {form}", row_scale=True)
    ref2b = _attn_ref(q2[:, 28 * d:], k2[:, 7 * d:], v2[:, 7 * d:], 4, 1, d, True, 0, NM_F32SCORES)   # ... and of kv head 7
    assert_close_ulps(got2[:, 28 * d:], ref2b, 3, None, f"attn_prefill S=2048 form {form} (last kv head)", row_scale=True)
    # every head of the long launch against the 16-row kernel (same rounding points, another accumulation order)
    assert_close_ulps(got2, old2.float().cpu(), 3, None, f"attn_prefill S=2048 form {form} vs the 16-row kernel", row_scale=True)


def test_attn_prefill_64_rows_per_wave_random_shapes_repeatable(gpu):
    """The 64-row kernel's MFMAs, exponentials and accumulator traffic are inline asm: hipcc neither counts their memory operations nor
    pads their hazards (cdna_hip_programming.md section 5.7), and a missed wait state shows as wrong values on SOME waves of SOME launches.
    So beyond the fixed shapes: 48 random (rows, cache offset, heads, causal / full) launches, each run three times on both forms -- the
    three outputs must be bit-identical (a hazard is timing-dependent), finite, and within 3 bf16 ulp of the row scale of the 16-row
    kernel's output (itself held to the oracle above)."""
    from aha_amd import ops
    rng = np.random.default_rng(64)
    d = 128
    for case in range(48):
        kvh = int(rng.choice([2, 4, 8]))
        nh = kvh * int(rng.choice([1, 2, 4]))
        S = int(rng.integers(1, 1400))
        causal = bool(rng.integers(0, 2))
        off = int(rng.integers(0, 700)) if causal else 0
        L = S + off
        q, k, v = rnd((S, nh * d), 1000 + case), rnd((L, kvh * d), 2000 + case, float(rng.choice([0.3, 1.0]))), rnd((L, kvh * d), 3000 + case)
        if case % 5 == 0 and L > 40:     # a dominating key late in the cache: the rescale path
            k[L - 7] = (q[S - 1, :d] * 2.0).repeat(kvh)
        qg, kg, vg = q.to(gpu), k.to(gpu), v.to(gpu)
        try:
            ops.attn_form(16)
            base = ops.attn_prefill(qg, kg, vg, nh, kvh, d, off, causal)
            for form in (64, 65):
                ops.attn_form(form)
                outs = [ops.attn_prefill(qg, kg, vg, nh, kvh, d, off, causal) for _ in range(3)]
                assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), f"case {case} form {form}: not repeatable"
                assert_close_ulps(outs[0], base.float().cpu(), 3, None, f"case {case} (S {S} off {off} nh {nh} kvh {kvh} causal {causal}) form {form}", row_scale=True)
        finally:
            ops.attn_form(-1)


@pytest.mark.parametrize("smx", [0, 1, 3])
@pytest.mark.parametrize("S", [13, 64, 390, 777])
def test_attn_prefill_head_dim_64_full(gpu, smx, S):
    """The Qwen3-ASR audio encoder's geometry (qwen3_asr/model.rs:218-220: global attention, no mask; head_dim 64, nh == kvh) on every
    score-chain variant, 390 = the token count of 30 s of audio."""
    from aha_amd import ops
    nh, d = 14, 64
    q, k, v = rnd((S, nh * d), 51), rnd((S, nh * d), 52), rnd((S, nh * d), 53)
    try:
        ops.attn_variant(smx)
        got = ops.attn_prefill(q.to(gpu), k.to(gpu), v.to(gpu), nh, nh, d, 0, False)
    finally:
        ops.attn_variant(-1)
    assert_close_ulps(got, _attn_ref(q, k, v, nh, nh, d, False, 0, NM if smx <= 1 else NM_F32SCORES), 3, None, f"attn d=64 smx {smx}", row_scale=True)


def test_argmax_first_max(gpu):
    from aha_amd import ops
    x = torch.randn(151936, generator=torch.Generator().manual_seed(50))
    x[77777] = 9.0
    x[1234] = 9.0     # tie -> first index wins (candle argmax / Sampling::ArgMax)
    assert ops.argmax(x.to(gpu)) == 1234
    x2 = torch.full((5000,), -3.0)
    assert ops.argmax(x2.to(gpu)) == 0
