"""-m gpu: the persistent, segment-table-driven GEMM kernel (csrc/kernels_gemm_sk.hip gemm256s_kernel) against the one-tile-per-block
kernels and the oracle's Linear (/root/reference/src/models/common/modules.rs:81-85,577 -- Linear -> bf16, SiLU(gate) * up, + residual;
rounding points restated in oracle/numerics.py).

What must hold:
  * whole tiles (no cut): every output element is the same K-ordered f32 sum as in gemm256q_kernel -> BIT-identical;
  * tiles cut in two equal pieces == the f32-slab split-K plan with two slices (same partial sums, added in the same order, same
    epilogue chain) -> BIT-identical as well;
  * any other cut changes only the f32 summation order: within the oracle bound of the other GEMM kernels, and run-to-run identical
    (the last arriver adds the chunks in K order whoever it is; counters return to zero);
  * ragged M / N, every epilogue that has an instantiation (plain, bias, residual, bias + residual, bias + GELU, gate * up), both tile
    widths, worker counts below the CU count (the CU reservation knob)."""
import numpy as np
import pytest
import torch

from oracle import qwen3 as oq
from oracle.numerics import Numerics
from tests.test_ops_gpu import assert_close_ulps, rnd

pytestmark = pytest.mark.gpu
NM = Numerics("bf16")


def run(plan, fn):
    from aha_amd import ops
    ops.gemm_plan(*plan)
    try:
        return fn()
    finally:
        ops.gemm_plan(0, 0)


@pytest.mark.parametrize("M,N,K", [(1542, 1024, 4096), (512, 2048, 1024), (289, 512, 192), (4096, 1152, 1152), (700, 328, 448)])
def test_whole_tiles_equal_the_one_tile_per_block_kernel_bitwise(gpu, M, N, K):
    from aha_amd import ops, _lib
    A, W, b, res = rnd((M, K), 51), rnd((N, K), 52, 0.02), rnd((N,), 53, 0.5), rnd((M, N), 54)
    Ag, Wg, bg, rg = A.to(gpu), W.to(gpu), b.to(gpu), res.to(gpu)
    for args in [(), (None, rg), (bg,), (bg, rg), (bg, None, _lib.ACT_GELU_TANH)]:
        ref = run((256, 1), lambda: ops.gemm(Ag, Wg, *args))
        got = run((1256, 1), lambda: ops.gemm(Ag, Wg, *args))          # persistent kernel, last round NOT cut
        assert torch.equal(got, ref), f"persistent kernel (whole tiles) != gemm256q_kernel for epilogue {len(args)}"
    # (the oracle bounds of these epilogues are tests/test_ops_gpu.py's, on the one-tile-per-block kernels this one equals bit for bit)
    plain = run((1256, 1), lambda: ops.gemm(Ag, Wg))
    assert_close_ulps(plain, NM.linear(A.float(), W.float()), 1, 0.98, "persistent kernel, plain")


@pytest.mark.parametrize("M,N,K", [(1542, 1024, 4096), (600, 768, 2048), (300, 520, 1024)])
def test_two_equal_pieces_equal_the_two_slab_split_k_plan_bitwise(gpu, M, N, K):
    from aha_amd import ops
    A, W, res = rnd((M, K), 55), rnd((N, K), 56, 0.02), rnd((M, N), 57)
    Ag, Wg, rg = A.to(gpu), W.to(gpu), res.to(gpu)
    ref = run((256, 2), lambda: ops.gemm(Ag, Wg, None, rg))
    got = run((1256, 2), lambda: ops.gemm(Ag, Wg, None, rg))
    assert torch.equal(got, ref)
    assert_close_ulps(got, NM.r(res.float() + NM.linear(A.float(), W.float())), 2, 0.97, "residual, two pieces")


@pytest.mark.parametrize("cut", [2, 3, 4, 11, 12, 13])
@pytest.mark.parametrize("M,N,K,tile", [(1542, 768, 4160, 1256), (520, 1536, 2048, 1256), (1542, 1152, 4096, 1192), (130, 256, 1280, 1256)])
def test_every_cut_style_meets_the_oracle_bound_and_is_deterministic(gpu, cut, M, N, K, tile):
    """cut: style * 10 + cuts (style 0 = equal pieces, 1 = `cuts` big pieces + one remainder).  Odd K-tile counts (65), ragged M (1542,
    130, 520), ragged N for the 192-column tile (1152 = 6 x 192) and the 256 one (768 = 3 x 256)."""
    from aha_amd import ops
    A, W = rnd((M, K), 58), rnd((N, K), 59, 0.02)
    Ag, Wg = A.to(gpu), W.to(gpu)
    outs = [run((tile, cut), lambda: ops.gemm(Ag, Wg)) for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "the sum order must not depend on which piece arrives last"
    assert_close_ulps(outs[0], NM.linear(A.float(), W.float()), 1, 0.98, f"persistent kernel cut {cut}")


@pytest.mark.parametrize("tile", [1192])
def test_gate_up_pairs_on_the_persistent_kernel(gpu, tile):
    """SiLU(gate) * up epilogue (modules.rs:81-85) behind a cut: the finisher runs it on the summed tile.  (192-column tiles: the only
    persistent instantiation of this epilogue.)"""
    from aha_amd import ops, _lib
    M, I, K = 1542, 1536, 1024
    A, Wg_, Wu = rnd((M, K), 60), rnd((I, K), 61, 0.05), rnd((I, K), 62, 0.05)
    ref = NM.r(NM.r(oq.silu(NM.linear(A.float(), Wg_.float()))) * NM.linear(A.float(), Wu.float()))
    Wf = ops.interleave_gate_up(Wg_, Wu).to(gpu)
    Ag = A.to(gpu)
    whole = run((tile, 1), lambda: ops.gemm(Ag, Wf, act=_lib.ACT_SILU_MUL_PAIRS))
    base = run((tile - 1000, 1), lambda: ops.gemm(Ag, Wf, act=_lib.ACT_SILU_MUL_PAIRS))
    assert torch.equal(whole, base)
    for cut in (2, 3, 12):
        got = run((tile, cut), lambda: ops.gemm(Ag, Wf, act=_lib.ACT_SILU_MUL_PAIRS))
        assert got.shape == (M, I)
        assert_close_ulps(got, ref, 2, 0.97, f"gate/up pairs, cut {cut}")


def test_automatic_plans_of_the_cfg3_shapes(gpu):
    """BASELINE cfg 3 text-layer shapes through the automatic plan with the persistent kernel forced wherever it can run
    (AHA_GEMM_STREAMK=2 semantics via the 1256 override are per tile width; here: the planner's own cut), against the oracle."""
    from aha_amd import ops, _lib
    M = 1542
    for name, N, K, kind in [("qkv", 6144, 4096, "plain"), ("o", 4096, 4096, "res"), ("down", 4096, 12288, "res")]:
        A, W, res = rnd((M, K), 63), rnd((N, K), 64, 0.02), rnd((M, N), 65)
        Ag, Wg, rg = A.to(gpu), W.to(gpu), res.to(gpu)
        args = (None, rg) if kind == "res" else ()
        got = run((1256, 0), lambda: ops.gemm(Ag, Wg, *args))
        again = run((1256, 0), lambda: ops.gemm(Ag, Wg, *args))
        assert torch.equal(got, again)
        lin = NM.linear(A.float(), W.float())
        ref = NM.r(res.float() + lin) if kind == "res" else lin
        assert_close_ulps(got, ref, 2, 0.97, f"cfg 3 {name} on the persistent kernel")
    I, K = 12288, 4096
    A, Wg_, Wu = rnd((M, K), 66), rnd((I, K), 67, 0.02), rnd((I, K), 68, 0.02)
    ref = NM.r(NM.r(oq.silu(NM.linear(A.float(), Wg_.float()))) * NM.linear(A.float(), Wu.float()))
    Wf = ops.interleave_gate_up(Wg_, Wu).to(gpu)
    got = run((1192, 0), lambda: ops.gemm(A.to(gpu), Wf, act=_lib.ACT_SILU_MUL_PAIRS))
    base = run((256, 1), lambda: ops.gemm(A.to(gpu), Wf, act=_lib.ACT_SILU_MUL_PAIRS))
    assert_close_ulps(got, ref, 3, 0.97, "cfg 3 gate+up on the persistent kernel")      # 19 M products: a handful land 3 ulps away ...
    assert_close_ulps(base, ref, 3, 0.97, "cfg 3 gate+up on gemm256q_kernel")            # ... in the one-tile-per-block kernel as well


@pytest.mark.parametrize("reserve", [8, 32, 100])
def test_fewer_workers_than_cus_give_the_same_bits(gpu, reserve):
    """aha_hip_set_gemm_reserved_cus: the same GEMM on 248 / 224 / 152 workgroups.  Whole tiles: bit-identical to the full-width launch;
    cut tiles: the pieces (hence the f32 sums) are the same as long as the cut is the same."""
    from aha_amd import ops, _lib
    A, W, res = rnd((1542, 4096), 69), rnd((2048, 4096), 70, 0.02), rnd((1542, 2048), 71)
    Ag, Wg, rg = A.to(gpu), W.to(gpu), res.to(gpu)
    full = {cut: run((1256, cut), lambda: ops.gemm(Ag, Wg, None, rg)) for cut in (1, 2, 12)}
    assert _lib.lib().aha_hip_set_gemm_reserved_cus(reserve) == 0
    try:
        for cut, ref in full.items():
            got = run((1256, cut), lambda: ops.gemm(Ag, Wg, None, rg))
            assert torch.equal(got, ref), f"reserve {reserve}, cut {cut}"
    finally:
        _lib.lib().aha_hip_set_gemm_reserved_cus(-1)
