"""CPU (-m "not gpu"): the C-ABI library loads and exports every symbol include/aha_hip.h declares, the product path
refuses to run without it, and the host-side logic (generate loop, smart resize, request sharding) behaves like the
reference's."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "aha_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(aha_hip_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(hip_lib):
    from aha_amd import _lib
    names = _declared_symbols()
    assert len(names) >= 25
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"libaha_hip.so does not export {n}"
        assert n in _lib.SIGNATURES, f"ctypes binding is missing {n}"
    assert set(_lib.SIGNATURES) == set(names)
    assert b"gfx950" in hip_lib.aha_hip_version()


def test_struct_layout_matches_header(hip_lib):
    from aha_amd import _lib
    # aha_tensor_view: ptr, ptr, i32, i32, i64[5], i32 (+pad) ; aha_mm_input: 11 eight-byte slots + the video fields (ptr, i64, ptr, i32 + pad)
    assert ctypes.sizeof(_lib.TensorView) == 8 + 8 + 4 + 4 + 40 + 8
    assert ctypes.sizeof(_lib.MmInput) == 8 * 15
    assert _lib.MmInput.pixel_values_video.offset == 8 * 11 and _lib.MmInput.n_videos.offset == 8 * 14
    assert ctypes.sizeof(_lib.ModelDesc) % 4 == 0 and _lib.ModelDesc.stop_tokens.offset > _lib.ModelDesc.kv_reserve_tokens.offset


def test_no_gpu_is_a_loud_error_not_a_fallback(hip_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from aha_amd._lib import AhaHipError
    from aha_amd.model import HipContext
    with pytest.raises(AhaHipError):
        HipContext(0)


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "aha_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f"{f} imports the oracle"


class _FakeModel:
    """Deterministic stand-in for the C ABI so the generate loop's control flow can be checked on CPU."""

    def __init__(self, script, eos):
        self.script, self.eos, self.calls, self.i = list(script), eos, [], 0

    def stop_token_ids(self):
        return list(self.eos)

    def _next(self):
        t = self.script[self.i]
        self.i += 1
        return None, t

    def forward_initial(self, ids, off, data=None, want_logits=True):
        self.calls.append(("init", len(ids), off))
        return self._next()

    def forward_step(self, tok, off, want_logits=True):
        self.calls.append(("step", tok, off))
        return self._next()

    def clear_cache(self):
        self.calls.append(("clear",))


def test_generate_generic_control_flow():
    """generate.rs:115-159: prefill samples token 0 (never checked against eos), then up to max_tokens-1 steps with
    seqlen_offset advancing by seq_len then 1; stop right after pushing an eos id; clear_cache at the end."""
    from aha_amd.model import generate_generic
    m = _FakeModel([9, 9, 5, 7, 3, 3], eos=[7])
    toks, usage = generate_generic(m, [1, 2, 3, 4], 6)
    assert toks == [9, 9, 5, 7]
    assert m.calls == [("init", 4, 0), ("step", 9, 4), ("step", 9, 5), ("step", 5, 6), ("clear",)]
    assert usage.prompt_tokens == 4 and usage.completion_tokens == 4
    m = _FakeModel([7, 1, 2], eos=[7])               # eos from the prefill does not stop the loop (generate.rs:127-143)
    toks, _ = generate_generic(m, [1], 3)
    assert toks == [7, 1, 2]


def test_img_smart_resize():
    """img_utils.rs:295-331."""
    from aha_amd.vision_host import img_smart_resize
    assert img_smart_resize(1024, 1024) == (1024, 1024)
    assert img_smart_resize(2048, 2048) == (2048, 2048)
    assert img_smart_resize(100, 100) == (256, 256)             # below min_pixels 65536 -> scaled up, ceil to x32
    assert img_smart_resize(1000, 700) == (992, 704)            # round to the nearest multiple of 32
    assert img_smart_resize(80, 880) == (96, 896)               # 80 / 32 = 2.5 rounds AWAY from zero (f32::round), not to even
    h, w = img_smart_resize(6000, 5000)
    assert h % 32 == 0 and w % 32 == 0 and h * w <= 16777216
    with pytest.raises(ValueError):
        img_smart_resize(10, 5000)


def test_shard_units():
    from aha_amd.parallel import shard_units
    for n in (0, 1, 7, 8, 9, 64):
        for world in (1, 2, 3, 8):
            spans = [shard_units(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_v_slot_permutation_is_a_bijection():
    """common.h v_slot(): t = kk*32 + sub1*16 + G*4 + j  ->  kk*32 + G*8 + sub1*4 + j."""
    def v_slot(t):
        return (t & 32) | (((t >> 2) & 3) << 3) | (((t >> 4) & 1) << 2) | (t & 3)
    assert sorted(v_slot(t) for t in range(64)) == list(range(64))
    assert v_slot(16) == 4 and v_slot(4) == 8 and v_slot(47) == 32 + 3 * 8 + 0 * 4 + 3


# ---- get_rope_index: the library's host code vs the oracle restatement (pure integer work: bit-exact) -------------------
def _rope_case(rng, cfg, n_images):
    """A random prompt: text, then per image <|vision_start|> + t*gh*gw <|image_pad|> + <|vision_end|>, text in between
    (possibly empty), optional trailing text."""
    ids, grids = [], []
    ids += [int(x) for x in rng.integers(0, 1000, size=rng.integers(0, 5))]
    for _ in range(n_images):
        gh, gw = int(rng.integers(1, 5)), int(rng.integers(1, 6))
        grids.append([1, 2 * gh, 2 * gw])   # grid in patches; merge size 2
        ids += [cfg.vision_start_token_id] + [cfg.image_token_id] * (gh * gw) + [cfg.vision_end_token_id]
        ids += [int(x) for x in rng.integers(0, 1000, size=rng.integers(0, 4))]
    return ids, np.asarray(grids, dtype=np.uint32).reshape(-1, 3)


@pytest.mark.parametrize("seed", range(12))
def test_get_rope_index_matches_oracle(seed):
    from aha_amd.configs import tiny_qwen3vl
    from aha_amd.vision_host import get_rope_index
    from oracle import qwen3vl as ov
    cfg = tiny_qwen3vl()
    rng = np.random.default_rng(seed)
    ids, grid = _rope_case(rng, cfg, n_images=int(rng.integers(1, 4)))
    pos, delta = get_rope_index(cfg, ids, grid)
    ref_pos, ref_delta = ov.get_rope_index(ids, grid, cfg)
    assert pos.shape == (3, len(ids))
    np.testing.assert_array_equal(pos, np.asarray(ref_pos))
    assert delta == int(ref_delta) and delta <= 0


def test_get_rope_index_hand_checked_and_errors():
    from aha_amd._lib import AhaHipError
    from aha_amd.configs import tiny_qwen3vl
    from aha_amd.vision_host import get_rope_index
    cfg = tiny_qwen3vl()
    ids = [1, 2, cfg.vision_start_token_id] + [cfg.image_token_id] * 6 + [cfg.vision_end_token_id, 3, 4]
    pos, delta = get_rope_index(cfg, ids, [[1, 4, 6]])
    assert pos[:, :3].tolist() == [[0, 1, 2]] * 3
    assert pos[0, 3:9].tolist() == [3] * 6 and pos[1, 3:9].tolist() == [3, 3, 3, 4, 4, 4] and pos[2, 3:9].tolist() == [3, 4, 5] * 2
    assert pos[:, 9:].tolist() == [[6, 7, 8]] * 3 and delta == 9 - len(ids)
    p0, d0 = get_rope_index(cfg, [5, 6, 7], np.zeros((0, 3), dtype=np.uint32))   # text only: arange on all three rows
    assert p0.tolist() == [[0, 1, 2]] * 3 and d0 == 0
    with pytest.raises(AhaHipError, match="more <\\|vision_start\\|>"):   # two image runs, one grid
        get_rope_index(cfg, ids + ids, [[1, 4, 6]])
    with pytest.raises(AhaHipError, match="longer than the remaining"):   # grid says 12 tokens, prompt has 6
        get_rope_index(cfg, ids, [[1, 4, 12]])


def test_get_dtype_and_check_dtype():
    """C0: get_dtype (utils/mod.rs:77-115) as a hip feature extends it + the explicit bf16-only compute contract."""
    import ctypes as C
    from aha_amd import _lib
    lib = _lib.lib()
    BF16, F16, F32 = 0, 1, 2
    out = C.c_int32(-7)
    for cfg, want in [(b"bfloat16", BF16), (b"float16", F16), (b"float32", F32), (b"float", F32), (b"int8", F32), (b"", F32)]:
        assert lib.aha_hip_get_dtype(-1, cfg, C.byref(out)) == 0 and out.value == want, cfg
    assert lib.aha_hip_get_dtype(-1, None, C.byref(out)) == 0 and out.value == F32
    for req in (BF16, F16, F32):                      # Some(d) => d, whatever the checkpoint says
        assert lib.aha_hip_get_dtype(req, b"bfloat16", C.byref(out)) == 0 and out.value == req
    assert lib.aha_hip_get_dtype(4, b"bfloat16", C.byref(out)) < 0      # u8 is not a model dtype
    assert lib.aha_hip_check_dtype(BF16) == 0
    for bad in (F16, F32):
        assert lib.aha_hip_check_dtype(bad) == -6     # AHA_ERR_UNSUPPORTED
        assert b"bf16" in lib.aha_hip_last_error()
    assert _lib.ModelDesc.compute_dtype.offset == _lib.ModelDesc.tp_size.offset + 4


def test_rust_shim_declares_the_header_symbols():
    """rust/aha-hip/src/lib.rs cannot be compiled here (no Rust toolchain): keep it honest textually -- every extern "C" function
    it declares is declared in include/aha_hip.h and exported by the built library, and its #[repr(C)] structs list the fields
    of the ctypes mirrors (which test_struct_layouts ties to the header) in the same order."""
    import re
    from aha_amd import _lib
    src = open(os.path.join(ROOT, "rust", "aha-hip", "src", "lib.rs")).read()
    header = open(os.path.join(ROOT, "include", "aha_hip.h")).read()
    ext = src[src.index('extern "C" {'):]
    ext = ext[:ext.index("\n    }\n")]
    fns = re.findall(r"pub fn (aha_hip_\w+)\(", ext)
    assert len(fns) >= 20 and len(set(fns)) == len(fns)
    lib = _lib.lib()
    for f in fns:
        assert re.search(r"\b%s\s*\(" % f, header), f"{f} is not in include/aha_hip.h"
        assert hasattr(lib, f), f"{f} is not exported by libaha_hip.so"

    def rust_fields(name):
        body = src[src.index("pub struct %s {" % name):]
        body = body[:body.index("\n    }\n")]
        return re.findall(r"pub (\w+):", body)
    assert rust_fields("AhaModelDesc") == [n for n, _ in _lib.ModelDesc._fields_]
    assert rust_fields("AhaTensorView") == [n for n, _ in _lib.TensorView._fields_]
    assert rust_fields("AhaMmInput") == [n for n, _ in _lib.MmInput._fields_]


def test_gemm_plans_of_the_baseline_shapes(hip_lib):
    """csrc/kernels_gemm.hip plan_gemm is a cost model fitted to MI355X measurements (scripts/tune_gemm.py); the plans it yields for
    the BASELINE shapes are the ones the profiles under profiles/r02_* were taken with.  Host-only query: a change of the model's
    constants that moves one of them shows up here, on the CPU tier, and has to come with a new measurement."""
    from aha_amd import _lib

    def plan(M, N, K, act=_lib.ACT_NONE, bias=0, res=0, ws=1 << 30):
        o = (ctypes.c_int32 * 3)()
        assert hip_lib.aha_hip_debug_plan_gemm(M, N, K, act, bias, res, ws, o) == 0
        return tuple(o)

    # cfg 3 text stack at M = 1542 (8B): qkv on 256 x 192 tiles (224 tiles instead of 168 on 256 CUs, round 3), gate+up on 256 x 192
    # tiles with the six extra rows as a fifth fragment row of the last row tile (768 tiles = three rounds; round 4: the same 272 us as
    # the 256^2 tiling, profiles/r04_gemm_row5.md), o_proj / down_proj two K slices of 256^2 tiles -- and NOT the persistent kernel:
    # every cut of its last round measured slower (profiles/r04_gemm_sk.md)
    assert plan(1542, 6144, 4096) == (192, 1, 0)
    assert plan(1542, 4096, 4096, res=1) == (256, 2, 0)
    assert plan(1542, 24576, 4096, act=_lib.ACT_SILU_MUL_PAIRS) == (192, 1, 0)
    # by rounds: three exact rounds of 3/4-size tiles at M = 1536, two rounds of 256^2 tiles at M = 1280, three at M = 1792
    assert plan(1536, 24576, 4096, act=_lib.ACT_SILU_MUL_PAIRS) == (192, 1, 0)
    assert plan(1280, 24576, 4096, act=_lib.ACT_SILU_MUL_PAIRS) == (256, 1, 0)
    assert plan(1792, 24576, 4096, act=_lib.ACT_SILU_MUL_PAIRS) == (256, 1, 0)
    assert plan(1542, 4096, 12288, res=1) == (256, 2, 0)
    assert plan(1542, 4096, 4096, res=1, ws=0) == (128, 1, 0)            # no workspace: no split-K, and 112 tiles lose to the 128^2 kernel
    # ViT at N = 4096 patches: qkv on 256^2, proj on 256 x 128 (round 6), fc1 as 4096 columns + a 208-column tail; fc2 contracts over the MLP width padded
    # to whole K tiles (4352; csrc/vision_tower.hip Ipad) as two K slices of 192-column tiles = 192 blocks (scripts/bench_gemm_fc2.py:
    # 50.8 us against 54.9 us for three slices of 256^2 tiles and 61.6-64.9 us for the unpadded K = 4304 on the 8-wave kernel)
    assert plan(4096, 1152, 4352, bias=1, res=1) == (192, 2, 0)
    assert plan(4096, 3456, 1152, bias=1) == (256, 1, 0)
    assert plan(4096, 1152, 1152, bias=1, res=1) == (2128, 1, 0)   # round 6: 144 tiles of 256 x 128 on the eight-wave ring kernel (288 of 128^2 = one block per CU + 32: 31.7 -> 24.7 us in the model)
    assert plan(4096, 4304, 1152, act=_lib.ACT_GELU_TANH, bias=1) == (256, 1, 1)
    assert plan(4096, 1152, 4304, bias=1, res=1) == (256, 3, 0)
    # below one 256-row tile everything stays on the 128^2 kernel; a 41 k-token prompt fills whole rounds unsplit
    assert plan(70, 512, 512) == (128, 1, 0) and plan(255, 4096, 4096)[0] == 128
    assert plan(40980, 6144, 4096) == (256, 1, 0) and plan(40980, 4096, 12288, res=1) == (256, 1, 0)
    # the other BASELINE configs, as README.md tabulates them (round-4 verdict, item 7): cfg 1 stays on the 128^2 kernel, cfg 2 / cfg 4 (0.6B
    # widths) and one context-parallel rank of cfg 5 (~5200 compact rows)
    P = _lib.ACT_SILU_MUL_PAIRS
    assert [plan(128, 4096, 1024)[0], plan(128, 1024, 2048, res=1)[0], plan(128, 6144, 1024, act=P)[0], plan(128, 1024, 3072, res=1)[0]] == [128] * 4
    assert (plan(2048, 4096, 1024), plan(2048, 1024, 2048, res=3), plan(2048, 6144, 1024, act=P), plan(2048, 1024, 3072, res=3)) == \
        ((2128, 1, 0), (128, 2, 0), (192, 1, 0), (128, 2, 0))   # round 6: qkv = 256 tiles of 256 x 128, o / down = 128^2 tiles x 2 K slices (one round each)
    # cfg 4 (Qwen3-ASR: M = 406 text rows, 390 audio rows).  Round 6: the short-K projections (14-16 k steps) run unsplit on the 128^2 ring
    # kernel (gemm_glds_ring_kernel: one block per CU at most, a four-stage LDS-DMA ring, fragment reads half a tile ahead of the MFMAs),
    # the long-K ones (o_proj / down_proj / fc2: 32-56 k steps over 28-32 tiles) as 128^2 tiles x K slices on the same kernel + the reduce
    # pass that holds the riding RMSNorm (scripts/tune_gemm.py: 15.6 / 16.8 / 17.2 us against 21.9 / 23.5 / 24.6 for the 256^2 slices);
    # prefill 6.60 -> 4.45 ms.  has_residual bit 1 = a norm rides on the call (folded into a reduce pass, a launch of its own otherwise)
    assert (plan(406, 4096, 1024), plan(406, 1024, 2048, res=3), plan(406, 6144, 1024, act=P), plan(406, 1024, 3072, res=3)) == \
        ((128, 1, 0), (128, 4, 0), (128, 1, 0), (128, 6, 0))
    assert (plan(390, 2688, 896, bias=1), plan(390, 896, 896, bias=1, res=1), plan(390, 3584, 896, act=_lib.ACT_GELU_ERF, bias=1),
            plan(390, 896, 3584, bias=1, res=1)) == ((128, 1, 0), (128, 1, 0), (128, 1, 0), (128, 8, 0))
    # cfg 2 o_proj: with its riding norm two K slices of 128^2 tiles (256 blocks; the norm in the reduce pass), without one an unsplit ring launch;
    # few-row launches of the 8B widths (128-token prompts, 16-row batches) split K on the ring kernel too; K < 1536 never does
    assert plan(2048, 1024, 2048, res=3) == (128, 2, 0) and plan(2048, 1024, 2048, res=1) == (128, 1, 0)
    assert (plan(128, 6144, 4096), plan(128, 4096, 4096, res=3), plan(128, 4096, 12288, res=3), plan(16, 4096, 4096)) == \
        ((128, 4, 0), (128, 6, 0), (128, 8, 0), (128, 8, 0))
    assert plan(64, 512, 1024, res=3) == (128, 1, 0) and plan(406, 1024, 2048, res=3, ws=0) == (128, 1, 0)
    assert (plan(5184, 6144, 4096), plan(5184, 4096, 4096, res=1), plan(5184, 24576, 4096, act=P), plan(5184, 4096, 12288, res=1)) == \
        ((256, 1, 0), (256, 1, 0), (256, 1, 0), (256, 3, 0))
    with pytest.raises(Exception):
        assert hip_lib.aha_hip_debug_plan_gemm(0, 1, 1, 0, 0, 0, 0, (ctypes.c_int32 * 3)()) == 0


@pytest.mark.parametrize("S", [512, 1542, 2048, 8192, 40980, 131072])
@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_context_parallel_plan_partitions_the_prompt(hip_lib, S, world):
    """csrc/model.hip cp_make_plan (aha_hip_set_context_parallel): the 64-token pages of the prompt are cut into 2 x world chunks, rank r
    owns chunks r and 2 x world - 1 - r.  Every row belongs to exactly one rank, chunk boundaries are page boundaries, rank 0 owns the last
    row, the causal attention work (sum over owned rows of the keys they see) is balanced, and the staging slot count covers every rank."""
    P = -(-S // 64)
    owner = np.full(S, -1)
    work, pages = [], []
    for r in range(world):
        o = (ctypes.c_int32 * 5)()
        rc = hip_lib.aha_hip_debug_cp_plan(S, world, r, o)
        if P < 4 * world:
            assert rc == 1
            return
        assert rc == 0
        (a0, an, b0, bn, pmax) = tuple(o)
        assert a0 % 64 == 0 and b0 % 64 == 0 and an > 0 and bn > 0 and a0 + an <= b0
        assert (a0 + an) % 64 == 0 and ((b0 + bn) % 64 == 0 or b0 + bn == S)
        for (x0, n) in ((a0, an), (b0, bn)):
            assert np.all(owner[x0:x0 + n] == -1)
            owner[x0:x0 + n] = r
        rows = np.concatenate([np.arange(a0, a0 + an), np.arange(b0, b0 + bn)])
        work.append(float((rows + 1).sum()))
        pages.append(-(-an // 64) + -(-bn // 64))
        assert pages[-1] <= pmax
    assert np.all(owner >= 0) and owner[S - 1] == 0
    assert max(pages) == pmax
    # zigzag: no rank has more than ~(1 + 2 / pages-per-chunk) x the mean causal work
    assert max(work) <= np.mean(work) * (1.0 + 2.5 * 2 * world / P), (work, P)
