"""-m gpu: tensor-parallel decoder stack (SURVEY.md section 8e "next": the model does not fit / latency-bound decode).
Two model handles (tp_rank 0/1 of 2) live on the one GPU of the test box, each driven by its own thread; the all-reduce
seam is the host callback (aha_hip_set_allreduce) implemented with a barrier + a torch sum over the two device buffers.
The sharded stack must reproduce the single-GPU logits up to f32 summation order of the row-parallel projections."""
import threading

import numpy as np
import pytest
import torch

from aha_amd.configs import tiny_qwen3, tiny_qwen3vl
from aha_amd.weights import qwen3_text_weights, qwen3vl_weights

pytestmark = pytest.mark.gpu


class TwoRankSum:
    """In-place sum of two ranks' f32 device buffers; every rank calls allreduce(rank, ptr, count)."""

    def __init__(self, n=2):
        self.n = n
        self.bar = threading.Barrier(n, timeout=60)
        self.slots = [None] * n
        self.total = None
        self.calls = 0

    def view(self, ptr, count):
        # a torch tensor over the library's device buffer
        iface = {"shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 2}
        holder = type("H", (), {"__cuda_array_interface__": iface})()
        return torch.as_tensor(holder, device="cuda:0")

    def allreduce(self, rank, ptr, count):
        self.slots[rank] = self.view(ptr, count)
        self.bar.wait()
        if rank == 0:
            self.total = torch.stack([s for s in self.slots]).sum(0)
            torch.cuda.synchronize()
            self.calls += 1
        self.bar.wait()
        self.slots[rank].copy_(self.total)
        torch.cuda.synchronize()
        self.bar.wait()

    # sequence-parallel seam (aha_hip_set_seq_parallel): in place; rank r keeps / contributes slice r
    def reduce_scatter(self, rank, ptr, count_per_rank):
        self.slots[rank] = self.view(ptr, count_per_rank * self.n)
        self.bar.wait()
        if rank == 0:
            self.total = torch.stack([s for s in self.slots]).sum(0)
            torch.cuda.synchronize()
            self.rs_calls = getattr(self, "rs_calls", 0) + 1
        self.bar.wait()
        sl = slice(rank * count_per_rank, (rank + 1) * count_per_rank)
        self.slots[rank].fill_(float("nan"))           # the other slices are undefined by contract: poison them
        self.slots[rank][sl].copy_(self.total[sl])
        torch.cuda.synchronize()
        self.bar.wait()

    def view_bytes(self, ptr, nbytes):
        iface = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
        holder = type("H", (), {"__cuda_array_interface__": iface})()
        return torch.as_tensor(holder, device="cuda:0")

    def all_gather(self, rank, ptr, bytes_per_rank):
        self.slots[rank] = self.view_bytes(ptr, bytes_per_rank * self.n)
        self.bar.wait()
        for r in range(self.n):
            if r != rank:
                sl = slice(r * bytes_per_rank, (r + 1) * bytes_per_rank)
                self.slots[rank][sl].copy_(self.slots[r][sl])
        torch.cuda.synchronize()
        self.ag_calls = getattr(self, "ag_calls", 0) + 1
        self.bar.wait()


def run_ranks(fns):
    out, err = [None] * len(fns), [None] * len(fns)

    def wrap(i):
        try:
            out[i] = fns[i]()
        except BaseException as e:  # noqa: BLE001
            err[i] = e
    ts = [threading.Thread(target=wrap, args=(i,)) for i in range(len(fns))]
    [t.start() for t in ts]
    [t.join(120) for t in ts]
    for e in err:
        if e is not None:
            raise e
    return out


def close(a, b, what, rms=0.004):
    """Sharded vs unsharded logits: the row-parallel projections sum their f32 partials in another order, so individual bf16
    roundings flip.  Bound: max error 0.02 std OR one bf16 ulp of the largest logit (a single flip on a logit in [1, 2) is
    2^-7 = 0.024 std with these weights), rms 0.004 std."""
    s = float(b.std())
    ulp_max = 2.0 ** (np.floor(np.log2(max(float(np.abs(b).max()), 1e-30))) - 7)
    tol = max(0.02 * s, ulp_max)
    assert float(np.abs(a - b).max()) <= tol, f"{what}: max {np.abs(a - b).max() / s:.4f} std units (bound {tol / s:.4f})"
    assert float(np.sqrt(((a - b) ** 2).mean())) <= rms * s, f"{what}: rms {np.sqrt(((a - b) ** 2).mean()) / s:.5f} std units (bound {rms})"


@pytest.mark.parametrize("S", [5, 77, 200])
def test_tp2_matches_single_gpu_text(gpu, S):
    from aha_amd.model import HipContext, HipInferenceModel
    cfg = tiny_qwen3()
    w = qwen3_text_weights(cfg, seed=0)
    ctx = HipContext(0)
    single = HipInferenceModel(cfg, w, ctx=ctx)
    red = TwoRankSum()
    ranks = [HipInferenceModel(cfg, w, tp_rank=r, tp_size=2,
                               allreduce=lambda p, n, r=r: red.allreduce(r, p, n)) for r in range(2)]
    ids = [int(x) for x in np.random.default_rng(S).integers(0, cfg.vocab_size, size=S)]
    ref, _ = single.forward_initial(ids, 0)
    got = run_ranks([lambda m=m: m.forward_initial(ids, 0)[0].copy() for m in ranks])
    # two row-parallel projections per layer + the vocab-parallel lm_head's argmax pair exchange + the full-logits assembly
    assert red.calls == 2 * cfg.num_hidden_layers + 2
    np.testing.assert_array_equal(got[0], got[1])  # both ranks see the same reduced sums
    close(got[0], ref, f"tp2 prefill S={S}")
    tok, off = int(np.argmax(ref)), S
    for step in range(5):
        ref, _ = single.forward_step(tok, off)
        got = run_ranks([lambda m=m: m.forward_step(tok, off)[0].copy() for m in ranks])
        np.testing.assert_array_equal(got[0], got[1])
        close(got[0], ref, f"tp2 decode step {step}")
        tok, off = int(np.argmax(ref)), off + 1
    for m in ranks + [single]:
        m.close()


@pytest.mark.parametrize("S", [5, 77, 200, 257])
def test_tp2_sequence_parallel_prefill_equals_allreduce_prefill(gpu, S):
    """Sequence-parallel prefill (reduce-scatter of the f32 partial sums over row slices, RMSNorm on the owned rows, all-gather
    of the normalised bf16 rows): the same f32 sums in the same order as the all-reduce path, so the logits must be
    bit-identical to it (and close to the single-GPU ones); S = 5 leaves the second rank 2 of 3 rows, 257 is ragged."""
    from aha_amd.model import HipContext, HipInferenceModel
    cfg = tiny_qwen3()
    w = qwen3_text_weights(cfg, seed=0)
    ids = [int(x) for x in np.random.default_rng(S).integers(0, cfg.vocab_size, size=S)]
    outs = {}
    for sp in (False, True):
        red = TwoRankSum()
        kw = dict(reduce_scatter=None, all_gather=None)
        ranks = [HipInferenceModel(cfg, w, tp_rank=r, tp_size=2, allreduce=lambda p, n, r=r: red.allreduce(r, p, n),
                                   **(dict(reduce_scatter=lambda p, n, r=r: red.reduce_scatter(r, p, n),
                                           all_gather=lambda p, n, r=r: red.all_gather(r, p, n)) if sp else kw)) for r in range(2)]
        got = run_ranks([lambda m=m: m.forward_initial(ids, 0)[0].copy() for m in ranks])
        np.testing.assert_array_equal(got[0], got[1])
        if sp:
            assert red.rs_calls == 2 * cfg.num_hidden_layers and red.ag_calls == 2 * 2 * cfg.num_hidden_layers  # ag counted per rank
        else:
            assert not hasattr(red, "rs_calls")
        # decode after a sequence-parallel prefill runs on the all-reduce seam over the same KV cache
        tok = int(np.argmax(got[0]))
        step = run_ranks([lambda m=m: m.forward_step(tok, S)[0].copy() for m in ranks])
        np.testing.assert_array_equal(step[0], step[1])
        outs[sp] = (got[0], step[0])
        for m in ranks:
            m.close()
    np.testing.assert_array_equal(outs[True][0], outs[False][0])
    np.testing.assert_array_equal(outs[True][1], outs[False][1])
    single = HipInferenceModel(cfg, w)
    ref, _ = single.forward_initial(ids, 0)
    close(outs[True][0], ref, f"sequence-parallel tp2 prefill S={S}")
    single.close()


def test_tp2_greedy_tokens_and_vl(gpu):
    """Device greedy loop under TP (one callback per row-parallel projection per step) + the VL path: the vision tower
    is replicated per rank, only the decoder stack is sharded."""
    from aha_amd.model import HipContext, HipInferenceModel
    from aha_amd.vision_host import synthetic_image_request
    cfg = tiny_qwen3vl()
    w = qwen3vl_weights(cfg, seed=0)
    ctx = HipContext(0)
    single = HipInferenceModel(cfg, w, ctx=ctx)
    red = TwoRankSum()
    ranks = [HipInferenceModel(cfg, w, tp_rank=r, tp_size=2,
                               allreduce=lambda p, n, r=r: red.allreduce(r, p, n)) for r in range(2)]
    ids, mm = synthetic_image_request(cfg, 256, 9, torch.Generator().manual_seed(1), device=gpu)
    ref, am = single.forward_initial(ids, 0, mm)
    got = run_ranks([lambda m=m: m.forward_initial(ids, 0, mm)[0].copy() for m in ranks])
    close(got[0], ref, "tp2 VL prefill")
    want = single.decode_greedy(am, len(ids), 8)
    toks = run_ranks([lambda m=m: m.decode_greedy(am, len(ids), 8) for m in ranks])
    assert list(toks[0]) == list(toks[1])
    # greedy ties can flip under a different f32 summation order; the tiny random model has well separated logits
    assert list(toks[0]) == list(want)
    for m in ranks + [single]:
        m.close()


def test_tp2_greedy_loop_with_a_stop_token_mid_sequence(gpu):
    """A stop token in the middle of a TP greedy run (round-3 advisor, high): every decode step holds all-reduces, so both ranks must
    enqueue the same number of steps whatever the moment their host threads read the published tokens.  Under TP the loop enqueues
    whole groups of AHA_DECODE_RUNAHEAD steps as a function of the token sequence alone: no rank may be left waiting in a collective
    (the barrier of TwoRankSum times out after 60 s), both ranks return the tokens up to and including the stop token, executed the
    same number of steps, and a second call on the same handles (leftover collectives would pair with it) still agrees."""
    from aha_amd.model import HipContext, HipInferenceModel
    cfg = tiny_qwen3()
    w = qwen3_text_weights(cfg, seed=0)
    ids = [int(x) for x in np.random.default_rng(3).integers(0, cfg.vocab_size, size=21)]
    single = HipInferenceModel(cfg, w, ctx=HipContext(0))
    _, am = single.forward_initial(ids, 0)
    base = single.decode_greedy(am, len(ids), 24)
    single.close()
    for stop_at in (2, 5, 6):      # inside the first group of 4, first / second step of the second group
        if base[stop_at] in base[:stop_at]:
            continue
        cfg.eos_token_ids = [base[stop_at]]
        red = TwoRankSum()
        ranks = [HipInferenceModel(cfg, w, tp_rank=r, tp_size=2, allreduce=lambda p, n, r=r: red.allreduce(r, p, n)) for r in range(2)]
        run_ranks([lambda m=m: m.forward_initial(ids, 0, want_logits=False) for m in ranks])
        toks = run_ranks([lambda m=m: m.decode_greedy(am, len(ids), 24) for m in ranks])
        assert list(toks[0]) == list(toks[1]) == list(base[: stop_at + 1])
        steps = [m.debug_steps_executed() for m in ranks]
        assert steps[0] == steps[1] and stop_at + 1 <= steps[0] <= stop_at + 4, steps
        more = run_ranks([lambda m=m: m.decode_greedy(int(toks[0][-1]), len(ids) + stop_at + 1, 4) for m in ranks])
        assert list(more[0]) == list(more[1]) and len(more[0]) >= 1
        for m in ranks:
            m.close()
    cfg.eos_token_ids = []


def test_tp2_vocab_parallel_lm_head_is_exact_given_the_hidden_state(gpu):
    """With ONE decoder layer whose row-parallel partial sums are exact in f32 (weights and activations are small integers /
    powers of two in bf16), the TP stack's hidden state equals the single-GPU one bit for bit, so the vocab-sharded lm_head
    (each rank streams half the rows, argmax pairs and logits slices exchanged through the all-reduce seam) must reproduce
    logits and argmax EXACTLY, including a tie across the shard boundary (first maximal index wins)."""
    from aha_amd.model import HipContext, HipInferenceModel
    cfg = tiny_qwen3(layers=1, hidden=256, heads=4, kv_heads=2, inter=512, vocab=1024, tie=False)
    w = qwen3_text_weights(cfg, seed=0)
    g = torch.Generator().manual_seed(5)
    for k in list(w):   # integer-valued weights: every partial sum is exact, so TP == single GPU bit for bit
        if k.endswith("proj.weight"):
            w[k] = torch.randint(-2, 3, w[k].shape, generator=g).to(torch.bfloat16) / 64
    lm = torch.randint(-3, 4, w["lm_head.weight"].shape, generator=g).to(torch.bfloat16) / 8
    lm[900] = lm[100]            # identical rows on both sides of the shard boundary (512): a guaranteed tie
    w["lm_head.weight"] = lm
    single = HipInferenceModel(cfg, w, ctx=HipContext(0))
    red = TwoRankSum()
    ranks = [HipInferenceModel(cfg, w, tp_rank=r, tp_size=2, allreduce=lambda p, n, r=r: red.allreduce(r, p, n)) for r in range(2)]
    ids = [int(x) for x in np.random.default_rng(0).integers(0, cfg.vocab_size, size=9)]
    ref, am = single.forward_initial(ids, 0)
    got = run_ranks([lambda m=m: m.forward_initial(ids, 0) for m in ranks])
    for lg, tok in got:
        np.testing.assert_array_equal(lg, ref)
        assert tok == am == int(np.argmax(ref))
    assert ref[100] == ref[900]
    for m in ranks + [single]:
        m.close()


def test_tp_rejects_indivisible_heads(gpu):
    from aha_amd._lib import AhaHipError
    from aha_amd.model import HipInferenceModel
    cfg = tiny_qwen3()
    w = qwen3_text_weights(cfg, seed=0)
    with pytest.raises(AhaHipError, match="tp_size"):
        HipInferenceModel(cfg, w, tp_rank=0, tp_size=3)


def test_tp_without_allreduce_is_an_error(gpu):
    from aha_amd._lib import AhaHipError
    from aha_amd.model import HipInferenceModel
    cfg = tiny_qwen3()
    m = HipInferenceModel(cfg, qwen3_text_weights(cfg, seed=0), tp_rank=0, tp_size=2)
    with pytest.raises(AhaHipError, match="all-reduce"):
        m.forward_initial([1, 2, 3], 0)
    m.close()


def test_rccl_world1_smoke(gpu):
    """RCCL communicator of size 1 is legal: exercises unique-id / init / stream-ordered all-reduce wiring.  (tp_size 1
    skips the seam entirely, so this drives a tp_size=1 model only through init.)"""
    from aha_amd.model import HipInferenceModel, tp_unique_id
    uid = tp_unique_id()
    assert len(uid) == 128 and any(uid)
    cfg = tiny_qwen3()
    m = HipInferenceModel(cfg, qwen3_text_weights(cfg, seed=0), rccl_unique_id=uid)
    t = torch.arange(1000, dtype=torch.float32, device=gpu)
    m.debug_allreduce(t)
    assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float32))
    m.close()
    # round 5: the communication stream's own communicator (ncclCommSplit of the model's at init; tp_rccl.hip) -- forced for a group of one:
    # the split, the all-reduce beside it and the teardown of both communicators run on this RCCL build
    import os
    os.environ["AHA_TP_SIDE_COMM"] = "2"
    try:
        m = HipInferenceModel(cfg, qwen3_text_weights(cfg, seed=0), rccl_unique_id=tp_unique_id())
        m.debug_allreduce(t)
        assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float32))
        m.close()
    finally:
        del os.environ["AHA_TP_SIDE_COMM"]


def test_tp2_two_processes_gloo(gpu):
    """The TP seam across PROCESSES (one per rank, as on a multi-GPU node; here both on the box's single GPU): partial sums
    staged through host memory and summed by torch.distributed/gloo inside the aha_hip_set_allreduce callback, image-parallel
    ViT with a gloo all-gather (tests/tools/tp_worker.py).  Ranks must agree on the greedy tokens and rank 0's sharded
    logits must equal the unsharded model's up to f32 summation order."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(root, "tests", "tools", "tp_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "TP_WORKER_OK" in r.stdout, r.stdout[-2000:]
    assert "phases=[" in r.stdout and "reduce_scatter_s" in r.stdout and "vit_s" in r.stdout   # the per-phase diagnosis of the sharded prefill


def test_kv_export_import_round_trip_is_exact(gpu):
    """aha_hip_kv_export / aha_hip_kv_import (the KV hand-back after a sharded prefill): the packed buffer is the byte image of the
    pages, so a cache exported from one model and imported into a fresh one -- as ONE shard, and as two half-head shards the way a
    TP = 2 gather delivers them, also under a scrambled physical page order on the importing side -- must decode bit-identically,
    text and Qwen3-VL (rope_delta travels with the cache)."""
    from aha_amd.model import HipInferenceModel, MultiModalData
    from oracle.numerics import Numerics
    from oracle import qwen3vl as ov
    cfg = tiny_qwen3()
    w = qwen3_text_weights(cfg, seed=0)
    ids = [int(x) for x in np.random.default_rng(5).integers(0, cfg.vocab_size, size=150)]   # 3 pages, the last one partly filled
    a = HipInferenceModel(cfg, w)
    _, tok = a.forward_initial(ids, 0)
    buf, ntok, delta = a.kv_export()
    kvh = cfg.num_key_value_heads
    assert ntok == 150 and buf.numel() == cfg.num_hidden_layers * 3 * kvh * 2 * 16384
    ref = [a.forward_step(tok, 150)[0].copy()]
    ref.append(a.forward_step(7, 151)[0].copy())
    for halves, scramble in ((False, False), (True, False), (True, True)):
        c = HipInferenceModel(cfg, w)
        c.debug_scramble_pages(scramble)
        if halves:
            h = kvh // 2
            c.kv_import(buf, kvh, 0, 0, h, ntok, delta)          # heads [0, h) of the buffer -> heads [0, h)
            c.kv_import(buf, kvh, h, h, kvh - h, ntok, delta)    # heads [h, kvh)
        else:
            c.kv_import(buf, kvh, 0, 0, kvh, ntok, delta)
        assert c.cache_len() == 150
        np.testing.assert_array_equal(c.forward_step(tok, 150)[0], ref[0])
        np.testing.assert_array_equal(c.forward_step(7, 151)[0], ref[1])
        c.close()
    a.close()
    # a shard buffer that holds only h heads (what a TP rank exports): built by slicing the full buffer's head axis
    full = buf.view(cfg.num_hidden_layers, 3, kvh, 2 * 16384)
    c = HipInferenceModel(cfg, w)
    for r in range(2):
        h = kvh // 2
        shard = full[:, :, r * h:(r + 1) * h].contiguous().view(-1)
        c.kv_import(shard, h, 0, r * h, h, ntok, delta)
    np.testing.assert_array_equal(c.forward_step(tok, 150)[0], ref[0])
    c.close()
    # Qwen3-VL: the decode position is seqlen_offset + rope_delta
    vcfg = tiny_qwen3vl()
    vw = qwen3vl_weights(vcfg, seed=0)
    g = np.random.default_rng(7)
    img = g.integers(0, 256, size=(96, 160, 3), dtype=np.uint8)
    pv, grid = ov.process_images(Numerics("bf16"), [img])
    vids = [3, 4] + [vcfg.vision_start_token_id] + [vcfg.image_token_id] * (int(grid[0, 1] * grid[0, 2]) // 4) + [vcfg.vision_end_token_id] + [9, 10, 11]
    va = HipInferenceModel(vcfg, vw)
    _, vt = va.forward_initial(vids, 0, MultiModalData(pv.to(torch.bfloat16), grid))
    vbuf, vn, vdelta = va.kv_export()
    assert vdelta < 0
    vref = va.forward_step(vt, len(vids))[0].copy()
    vc = HipInferenceModel(vcfg, vw)
    vc.kv_import(vbuf, vcfg.text.num_key_value_heads, 0, 0, vcfg.text.num_key_value_heads, vn, vdelta)
    np.testing.assert_array_equal(vc.forward_step(vt, len(vids))[0], vref)
    va.close(); vc.close()
    # bounds are checked
    from aha_amd._lib import AhaHipError
    c = HipInferenceModel(cfg, w)
    with pytest.raises(AhaHipError):
        c.kv_import(buf, kvh, 0, 1, kvh, ntok, delta)
    # the buffer's size is part of the call: a token count (or head count) the buffer cannot hold is refused before anything is read
    with pytest.raises(AhaHipError, match="bytes"):
        c.kv_import(buf, kvh, 0, 0, kvh, ntok + 64, delta)
    with pytest.raises(AhaHipError, match="bytes"):
        c.kv_import(buf[: buf.numel() // 2].contiguous(), kvh, 0, 0, kvh, ntok, delta)
    # a logits query after an import must not return the previous cache's logits
    c.forward_initial(ids[:5], 0)
    c.clear_cache()
    c.kv_import(buf, kvh, 0, 0, kvh, ntok, delta)
    with pytest.raises(AhaHipError):
        c.sample_candidates([], 1.0, 1.0, 4)
    c.close()
    t = HipInferenceModel(cfg, w, tp_rank=0, tp_size=2, allreduce=lambda p, n: 0)
    with pytest.raises(AhaHipError, match="tensor-parallel"):
        t.kv_import(buf, kvh, 0, 0, kvh // 2, ntok, delta)
    t.close()


@pytest.mark.parametrize("S", [300, 513])
def test_tp2_column_chunked_reduce_scatter_equals_the_unchunked_paths(gpu, S, monkeypatch):
    """The overlap form of the sequence-parallel prefill (model.hip gemm_row_parallel: the row-parallel projection cut into column
    blocks, one reduce-scatter per block -- on the RCCL communication stream while the next block's GEMM runs; through the host
    callbacks here, which serialise it but move the same data): every output element is still one K-sum and one sum over ranks, so
    the logits must equal the unchunked sequence-parallel path AND the all-reduce path bit for bit.  hidden 512 -> two 256-column
    blocks; ragged row slices at S = 513."""
    from aha_amd.model import HipInferenceModel
    cfg = tiny_qwen3(layers=2, hidden=512, heads=4, kv_heads=2, inter=1024, vocab=1024)
    w = qwen3_text_weights(cfg, seed=0)
    ids = [int(x) for x in np.random.default_rng(S).integers(0, cfg.vocab_size, size=S)]
    outs = {}
    for mode in ("allreduce", "sp", "sp_chunked"):
        monkeypatch.setenv("AHA_TP_OVERLAP_MIN_ROWS", "64" if mode == "sp_chunked" else "1000000")
        red = TwoRankSum()
        sp = mode != "allreduce"
        ranks = [HipInferenceModel(cfg, w, tp_rank=r, tp_size=2, allreduce=lambda p, n, r=r: red.allreduce(r, p, n),
                                   reduce_scatter=(lambda p, n, r=r: red.reduce_scatter(r, p, n)) if sp else None,
                                   all_gather=(lambda p, n, r=r: red.all_gather(r, p, n)) if sp else None) for r in range(2)]
        got = run_ranks([lambda m=m: m.forward_initial(ids, 0)[0].copy() for m in ranks])
        np.testing.assert_array_equal(got[0], got[1])
        if mode == "sp":
            assert red.rs_calls == 2 * cfg.num_hidden_layers
        if mode == "sp_chunked":
            assert red.rs_calls == 2 * 2 * cfg.num_hidden_layers      # two column blocks per projection
        tok = int(np.argmax(got[0]))
        step = run_ranks([lambda m=m: m.forward_step(tok, S)[0].copy() for m in ranks])
        outs[mode] = (got[0], step[0])
        for m in ranks:
            m.close()
    for mode in ("sp", "sp_chunked"):
        np.testing.assert_array_equal(outs[mode][0], outs["allreduce"][0])
        np.testing.assert_array_equal(outs[mode][1], outs["allreduce"][1])


@pytest.mark.parametrize("S,tile", [(1200, 256), (1201, 256), (1100, 192), (700, 128)])
def test_tp2_chunked_all_gather_equals_the_single_all_gather(gpu, S, tile, monkeypatch):
    """Round 4 (model.hip norm_gather_gemm): the all-gather of the normalised rows in front of the column-parallel projections runs in row
    chunks (whole row tiles of every rank's slice; on the RCCL communication stream beside the previous chunk's GEMM -- through the host
    callbacks here, which serialise it but move the same data), each chunk consumed by ONE row-grouped GEMM out of the staging layout
    [chunk][rank][rows].  A GEMM's rows are independent, so on the same kernel every output element is the same K-ordered sum: logits of
    the prefill and of a decode step (i.e. the KV cache) bit-identical to the single-gather path.  The tile is forced so that both paths
    run the same kernel (256 / 192: the grouped launch; 128: one launch per segment); S = 1201: ragged slices, the last rank one row
    short; 2 chunks of 512 + 88 / 89 rows per rank."""
    from aha_amd import ops
    from aha_amd.model import HipInferenceModel
    cfg = tiny_qwen3(layers=2, hidden=512, heads=4, kv_heads=2, inter=1536, vocab=1024)
    w = qwen3_text_weights(cfg, seed=0)
    ids = [int(x) for x in np.random.default_rng(S).integers(0, cfg.vocab_size, size=S)]
    monkeypatch.setenv("AHA_TP_AG_CHUNKS", "2")
    outs = {}
    ops.gemm_plan(tile, 1)
    try:
        for mode in ("single", "chunked"):
            monkeypatch.setenv("AHA_TP_AG_MIN_ROWS", "64" if mode == "chunked" else "100000000")
            red = TwoRankSum()
            ranks = [HipInferenceModel(cfg, w, tp_rank=r, tp_size=2, allreduce=lambda p, n, r=r: red.allreduce(r, p, n),
                                       reduce_scatter=lambda p, n, r=r: red.reduce_scatter(r, p, n),
                                       all_gather=lambda p, n, r=r: red.all_gather(r, p, n)) for r in range(2)]
            got = run_ranks([lambda m=m: m.forward_initial(ids, 0)[0].copy() for m in ranks])
            np.testing.assert_array_equal(got[0], got[1])
            # two gathers per layer (qkv, gate+up), each in two chunks; x 2 ranks' callbacks
            assert red.ag_calls == 2 * 2 * cfg.num_hidden_layers * (2 if mode == "chunked" else 1)
            tok = int(np.argmax(got[0]))
            step = run_ranks([lambda m=m: m.forward_step(tok, S)[0].copy() for m in ranks])
            outs[mode] = (got[0], step[0])
            for m in ranks:
                m.close()
    finally:
        ops.gemm_plan(0, 0)
    np.testing.assert_array_equal(outs["chunked"][0], outs["single"][0])
    np.testing.assert_array_equal(outs["chunked"][1], outs["single"][1])
    # and with the automatic plans (the two paths may then run different kernels): the sharded-vs-unsharded bound
    monkeypatch.setenv("AHA_TP_AG_MIN_ROWS", "64")
    red = TwoRankSum()
    ranks = [HipInferenceModel(cfg, w, tp_rank=r, tp_size=2, allreduce=lambda p, n, r=r: red.allreduce(r, p, n),
                               reduce_scatter=lambda p, n, r=r: red.reduce_scatter(r, p, n),
                               all_gather=lambda p, n, r=r: red.all_gather(r, p, n)) for r in range(2)]
    got = run_ranks([lambda m=m: m.forward_initial(ids, 0)[0].copy() for m in ranks])
    for m in ranks:
        m.close()
    close(got[0], outs["single"][0], "chunked all-gather, automatic plans")


def test_tp2_prefill_with_reserved_cus_runs_the_persistent_gemm_and_keeps_the_bits(gpu, monkeypatch):
    """Round 4: when the RCCL communication stream is set up the library reserves 16 CUs for it (aha_hip_set_gemm_reserved_cus) and
    every eligible GEMM of the rank -- the f32-partial row-parallel projections included -- then runs as the persistent kernel on
    CUs - 16 workgroups (whole tiles: the planner cuts nothing that costs more than it saves).  Same K-ordered sums per element, so a
    sequence-parallel prefill + a decode step under the reservation must agree with the unreserved run within the sharded-vs-unsharded
    bound, and two different reservations with each other bit for bit (host-callback seam)."""
    from aha_amd import _lib
    from aha_amd.model import HipInferenceModel
    cfg = tiny_qwen3(layers=2, hidden=512, heads=4, kv_heads=2, inter=1024, vocab=1024)
    w = qwen3_text_weights(cfg, seed=0)
    S = 600
    ids = [int(x) for x in np.random.default_rng(S).integers(0, cfg.vocab_size, size=S)]
    monkeypatch.setenv("AHA_TP_OVERLAP_MIN_ROWS", "64")
    outs = []
    for reserve in (0, 16, 96):
        assert _lib.lib().aha_hip_set_gemm_reserved_cus(reserve) == 0
        try:
            red = TwoRankSum()
            ranks = [HipInferenceModel(cfg, w, tp_rank=r, tp_size=2, allreduce=lambda p, n, r=r: red.allreduce(r, p, n),
                                       reduce_scatter=lambda p, n, r=r: red.reduce_scatter(r, p, n),
                                       all_gather=lambda p, n, r=r: red.all_gather(r, p, n)) for r in range(2)]
            got = run_ranks([lambda m=m: m.forward_initial(ids, 0)[0].copy() for m in ranks])
            tok = int(np.argmax(got[0]))
            step = run_ranks([lambda m=m: m.forward_step(tok, S)[0].copy() for m in ranks])
            outs.append((got[0], step[0]))
            for m in ranks:
                m.close()
        finally:
            _lib.lib().aha_hip_set_gemm_reserved_cus(-1)
    # (not bit-equal to the unreserved run: that one takes f32-slab split-K plans for some projections, the persistent kernel sums
    # every element's K range in one piece -- another f32 summation order, the same bound as sharded vs unsharded)
    for o in outs[1:]:
        for got, ref, what in ((o[0], outs[0][0], "prefill"), (o[1], outs[0][1], "decode step")):
            s = float(ref.std())
            assert float(np.abs(got - ref).max()) <= 0.04 * s and float(np.sqrt(((got - ref) ** 2).mean())) <= 0.01 * s, what
    np.testing.assert_array_equal(outs[1][0], outs[2][0])    # 240 and 160 workers: whole tiles either way -> the same bits


def test_tp8_sequence_parallel_prefill_at_the_cfg5_prompt_length(gpu, monkeypatch):
    """The tensor-parallel form at BASELINE cfg 5's geometry, which no 2-rank tiny-model test reaches: EIGHT ranks (one kv head + four q
    heads each, MLP columns 1536), 8B widths, 40 980 tokens, 3 of the 36 layers -- and NO environment overrides, so the thresholds and
    chunk counts are the ones the first multi-GPU run will use: row-parallel projections in 4 column blocks of 1024 with one reduce-scatter
    each, the all-gather of the normalised rows in 4 row chunks of 1536 / 515 rows per rank, each consumed by one row-grouped GEMM launch
    (8 segments).  Host-callback seam (the collectives serialise; the data movement and the GEMM launches are the real ones).  All ranks
    must agree bit for bit, and match the single-GPU prefill within the full-size bound; a decode step on the sharded stack follows."""
    from aha_amd.configs import qwen3vl_8b_text
    from aha_amd.model import HipInferenceModel
    from tests.test_fullsize_gpu import close as close_full, rnd_ids
    for k in ("AHA_TP_AG_CHUNKS", "AHA_TP_AG_MIN_ROWS", "AHA_TP_OVERLAP_CHUNKS", "AHA_TP_OVERLAP_MIN_ROWS", "AHA_TP_SP"):
        monkeypatch.delenv(k, raising=False)
    cfg = qwen3vl_8b_text()
    cfg.num_hidden_layers = 3
    w = qwen3_text_weights(cfg, seed=2, device=gpu)
    S, T = 40980, 8
    ids = rnd_ids(cfg.vocab_size, S, 9)
    single = HipInferenceModel(cfg, w)
    ref, rtok = single.forward_initial(ids, 0)
    ref = ref.copy()
    red = TwoRankSum(T)
    ranks = [HipInferenceModel(cfg, w, tp_rank=r, tp_size=T, allreduce=lambda p, n, r=r: red.allreduce(r, p, n),
                               reduce_scatter=lambda p, n, r=r: red.reduce_scatter(r, p, n),
                               all_gather=lambda p, n, r=r: red.all_gather(r, p, n)) for r in range(T)]
    del w
    torch.cuda.empty_cache()
    got = run_ranks([lambda m=m: m.forward_initial(ids, 0)[0].copy() for m in ranks])
    L = cfg.num_hidden_layers
    assert red.rs_calls == 2 * L * 4, red.rs_calls                      # o_proj and down_proj, 4 column blocks each
    assert red.ag_calls == 2 * L * 4 * T, red.ag_calls                  # qkv and gate+up inputs, 4 row chunks each, counted per rank
    for r in range(1, T):
        np.testing.assert_array_equal(got[r], got[0])
    close_full(got[0], ref, "8-rank sequence-parallel TP prefill at 40 980 tokens vs one GPU")
    tok = int(np.argmax(ref))
    want = single.forward_step(tok, S)[0].copy()
    step = run_ranks([lambda m=m: m.forward_step(tok, S)[0].copy() for m in ranks])
    close_full(step[0], want, "decode step on the 8-way sharded stack")
    for m in ranks + [single]:
        m.close()
