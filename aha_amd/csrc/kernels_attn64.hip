// Prefill attention, the one-wave-per-SIMD form (round 6; round-5 verdict, next-round item 1): a workgroup = 4 waves = one 256-row
// Q block, each wave 64 q rows (two 32-row q blocks) on v_mfma_f32_32x32x16_bf16 with the whole 512-register file of its SIMD.
// Same semantics, same page layout and the same rounding points as attn_prefill_kernel's f32 score chain (kernels_attn.hip, SMX 3;
// reference op sequence: eager_attention_forward, /root/reference/src/models/common/modules.rs:757-813 -- scores = q . k^T * scale,
// + causal mask, softmax, . v): the scores stay the f32 QK^T accumulators through scale, mask, maximum and exponential, P is rounded
// to bf16 once for the P.V MFMA, the row sum is the f32 sum of the un-rounded probabilities (text) or rides a ones row of the V^T pad
// through the matrix pipe (ViT, LSUM).  What differs is the work per flop:
//   * a 1-KB fragment read from LDS feeds TWO 32x32x16 MFMAs (both q blocks of the wave) = 64 Ki MACs, where the 16-row kernel's read
//     feeds one 16x16x32 MFMA = 8 Ki MACs: 1/8 of the LDS read traffic per flop;
//   * a staged K / V^T tile serves 256 q rows instead of 128 (or 64): half the LDS-DMA volume per flop;
//   * no partner wave on the SIMD: nothing to arbitrate the matrix pipe or the VALU issue with.
//
// Fragment scheme (wave64; q = lane & 31, hi = lane >> 5; crow(r, hi) = (r & 3) + 8 * (r >> 2) + 4 * hi):
//   S^T block (32 tokens x 32 q rows) = K . Q^T :  A = K   (row = token q,  k = dims ks*16 + 8*hi .. +8)
//                                                  B = Q^T (col = q row q,  k = dims ks*16 + 8*hi .. +8: 16 B straight from a q row)
//                                                  C[r]  = S[token crow(r, hi)][q row q]     -> softmax statistics are per lane column
//   O^T block (32 dims x 32 q rows)   = V^T . P^T: A = V^T (row = dim q,    k = 8 token slots)
//                                                  B = P^T (col = q row q,  k = the tokens of C[4s'..4s'+3] and C[8+4s'..8+4s'+3])
//                                                  C[r]  = O[q row q][dim crow(r, hi)]
// The fragment-major KV pages (common.h kpage_elem / vpage_elem, written for v_mfma_f32_16x16x32_bf16) feed this unchanged: lanes 0-15 /
// 16-31 read the same 256-byte window of two neighbouring 1-KB fragments (tokens or dims 0-15 / 16-31 of the 32-row block), lanes 32-63
// the next window -- every 16-lane group of a ds_read_b128 covers 256 contiguous bytes: conflict-free.  The V slot permutation (v_slot)
// is exactly the token order the 32x32 accumulator hands a lane: slot G'*8 + i of a 32-token run holds token (i >> 2) * 16 + G' * 4 +
// (i & 3), and with G' = 2 s' + hi that is crow(4 s' + (i & 3) + 8 (i >> 2), hi) -- so P goes from the S^T accumulators through
// v_cvt_pk_bf16_f32 straight into the next MFMA's B operand, no permlane, no LDS.
//
// Staging: K / V^T tiles by LDS-DMA (buffer_load_dwordx4 ... lds, one 1-KB piece per wave-instruction, buffer resources built from the
// scalar page pointer: out-of-range pieces -- the ViT's 10-KB V^T block read as 12 KB -- write zeros and fetch nothing) into a ring of
// three stages, two tiles ahead; one counted s_waitcnt vmcnt + one s_barrier per tile.
#include <stdio.h>
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "attn_common.h"

namespace aha {

namespace {

extern __shared__ __attribute__((aligned(16))) char attn64_smem[];   // the block's dynamic LDS (the only __shared__ object of the unit)

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
__device__ __forceinline__ f32x16_t mfma32(bf16x8_t a, bf16x8_t b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// QK^T MFMAs as asm statements with the register classes spelled out: S^T accumulators in the architectural VGPRs (the softmax reads
// them with vector instructions), the Q^T operand in the accumulator half of the 512-entry file.  With builtins the compiler puts every
// MFMA result of a 512-register kernel into AGPRs and keeps Q wherever it fits: 450 v_accvgpr_read copies per tile.  What hipcc does not
// do for an asm MFMA (cdna_hip_programming.md section 5.7): pad the MFMA-result -> vector-ALU-read hazard -- mfma_settle() below.
__device__ __forceinline__ void mfma32_first(f32x16_t& acc, bf16x8_t a, bf16x8_t b_agpr) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b_agpr));
}
__device__ __forceinline__ void mfma32_acc(f32x16_t& acc, bf16x8_t a, bf16x8_t b_agpr) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b_agpr));
}
// 20 wait states between the last asm MFMA that writes the four S^T blocks and their first vector-ALU reader (an 8-pass MFMA result needs
// 11, a 16-pass one 19); naming the blocks read-write keeps every reader below the statement
__device__ __forceinline__ void mfma_settle(f32x16_t& s00, f32x16_t& s01, f32x16_t& s10, f32x16_t& s11) {
  asm volatile("s_nop 15\n\ts_nop 3" : "+v"(s00), "+v"(s01), "+v"(s10), "+v"(s11));
}
// The O^T accumulators are ASM-OWNED: block (db, qb) of the wave's 64 x DV output lives in a[(db*2+qb)*16 .. +15], named literally in the
// statements below and in no C++ object.  Every statement that touches them lists a0..a127 as clobbers, so the compiler keeps its own AGPR
// values (the Q^T fragments, "a" operands of the QK^T MFMAs) above them and the kernel descriptor covers them; tests/test_isa_cpu.py audits
// the compiled kernels for any instruction outside these statements that names a0..a127.  Why not "+a" operands: each asm def is a new
// virtual register, and around the rescale branch / the tile loop the allocator renamed one q block's 64 accumulators and copied them
// a -> a on the fast path of every tile (v_accvgpr_mov x 128), or spilled; as builtins the compiler picked the VGPR form and copied all
// 128 a -> v -> a per tile.  (cdna_hip_programming.md section 5.7 item 4; register index expressions a[%c2+3] are assembler arithmetic.)
#define ATTN64_O_CLOBBERS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
#define ATTN64_REP16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
template <int BASE>
__device__ __forceinline__ void o_mfma(bf16x8_t a, bf16x8_t b) {   // O^T block += A . B
  asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" : : "v"(a), "v"(b), "i"(BASE), "i"(BASE + 15) : ATTN64_O_CLOBBERS);
}
template <int BASE>
__device__ __forceinline__ void o_zero() {
#define ATTN64_Z(i) "v_accvgpr_write_b32 a[%c0+" #i "], 0\n\t"
  asm volatile(ATTN64_REP16(ATTN64_Z) : : "i"(BASE) : ATTN64_O_CLOBBERS);
#undef ATTN64_Z
}
// block *= alpha (per lane = per q column), four registers in flight.  SETTLE: the statement opens with the 20 wait states an MFMA result
// needs in front of a vector-ALU reader (hipcc pads nothing inside or around an asm statement)
template <int BASE, bool SETTLE>
__device__ __forceinline__ void o_scale(float alpha) {
  float t0, t1, t2, t3;
#define ATTN64_S4(i0, i1, i2, i3)                                                                                      \
  "v_accvgpr_read_b32 %0, a[%c5+" #i0 "]\n\tv_accvgpr_read_b32 %1, a[%c5+" #i1 "]\n\t"                                 \
  "v_accvgpr_read_b32 %2, a[%c5+" #i2 "]\n\tv_accvgpr_read_b32 %3, a[%c5+" #i3 "]\n\t"                                 \
  "v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %3, %3, %4\n\t"                   \
  "v_accvgpr_write_b32 a[%c5+" #i0 "], %0\n\tv_accvgpr_write_b32 a[%c5+" #i1 "], %1\n\t"                               \
  "v_accvgpr_write_b32 a[%c5+" #i2 "], %2\n\tv_accvgpr_write_b32 a[%c5+" #i3 "], %3\n\t"
  if (SETTLE)
    asm volatile("s_nop 15\n\ts_nop 3\n\t" ATTN64_S4(0, 1, 2, 3) ATTN64_S4(4, 5, 6, 7) ATTN64_S4(8, 9, 10, 11) ATTN64_S4(12, 13, 14, 15)
                 : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(alpha), "i"(BASE) : ATTN64_O_CLOBBERS);
  else
    asm volatile(ATTN64_S4(0, 1, 2, 3) ATTN64_S4(4, 5, 6, 7) ATTN64_S4(8, 9, 10, 11) ATTN64_S4(12, 13, 14, 15)
                 : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(alpha), "i"(BASE) : ATTN64_O_CLOBBERS);
#undef ATTN64_S4
}
// the block into sixteen VGPR values (epilogue).  The outputs are early-clobber: written while the statement still runs.
template <int BASE, bool SETTLE>
__device__ __forceinline__ void o_read(float (&x)[16]) {
#define ATTN64_R(i) "v_accvgpr_read_b32 %" #i ", a[%c16+" #i "]\n\t"
  if (SETTLE)
    asm volatile("s_nop 15\n\ts_nop 3\n\t" ATTN64_REP16(ATTN64_R)
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]), "=&v"(x[8]), "=&v"(x[9]),
                   "=&v"(x[10]), "=&v"(x[11]), "=&v"(x[12]), "=&v"(x[13]), "=&v"(x[14]), "=&v"(x[15])
                 : "i"(BASE) : ATTN64_O_CLOBBERS);
  else
    asm volatile(ATTN64_REP16(ATTN64_R)
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]), "=&v"(x[8]), "=&v"(x[9]),
                   "=&v"(x[10]), "=&v"(x[11]), "=&v"(x[12]), "=&v"(x[13]), "=&v"(x[14]), "=&v"(x[15])
                 : "i"(BASE) : ATTN64_O_CLOBBERS);
#undef ATTN64_R
}
// compile-time loop: f(std::integral_constant<int, 0>) ... f(<N - 1>) -- the "i" operands above need constants, not unrolled loop variables
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}
// two wait states between the vector ALU writing the P^T fragments and the first MFMA that reads them
__device__ __forceinline__ void valu_settle(bf16x8_t& p0, bf16x8_t& p1) {
  asm volatile("s_nop 1" : "+v"(p0), "+v"(p1));
}
__device__ __forceinline__ void valu_settle4(u32x4_t& p0, u32x4_t& p1) {
  asm volatile("s_nop 1" : "+v"(p0), "+v"(p1));
}
// v_permlane32_swap of a value with itself: one of the two results is the lane's own value, the other the value lane ^ 32 holds
__device__ __forceinline__ float xhi_max(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhi_sum(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xhi_low(float v) {   // the value of the column's hi = 0 lane, in both lanes
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]);
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int ATTN64_ROWS = 256;   // q rows per workgroup

// PIPE: 0 = a tile's four parts in program order (QK^T, maximum, probabilities, P.V); 1 = software-pipelined inside the wave: QK^T of
// tile t+1 beside the exponentials of tile t, P.V of tile t beside the maximum of tile t+1 (two S^T register sets)
// TRACE (debug builds only, -DAHA_DEBUG_KERNELS + AHA_ATTN64_TRACE=1): the four waves of the middle block add up the shader cycles of the
// pipelined iteration's parts (phase A, phase B, rescale + wait + barrier)
template <int DQK, int DV, bool LSUM, int PIPE, bool TRACE = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_prefill64_kernel(AttnPrefillArgs a, unsigned long long* trace = nullptr) {
  constexpr int KS = DQK / 32;          // 32-dim fragment columns of a K block
  // k-steps of QK^T that carry data.  DQK = 96 is head_dim 72 padded (VIT_DQK): dims 72..95 of every Q and K row are zeros, so the sixth
  // 16-deep k-step (dims 80..95) only adds zeros -- skipped: 20 instead of 24 QK^T MFMAs and 10 instead of 12 K fragment reads per tile
  constexpr int NKS = DQK == 96 ? 5 : DQK / 16;
  constexpr int NDB = (DV + 31) / 32;   // 32-dim output blocks
  constexpr int KBYTES = KV_PAGE_TOKENS * DQK * 2, VBYTES = DV * KV_PAGE_TOKENS * 2;
  constexpr int VSPAN = NDB * 4096;     // bytes of V^T the fragment reads cover (ViT: 12 KB of a 10-KB block; the rest arrives as zeros)
  constexpr int KPW = KBYTES / 4096, VPW = VSPAN / 4096;   // 1-KB DMA pieces per wave and tile: 4 + 4 (text), 3 + 3 (ViT)
  constexpr int NP = KPW + VPW;
  constexpr int STAGE = KBYTES + VSPAN;
  static_assert(KBYTES % 4096 == 0 && NP < 16, "piece counts");
  char* const smem = attn64_smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q32 = lane & 31, hi = lane >> 5;

  // ---- block -> (head, q block, segment): the XCD-aware order of attn_prefill_kernel with 256-row blocks ----
  int head, qblk, seg_rows = a.S, seg_off = a.kv_offset, seg_tot = a.kv_total, seg_row0 = 0;
  if (a.nqb > 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int g = a.nh / a.kvh, hpx = a.nh >> 3;
    const int hq = slot % hpx, qi = slot / hpx;
    head = (xcd + 8 * (hq / g)) * g + hq % g;
    qblk = a.causal ? a.nqb - 1 - qi : qi;
    if (a.S2 > 0) {
      const int nqb2 = (a.S2 + ATTN64_ROWS - 1) / ATTN64_ROWS;
      if (qi < nqb2) {
        qblk = nqb2 - 1 - qi;
        seg_rows = a.S2, seg_off = a.kv_offset2, seg_tot = a.kv_total2, seg_row0 = a.S;
      }
    }
  } else {
    head = blockIdx.y;
    qblk = blockIdx.x;
  }
  const int kvhd = head / (a.nh / a.kvh);
  const int qb0 = qblk * ATTN64_ROWS;   // first q row of the block
  const int q0 = qb0 + wave * 64;       // first q row of the wave

  // ---- Q^T operand fragments, straight from the q rows ----
  bf16x8_t qf[2][NKS];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow = seg_row0 + min(q0 + qb * 32 + q32, seg_rows - 1);
    const bf16_t* qp = (const bf16_t*)a.q + (int64_t)qrow * (a.q_ld ? a.q_ld : (int64_t)a.nh * DQK) + (int64_t)head * DQK;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[qb][ks] = as_frag(ld16(qp + ks * 16 + hi * 8));
  }
  const int blk_last_q = min(qb0 + ATTN64_ROWS - 1, seg_rows - 1);
  const int last_tok = a.causal ? min(seg_off + blk_last_q, seg_tot - 1) : seg_tot - 1;
  const int ntiles = last_tok / KV_PAGE_TOKENS + 1;
  // tiles this WAVE needs (causal: up to its own last row); past them it only stages and meets the barriers
  const int wv_last_q = min(q0 + 63, seg_rows - 1);
  const int wtiles = (q0 < seg_rows) ? ((a.causal ? min(seg_off + wv_last_q, seg_tot - 1) : seg_tot - 1) / KV_PAGE_TOKENS + 1) : 0;

  // ---- staging ----
  typedef const __attribute__((address_space(4))) uint64_t* cptr64_t;
  const cptr64_t ptab = (cptr64_t)(uintptr_t)a.kv.page_ptrs;
  int voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) voff[i] = (wave + 4 * i) * 1024 + lane * 16;
  const uint64_t koff = a.kv.layer_off + (uint64_t)kvhd * KBYTES;
  const uint64_t voffb = a.kv.layer_off + (uint64_t)a.kvh * KBYTES + (uint64_t)kvhd * VBYTES;
  auto dma_tile = [&](uint64_t page, int stage_off) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(page + koff), 0, KBYTES, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(page + voffb), 0, VBYTES, 0x00020000);
    char* dst = smem + stage_off;
#pragma unroll
    for (int i = 0; i < KPW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_ptr_t)(dst + (wave + 4 * i) * 1024), 16, voff[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < VPW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr_t)(dst + KBYTES + (wave + 4 * i) * 1024), 16, voff[i], 0, 0, 0);
  };
#define ATTN64_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0F70 | (n))
#define ATTN64_BAR()                             \
  do {                                           \
    __builtin_amdgcn_sched_barrier(0);           \
    __builtin_amdgcn_s_barrier();                \
    __builtin_amdgcn_sched_barrier(0);           \
  } while (0)

  // ---- per-lane fragment read bases (everything else of a fragment address is an instruction offset) ----
  const int lk = (((q32 >> 4) * KS * 64) + hi * 16 + (q32 & 15)) * 16;
  const int lv = KBYTES + (((q32 >> 4) * 2 * 64) + hi * 16 + (q32 & 15)) * 16;
  auto kfrag = [&](int base, int tb, int ks) __attribute__((always_inline)) {
    return as_frag(*reinterpret_cast<const u32x4_t*>(smem + base + ((tb * 2 * KS + (ks >> 1)) * 1024 + (ks & 1) * 512)));
  };
  auto vfrag = [&](int base, int db, int kstep) __attribute__((always_inline)) {   // kstep = kk * 2 + s'
    return as_frag(*reinterpret_cast<const u32x4_t*>(smem + base + ((db * 4 + (kstep >> 1)) * 1024 + (kstep & 1) * 512)));
  };

  // ---- online-softmax state ----
  // The running maximum a row's exponentials refer to is kept QUANTISED: mq = the true running maximum in log2 units (score * scale *
  // log2 e) rounded UP to a multiple of MQ = 8.  p = exp2(s c2 - mq) is the reference's e^(s - max) times a per-row constant in (2^-8, 1]
  // that cancels in p / sum(p): same rounding points (P -> bf16 once, f32 row sum), and
  //   * the rescale factor of everything accumulated so far, 2^(mq_old - mq_new), is an exact power of two: O^T and l are rescaled
  //     without a rounding;
  //   * it differs from 1 only when a row's maximum crosses a multiple of 8 (5.5 natural-log units) -- a handful of times per row instead
  //     of every time any of the wave's 64 rows sees a new maximum (measured with the plain maximum: 100 of 128 tiles of a random-data
  //     launch took the 128-register rescale, 1100-1500 cycles per tile on average; profiles/r06_attn_prefill.md).
  constexpr float MQ = 8.f;
  float mq[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
  const float c2 = a.scale * 1.4426950408889634f;
  // tile maximum tm (raw score units, -inf on a fully masked tile) -> new quantised reference, rescale factor alpha, exponent offset m2
  auto new_max = [&](int qb, float tm, float& alpha, float& m2) __attribute__((always_inline)) {
    const float cand = __builtin_ceilf(tm * (c2 * (1.f / MQ))) * MQ;   // -inf stays -inf
    const float mq_new = fmaxf(mq[qb], cand);
    const float use = (mq_new == -INFINITY) ? 0.f : mq_new;             // fully masked so far: keep everything at zero
    // nothing accumulated yet (mq = -inf: O^T and l are zero) or the reference unchanged: exactly 1
    alpha = (mq[qb] == mq_new || mq[qb] == -INFINITY) ? 1.f : __builtin_ldexpf(1.f, (int)(mq[qb] - use));
    m2 = use;
    mq[qb] = mq_new;
  };
  static_for<NDB * 2>([&](auto B) { o_zero<decltype(B)::value * 16>(); });   // O^T block (db, qb) = a[(db * 2 + qb) * 16 ..]

  // QK^T of one tile: S^T[tb][qb].  The fragment reads run RING fragments ahead of the MFMAs that consume them (left to itself the
  // compiler emits read, lgkmcnt(0), two MFMAs: the whole LDS round trip in front of every pair -- ~5500 instead of ~3300 cycles per tile).
  constexpr int RING = 6;
  auto qk_part = [&](int kbase, f32x16_t (&s)[2][2]) __attribute__((always_inline)) {
    constexpr int NF = 2 * NKS;   // fragment f = (ks, tb) = (f >> 1, f & 1)
    bf16x8_t ring[RING];
#pragma unroll
    for (int f = 0; f < RING; ++f) ring[f] = kfrag(kbase, f & 1, f >> 1);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const bf16x8_t kf = ring[f % RING];
      if (f < 2) {
        mfma32_first(s[f & 1][0], kf, qf[0][f >> 1]);
        mfma32_first(s[f & 1][1], kf, qf[1][f >> 1]);
      } else {
        mfma32_acc(s[f & 1][0], kf, qf[0][f >> 1]);
        mfma32_acc(s[f & 1][1], kf, qf[1][f >> 1]);
      }
      if (f + RING < NF) ring[f % RING] = kfrag(kbase, (f + RING) & 1, (f + RING) >> 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // mask, tile maximum, new running maximum, rescale factor (the score side of softmax_scores<3>, attn_common.h)
  auto max_part = [&](int t0, f32x16_t (&s)[2][2], float (&alpha)[2], float (&m2)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qpos = seg_off + q0 + qb * 32 + q32;   // cache position of this lane's q row
      const int lim = a.causal ? min(qpos, seg_tot - 1) : seg_tot - 1;
      const int lim_min = a.causal ? min(seg_off + q0 + qb * 32, seg_tot - 1) : seg_tot - 1;
      if (t0 + KV_PAGE_TOKENS - 1 > lim_min) {   // wave-uniform: only diagonal / last tiles carry the per-element predicate
        const int dlim = lim - t0 - 4 * hi;   // one subtraction per lane; the 32 compares take instruction constants
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (tb * 32 + (r & 3) + 8 * (r >> 2) > dlim) s[tb][qb][r] = -INFINITY;
      }
      auto sv = [&](int i) { return s[i >> 4][qb][i & 15]; };
      float tmax = max3(sv(0), sv(1), sv(2));
#pragma unroll
      for (int i = 3; i < 31; i += 2) tmax = max3(tmax, sv(i), sv(i + 1));
      tmax = max3(tmax, sv(31), sv(31));
      tmax = xhi_max(tmax);
      new_max(qb, tmax, alpha[qb], m2[qb]);
    }
  };
  // p = exp2(s c2 - m c2), the row sum, the bf16 P^T operand fragments
  auto prob_part = [&](f32x16_t (&s)[2][2], const float (&alpha)[2], const float (&m2)[2], bf16x8_t (&pf)[2][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float sum0 = 0.f, sum1 = 0.f;
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) {
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[tb][qb][r], c2, -m2[qb]));
        if (!LSUM) {
#pragma unroll
          for (int r = 0; r < 16; r += 2) sum0 += p[r], sum1 += p[r + 1];
        }
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          const u32x4_t u = {pack_bf(p[4 * sp + 0], p[4 * sp + 1]), pack_bf(p[4 * sp + 2], p[4 * sp + 3]),
                             pack_bf(p[8 + 4 * sp + 0], p[8 + 4 * sp + 1]), pack_bf(p[8 + 4 * sp + 2], p[8 + 4 * sp + 3])};
          pf[qb][tb * 2 + sp] = as_frag(u);
        }
      }
      if (!LSUM) l[qb] = l[qb] * alpha[qb] + (sum0 + sum1);
    }
  };
  // The rescale of O^T by alpha, per q block.  Once the running maximum has settled alpha is exactly 1 in every lane of the wave: skip
  // the multiplies (x * 1 == x).  The accumulators are no C++ objects, so the branch has no merge to lower.
  constexpr int NB = NDB;
  auto rescale_part = [&](const float (&alpha)[2]) __attribute__((always_inline)) {
    if (__builtin_amdgcn_ballot_w64(alpha[0] != 1.f) != 0) {
      static_for<NB>([&](auto DB) { o_scale<(decltype(DB)::value * 2 + 0) * 16, decltype(DB)::value == 0>(alpha[0]); });
    }
    if (__builtin_amdgcn_ballot_w64(alpha[1] != 1.f) != 0) {
      static_for<NB>([&](auto DB) { o_scale<(decltype(DB)::value * 2 + 1) * 16, decltype(DB)::value == 0>(alpha[1]); });
    }
  };
  // P.V: fragment f = (kstep, db) = (f / NDB, f % NDB); the first RING reads are issued by pv_reads() in front of the exponentials
  constexpr int NFV = 4 * NDB;
  auto pv_reads = [&](int vbase, bf16x8_t (&vring)[RING]) __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < RING; ++f) vring[f] = vfrag(vbase, f % NDB, f / NDB);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto pv_part = [&](int vbase, bf16x8_t (&pf)[2][4], bf16x8_t (&vring)[RING]) __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) valu_settle(pf[0][i], pf[1][i]);
    static_for<NFV>([&](auto F) {
      constexpr int f = decltype(F)::value, db = f % NDB, kstep = f / NDB;
      const bf16x8_t vf = vring[f % RING];
      o_mfma<(db * 2 + 0) * 16>(vf, pf[0][kstep]);
      o_mfma<(db * 2 + 1) * 16>(vf, pf[1][kstep]);
      if (f + RING < NFV) vring[f % RING] = vfrag(vbase, (f + RING) % NDB, (f + RING) / NDB);
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // ---- prologue: tiles 0 and 1 requested, tile 0 landed ----
  dma_tile(ptab[0], 0);
  if (ntiles > 1) dma_tile(ptab[1], STAGE);
  uint64_t pg_next = ptab[__builtin_amdgcn_readfirstlane(min(2, ntiles - 1))];
  if (ntiles > 1) ATTN64_WAIT_VM(NP);
  else ATTN64_WAIT_VM(0);
  ATTN64_BAR();

  if (PIPE == 0) {
    int stage_off = 0, pre_off = 2 * STAGE;   // LDS offsets of tile t's stage and of the stage tile t + 2 goes to
    for (int tile = 0; tile < ntiles; ++tile) {
      if (tile + 2 < ntiles) dma_tile(pg_next, pre_off);
      pg_next = ptab[__builtin_amdgcn_readfirstlane(min(tile + 3, ntiles - 1))];
      if (tile < wtiles) {
        f32x16_t s[2][2];
        float alpha[2], m2[2];
        bf16x8_t pf[2][4];
        bf16x8_t vring[RING];
        qk_part(lk + stage_off, s);
        pv_reads(lv + stage_off, vring);
        mfma_settle(s[0][0], s[0][1], s[1][0], s[1][1]);
        max_part(tile * KV_PAGE_TOKENS, s, alpha, m2);
        rescale_part(alpha);
        prob_part(s, alpha, m2, pf);
        pv_part(lv + stage_off, pf, vring);
      }
      if (tile + 2 < ntiles) ATTN64_WAIT_VM(NP);   // tile t + 1 has landed, t + 2 may still fly
      else ATTN64_WAIT_VM(0);
      ATTN64_BAR();
      pre_off = stage_off;
      stage_off = (stage_off == 2 * STAGE) ? 0 : stage_off + STAGE;
    }
  } else {
    // Software pipeline INSIDE the wave, instruction groups placed by hand (one wave per SIMD: nobody else fills the matrix pipe while this
    // wave runs vector instructions, and an in-order wave only overlaps the two pipes when independent work is interleaved in program order).
    // A tile's vector work is 64 exponentials per lane at a quarter of the plain rate (16 cycles each: half of the ~1900 vector cycles of a
    // tile, against 2048 matrix cycles), so it is spread ONE SCORE PER MFMA GAP over three phases (a score = fma, exp, add, half a cvt_pk:
    // ~26 cycles under a 32-cycle MFMA).  Iteration `t` of the steady state, every MFMA followed by its share and a sched_barrier:
    //   phase A   QK^T(t + 1)  |  scores E .. E + NA of tile t (chunk order = the order P.V's k-steps consume P^T)               |  K reads
    //   phase B   P.V(t)       |  gaps 0 .. SB: the last scores of tile t (k-step c of P.V starts at gap c * 2 NDB: its chunk is complete)
    //                          |  gap M0: the diagonal mask of tile t + 1; gaps M0 .. M0 + 15: its maximum, 2 v_max3 per gap; gap MF: the new
    //                          |  quantised maximum; gaps MF + 1 ..: the first E scores of tile t + 1  |  V^T reads  |  LDS-DMA pieces of tile t + 2
    //   then      rescale by alpha(t + 1) if any lane needs it, vmcnt, barrier.
    // The S^T blocks of tile t + 1 are first read by vector instructions in gap M0 = 5 of phase B, > 25 instructions behind the last QK^T
    // MFMA: the MFMA-result hazard (11 wait states for an 8-pass MFMA) needs no padding there.  Two S^T register sets (loop body = two
    // iterations).
    // (The row sum on the matrix pipe -- one more MFMA per k-step and q block with a fragment of ones as A, 8 MFMAs instead of 64
    // v_add_f32 per tile -- was built and measured: phase A 1270 -> 1180 cycles, phase B 1690 -> 2144 for its 8 more gaps, 0.96 -> 1.06 ms at
    // S = 8192: a gap of phase B costs ~53 cycles whatever it holds, section 3 of profiles/r06_attn_prefill.md.  Not kept.)
    constexpr bool DMA_IN_A = true;                    // the LDS-DMA pieces of tile t + 2 ride in phase A (phase B: 1595 -> 1500 cycles, A + 55)
    constexpr int KG = 2 * NDB;                        // MFMA gaps of one P.V k-step
    constexpr int GA = 4 * NKS, GB = 4 * KG;           // MFMA gaps of the two phases: 32 + 32 (text), 20 + 24 (ViT)
    constexpr int PSTEP = GA / NP;                     // an LDS-DMA piece every PSTEP gaps of phase A: 4 (text: 8 pieces), 3 (ViT: 6)
    static_assert(PSTEP >= 2 && (NP - 1) * PSTEP + 1 < GA, "every piece of tile t + 2 has a gap in phase A");
    constexpr int MAXG = GB >= 32 ? 16 : 8;            // phase B gaps that carry the maximum of tile t + 1: 32 / MAXG v_max3 each
    constexpr int M0 = 5, MF = M0 + MAXG;              // ... gaps [M0, MF); gap MF: the new quantised maximum
    constexpr int SB = MF + 1;                         // phase B gaps [0, SB) carry the last SB scores of tile t, gaps [SB, GB) the first E of tile t + 1
    constexpr int E = GB - SB;                         // 10 (text), 10 (ViT)
    constexpr int NA = 64 - GB;                        // scores of phase A: 32 over 32 gaps (text), 40 over 20 (ViT); phase B gap g = position NA + g
    // A score is a chain of three dependent vector instructions (fma -> exp -> add / cvt_pk); issued back to back each waits out the previous
    // one's latency (measured: ~8 cycles per instruction in a gap of {MFMA, fma, exp, add} where independent instructions issue every ~4).
    // So the chain is itself software-pipelined over the stream of scores: position p issues stage 3 of score p - 2, stage 2 of score
    // p - 1 and stage 1 of score p -- three independent instructions -- and the stream runs on across the phases and across iterations (the
    // last two scores of a phase B finish in the first two positions of the next iteration).  A k-step of P.V therefore finds its P^T
    // fragments complete two positions after its last score: the static_asserts below.
    static_assert(E >= 2 && MF < GB && SB <= 3 * KG - 2 && NA >= 0, "score schedule");
    static_assert(47 - E + 2 - NA < 2 * KG && 31 - E + 2 - NA < KG, "score schedule against the k-step deadlines");
    f32x16_t sA[2][2], sB[2][2];
    float alphac[2], m2c[2];      // of the tile whose probabilities are due
    float alphan[2], m2n[2];      // of the next tile (valid from gap MF of phase B)
    u32x4_t pf4[2][4];            // P^T fragments [qb][k-step]
    float psum[2] = {0.f, 0.f}, phold[2];
    float tfr[2], per[2];         // stage 1 -> 2 and stage 2 -> 3 registers of the score pipeline, by position parity
    int stage_off = 0, nxt_off = STAGE, pre_off = 2 * STAGE;
    // score k (0..63) of a tile, in the order P.V consumes them: chunk c = k >> 4 = k-step c (tokens tb = c >> 1, registers 4 s' .. and
    // 8 + 4 s' .. with s' = c & 1), then q block, then the 8 registers; two consecutive scores make one dword of the fragment.
    // Position p of an iteration (0..63) -> score: p < 64 - E: score E + p of the current tile; else score p - (64 - E) of the next one.
    // (The three stages are asm statements, one instruction each: C++ arithmetic is free to sink to its use -- the fma next to its exp, all
    // adds into one chain behind the phase -- and an empty asm that pins a value costs an s_nop of boundary padding per use: 45 per tile.)
    const float c2v = c2;
    auto s1 = [&](auto P, f32x16_t (&sc)[2][2], f32x16_t (&sn)[2][2]) __attribute__((always_inline)) {
      constexpr int p = decltype(P)::value, nxt = p >= 64 - E, k = nxt ? p - (64 - E) : E + p;
      constexpr int c = k >> 4, qb = (k >> 3) & 1, idx = k & 7, tb = c >> 1, sp = c & 1, r = idx < 4 ? 4 * sp + idx : 8 + 4 * sp + (idx - 4);
      // (locals first: an asm operand inside a generic lambda does not capture by itself)
      float& t = tfr[p & 1];
      const float x = nxt ? sn[tb][qb][r] : sc[tb][qb][r], mm = nxt ? m2n[qb] : m2c[qb], cc = c2v;
      asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(t) : "v"(x), "v"(cc), "v"(mm));
    };
    auto s2 = [&](auto P) __attribute__((always_inline)) {
      constexpr int p = (decltype(P)::value + 64) & 63;
      float& e = per[p & 1];
      const float t = tfr[p & 1];
      asm volatile("v_exp_f32 %0, %1" : "=v"(e) : "v"(t));
    };
    auto s3 = [&](auto P) __attribute__((always_inline)) {
      constexpr int p = (decltype(P)::value + 64) & 63, nxt = p >= 64 - E, k = nxt ? p - (64 - E) : E + p;
      constexpr int c = k >> 4, qb = (k >> 3) & 1, idx = k & 7;
      const float pv = per[p & 1];
      float& ps = psum[qb];
      float& ph = phold[qb];
      if (!LSUM) asm volatile("v_add_f32 %0, %0, %1" : "+v"(ps) : "v"(pv));
      if ((idx & 1) == 0) {
        ph = pv;
      } else {
        uint32_t w;
        const float lo = ph;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(lo), "v"(pv));
        pf4[qb][c][idx >> 1] = w;
      }
    };
    // one position of the stream; FILL = how many of the older stages exist (the prologue starts the stream: 0, 1, then 2)
    auto position = [&](auto P, f32x16_t (&sc)[2][2], f32x16_t (&sn)[2][2]) __attribute__((always_inline)) {
      constexpr int p = decltype(P)::value;
      s3(std::integral_constant<int, p - 2>{});
      s2(std::integral_constant<int, p - 1>{});
      s1(P, sc, sn);
    };
    float tmax[2];
    auto max_slice = [&](auto K, f32x16_t (&sn)[2][2]) __attribute__((always_inline)) {   // K = 0..15: one v_max3 per q block, one statement
      constexpr int k = decltype(K)::value;
      auto sv = [&](int qb, int i) { return sn[i >> 4][qb][i & 15]; };
      float& t0 = tmax[0];
      float& t1 = tmax[1];
      if (k == 0) {
        const float a0 = sv(0, 0), a1 = sv(0, 1), a2 = sv(0, 2), b0 = sv(1, 0), b1 = sv(1, 1), b2 = sv(1, 2);
        asm volatile("v_max3_f32 %0, %2, %3, %4\n\tv_max3_f32 %1, %5, %6, %7" : "=&v"(t0), "=&v"(t1) : "v"(a0), "v"(a1), "v"(a2), "v"(b0), "v"(b1), "v"(b2));
      } else if (k < 15) {
        const float a1 = sv(0, 2 * k + 1), a2 = sv(0, 2 * k + 2), b1 = sv(1, 2 * k + 1), b2 = sv(1, 2 * k + 2);
        asm volatile("v_max3_f32 %0, %0, %2, %3\n\tv_max3_f32 %1, %1, %4, %5" : "+v"(t0), "+v"(t1) : "v"(a1), "v"(a2), "v"(b1), "v"(b2));
      } else {
        const float a1 = sv(0, 31), b1 = sv(1, 31);
        asm volatile("v_max3_f32 %0, %0, %2, %2\n\tv_max3_f32 %1, %1, %3, %3" : "+v"(t0), "+v"(t1) : "v"(a1), "v"(b1));
      }
    };
    auto mask_step = [&](int t0, f32x16_t (&sn)[2][2]) __attribute__((always_inline)) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const int qpos = seg_off + q0 + qb * 32 + q32;
        const int lim = a.causal ? min(qpos, seg_tot - 1) : seg_tot - 1;
        const int lim_min = a.causal ? min(seg_off + q0 + qb * 32, seg_tot - 1) : seg_tot - 1;
        if (t0 + KV_PAGE_TOKENS - 1 > lim_min) {
          const int dlim = lim - t0 - 4 * hi;
#pragma unroll
          for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (tb * 32 + (r & 3) + 8 * (r >> 2) > dlim) sn[tb][qb][r] = -INFINITY;
        }
      }
    };
    auto max_final = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        new_max(qb, xhi_max(tmax[qb]), alphan[qb], m2n[qb]);
      }
    };
    // one DMA piece of tile t + 2 (K pieces first, V^T pieces last: the end-of-iteration wait leaves exactly the V^T pieces in flight)
    __amdgpu_buffer_rsrc_t rk, rv;
    auto dma_rsrc = [&](uint64_t page) __attribute__((always_inline)) {
      rk = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(page + koff), 0, KBYTES, 0x00020000);
      rv = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(page + voffb), 0, VBYTES, 0x00020000);
    };
    auto dma_piece = [&](auto P) __attribute__((always_inline)) {
      constexpr int pc = decltype(P)::value;
      char* dst = smem + pre_off;
      if (pc < KPW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_ptr_t)(dst + (wave + 4 * pc) * 1024), 16, voff[pc % 4], 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr_t)(dst + KBYTES + (wave + 4 * (pc - KPW)) * 1024), 16, voff[(pc - KPW) % 4], 0, 0, 0);
    };
    auto iter_end = [&]() __attribute__((always_inline)) {
      // the next iteration reads K(t + 2) and V^T(t + 1): everything but the newest VPW pieces (V^T of t + 2) has landed
      ATTN64_WAIT_VM(VPW);
      ATTN64_BAR();
      pre_off = stage_off;
      stage_off = nxt_off;
      nxt_off = (nxt_off == 2 * STAGE) ? 0 : nxt_off + STAGE;
    };
    unsigned tr_acc[4] = {0, 0, 0, 0}, tr_prev = 0;
    const bool tr_on = TRACE && blockIdx.x == gridDim.x / 2;
    auto stamp = [&](int k) __attribute__((always_inline)) {
      if (TRACE) {
        if (tr_on) {
          const unsigned now = __builtin_amdgcn_readfirstlane((unsigned)__builtin_readcyclecounter());
          if (k >= 0) tr_acc[k] += now - tr_prev;
          tr_prev = now;
        }
      }
    };
    // MORE: the wave has a tile t + 1 (else only the probabilities and P.V of its last tile, the DMA pieces and the barrier)
    auto iter = [&](auto more_tag, int tile, f32x16_t (&sc)[2][2], f32x16_t (&sn)[2][2]) __attribute__((always_inline)) {
      constexpr bool MORE = decltype(more_tag)::value;
      stamp(-1);
      // tile t + 2 (past the end: the last tile again, into the stage nobody reads any more -- keeps the counted waits uniform)
      dma_rsrc(pg_next);
      pg_next = ptab[__builtin_amdgcn_readfirstlane(min(tile + 3, ntiles - 1))];
      const int kbase = lk + nxt_off, vbase = lv + stage_off;
      // ---- phase A ----
      bf16x8_t ring[RING];
      if (MORE) {
#pragma unroll
        for (int f = 0; f < RING; ++f) ring[f] = kfrag(kbase, f & 1, f >> 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      static_for<GA>([&](auto G) {
        constexpr int g = decltype(G)::value, f = g >> 1, qb = g & 1;
        if (MORE) {
          if (f < 2) mfma32_first(sn[f & 1][qb], ring[f % RING], qf[qb][f >> 1]);
          else mfma32_acc(sn[f & 1][qb], ring[f % RING], qf[qb][f >> 1]);
          if (qb == 1 && f + RING < 2 * NKS) ring[f % RING] = kfrag(kbase, (f + RING) & 1, (f + RING) >> 1);
        }
        constexpr int p0 = (g * NA) / GA, p1 = ((g + 1) * NA) / GA;   // this gap's positions of the score stream
        static_for<p1 - p0>([&](auto J) { position(std::integral_constant<int, p0 + decltype(J)::value>{}, sc, sn); });
        if (DMA_IN_A && g % PSTEP == 1 && g / PSTEP < NP) dma_piece(std::integral_constant<int, g / PSTEP>{});
        __builtin_amdgcn_sched_barrier(0);
      });
      stamp(0);
      // ---- phase B ----
      bf16x8_t vring[RING];
#pragma unroll
      for (int f = 0; f < RING; ++f) vring[f] = vfrag(vbase, f % NDB, f / NDB);
      __builtin_amdgcn_sched_barrier(0);
      static_for<GB>([&](auto G) {
        constexpr int g = decltype(G)::value, kstep = g / KG, j = g % KG, qb = j & 1, db = j >> 1, f = kstep * NDB + db;
        if (j == 0) valu_settle4(pf4[0][kstep], pf4[1][kstep]);   // a k-step's P^T fragments: complete, two wait states behind their last write
        o_mfma<(db * 2 + qb) * 16>(vring[f % RING], as_frag(pf4[qb][kstep]));
        if (qb == 1 && f + RING < NFV) vring[f % RING] = vfrag(vbase, (f + RING) % NDB, (f + RING) / NDB);
        if (g == SB + 2) {   // every score of tile t has been summed (stage 3 lags two positions)
          if (!LSUM) {
            l[0] = l[0] * alphac[0] + psum[0];
            l[1] = l[1] * alphac[1] + psum[1];
            psum[0] = 0.f, psum[1] = 0.f;
          }
        }
        if (MORE) {
          if (g == M0) mask_step((tile + 1) * KV_PAGE_TOKENS, sn);
          if (g >= M0 && g < MF) static_for<16 / MAXG>([&](auto J) { max_slice(std::integral_constant<int, (g - M0) * (16 / MAXG) + decltype(J)::value>{}, sn); });
          if (g == MF) max_final();
        }
        if (g < SB || MORE) {
          position(std::integral_constant<int, NA + g>{}, sc, sn);
        } else if (g == SB) {        // the wave's last tile: the stream ends here, its last two scores drain
          s3(std::integral_constant<int, NA + g - 2>{});
          s2(std::integral_constant<int, NA + g - 1>{});
        } else if (g == SB + 1) {
          s3(std::integral_constant<int, NA + g - 2>{});
        }
        if (!DMA_IN_A && g % (GB / NP) == 1 && g / (GB / NP) < NP) dma_piece(std::integral_constant<int, g / (GB / NP)>{});
        __builtin_amdgcn_sched_barrier(0);
      });
      stamp(1);
      if (MORE) {
        rescale_part(alphan);   // after P.V(t); opens with the MFMA-result wait states
        alphac[0] = alphan[0], alphac[1] = alphan[1], m2c[0] = m2n[0], m2c[1] = m2n[1];
      }
      stamp(2);
      iter_end();
      stamp(3);
    };
    auto idle = [&](int tile) __attribute__((always_inline)) {   // a wave past its own tiles: its share of the staging, the barrier
      dma_rsrc(pg_next);
      pg_next = ptab[__builtin_amdgcn_readfirstlane(min(tile + 3, ntiles - 1))];
      static_for<NP>([&](auto P) { dma_piece(P); });
      iter_end();
    };
    // prologue of the pipeline: QK^T and the maximum of tile 0 in program order (tile 0 landed: the wait + barrier above);
    // iteration 0 then reads tile 1's K: landed once only tile 1's V^T pieces may still fly
    if (wtiles > 0) {
      qk_part(lk, sA);
      mfma_settle(sA[0][0], sA[0][1], sA[1][0], sA[1][1]);
      max_part(0, sA, alphac, m2c);   // alpha = 1 on an empty accumulator: nothing to rescale
      // the scores the steady state has under way at the end of a phase B: the first E of the tile, the last two of them two / one stage
      // short of complete (as positions 64 - E .. 63 of a virtual previous iteration whose "next" tile is tile 0)
      m2n[0] = m2c[0], m2n[1] = m2c[1];
      static_for<E>([&](auto J) {
        constexpr int j = decltype(J)::value, p = 64 - E + j;
        if (j >= 2) s3(std::integral_constant<int, p - 2>{});
        if (j >= 1) s2(std::integral_constant<int, p - 1>{});
        s1(std::integral_constant<int, p>{}, sA, sA);
      });
    }
    if (ntiles > 1) {
      ATTN64_WAIT_VM(VPW);
      ATTN64_BAR();
    }
    int tile = 0;
    bool odd = false;
    while (tile + 1 < wtiles) {
      iter(std::true_type{}, tile, sA, sB);
      ++tile;
      if (!(tile + 1 < wtiles)) {
        odd = true;
        break;
      }
      iter(std::true_type{}, tile, sB, sA);
      ++tile;
    }
    if (tile < wtiles) {
      if (odd) iter(std::false_type{}, tile, sB, sA);
      else iter(std::false_type{}, tile, sA, sB);
      ++tile;
    }
    for (; tile < ntiles; ++tile) idle(tile);
    if (TRACE) {
      if (tr_on && lane == 0) {
        for (int k = 0; k < 4; ++k) trace[wave * 8 + k] = tr_acc[k];
        trace[wave * 8 + 7] = (unsigned long long)wtiles;
      }
    }
    ATTN64_WAIT_VM(0);   // the re-requested last tile: nothing may still be writing the LDS the epilogue reuses
    ATTN64_BAR();
  }

  // ---- epilogue: 1 / row sum, bf16, whole rows through this wave's slice of the (now free) staging LDS, 16 B per lane ----
  constexpr int EPITCH = DV * 2 + 16;   // bytes per LDS row
  char* wb = smem + wave * (64 * EPITCH);
  float inv[2];
  {
    // LSUM: output row 72 = block db 2, crow(4, 0) = 8 -> register 4 of the column's hi = 0 lane
    float x0[16], x1[16];
    if (LSUM) {
      o_read<((NDB - 1) * 2 + 0) * 16, true>(x0);
      o_read<((NDB - 1) * 2 + 1) * 16, false>(x1);
    }
    inv[0] = 1.0f / (LSUM ? xhi_low(x0[4]) : xhi_sum(l[0]));
    inv[1] = 1.0f / (LSUM ? xhi_low(x1[4]) : xhi_sum(l[1]));
  }
  static_for<NDB * 2>([&](auto B) {
    constexpr int blk = decltype(B)::value, db = blk >> 1, qb = blk & 1;
    float x[16];
    o_read<blk * 16, blk == 0>(x);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int dim = db * 32 + 8 * j + 4 * hi;
      if (dim < DV) {
        uint2 w;
        w.x = pack_bf(x[4 * j + 0] * inv[qb], x[4 * j + 1] * inv[qb]);
        w.y = pack_bf(x[4 * j + 2] * inv[qb], x[4 * j + 3] * inv[qb]);
        *reinterpret_cast<uint2*>(wb + (qb * 32 + q32) * EPITCH + dim * 2) = w;
      }
    }
  });
  // (same wave wrote and reads: no barrier, the LDS queue is in order)
  const int chunk = lane & 15;
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int row = it * 4 + (lane >> 4), qr = q0 + row;
    if (qr < seg_rows && chunk * 8 < a.d) {   // a.d = real head dim of the output rows: a multiple of 8 on this path (launcher)
      const u32x4_t v = *reinterpret_cast<const u32x4_t*>(wb + row * EPITCH + chunk * 16);
      *reinterpret_cast<u32x4_t*>((bf16_t*)a.o + ((int64_t)(seg_row0 + qr) * a.nh + head) * a.d + chunk * 8) = v;
    }
  }
#undef ATTN64_WAIT_VM
#undef ATTN64_BAR
}

template <typename K>
void set_max_lds(K kernel, size_t lds) {
  // > 64 KiB of dynamic LDS needs the attribute, per device; set first, publish afterwards, under a mutex (a second thread must not
  // see "done" before the attribute is in place)
  static std::mutex mu;
  static uint64_t done[4] = {0, 0, 0, 0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> g(mu);
  if (dev >= 0 && dev < 256 && (done[dev >> 6] >> (dev & 63) & 1)) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (dev >= 0 && dev < 256) done[dev >> 6] |= 1ull << (dev & 63);
}

}  // namespace

// The 64-rows-per-wave form of launch_attn_prefill (kernels_attn.hip decides when): f32 score chain only, head_dim 128 or the ViT's 72
// (96 / 80 padded pages), output rows in multiples of 8 dims at 16-byte aligned addresses.  Returns false when the shape is not its own.
bool launch_attn_prefill64(const AttnPrefillArgs& a_in, hipStream_t st, int pipe) {
  AttnPrefillArgs a = a_in;
  if (!(a.d == 128 || a.d == 72) || a.d % 8 != 0 || ((uintptr_t)a.o & 15) != 0 || !(a.scale > 0.f)) return false;
  if (a.S2 > 0 && !(a.causal && a.kvh % 8 == 0 && a.nh % a.kvh == 0)) {
    // two segments where the one-launch block order does not apply: two launches of THIS form (not the 16-row kernel: a context-parallel
    // rank must run the form the un-sharded prompt runs)
    AttnPrefillArgs b = a;
    b.S2 = 0;
    if (!launch_attn_prefill64(b, st, pipe)) return false;
    b.q = (const char*)a.q + (int64_t)a.S * (a.q_ld ? a.q_ld : (int64_t)a.nh * (a.d == 72 ? 96 : a.d)) * 2;
    b.o = (char*)a.o + (int64_t)a.S * a.nh * a.d * 2;
    b.S = a.S2, b.kv_offset = a.kv_offset2, b.kv_total = a.kv_total2;
    return launch_attn_prefill64(b, st, pipe);
  }
  const bool xcd_order = a.kvh % 8 == 0 && a.nh % a.kvh == 0;
  const int nqb = (a.S + ATTN64_ROWS - 1) / ATTN64_ROWS + (a.S2 + ATTN64_ROWS - 1) / ATTN64_ROWS;
  a.nqb = xcd_order ? nqb : 0;
  a.epi_rows = 1;
  const dim3 grid = a.nqb ? dim3(nqb * a.nh) : dim3(nqb, a.nh), block(256);
#define ATTN64_LAUNCH(DQK_, DV_, LSUM_)                                                                     \
  do {                                                                                                      \
    constexpr size_t lds = 3 * (KV_PAGE_TOKENS * DQK_ * 2 + ((DV_ + 31) / 32) * 4096);                      \
    if (pipe) {                                                                                             \
      set_max_lds(attn_prefill64_kernel<DQK_, DV_, LSUM_, 1>, lds);                                         \
      hipLaunchKernelGGL((attn_prefill64_kernel<DQK_, DV_, LSUM_, 1>), grid, block, lds, st, a, nullptr);            \
    } else {                                                                                                \
      set_max_lds(attn_prefill64_kernel<DQK_, DV_, LSUM_, 0>, lds);                                         \
      hipLaunchKernelGGL((attn_prefill64_kernel<DQK_, DV_, LSUM_, 0>), grid, block, lds, st, a, nullptr);            \
    }                                                                                                       \
  } while (0)
#ifdef AHA_DEBUG_KERNELS
  static const bool tr = [] { const char* e = getenv("AHA_ATTN64_TRACE"); return e && atoi(e) != 0; }();
  if (tr && pipe && (a.d == 128 || a.v_ones_row)) {
    static unsigned long long* d_tr = nullptr;
    if (!d_tr) (void)hipMalloc((void**)&d_tr, 32 * 8);
    (void)hipMemsetAsync(d_tr, 0, 32 * 8, st);
    if (a.d == 128) {
      constexpr size_t lds = 3 * (KV_PAGE_TOKENS * 128 * 2 + 4 * 4096);
      set_max_lds(attn_prefill64_kernel<128, 128, false, 1, true>, lds);
      hipLaunchKernelGGL((attn_prefill64_kernel<128, 128, false, 1, true>), grid, block, lds, st, a, d_tr);
    } else {
      constexpr size_t lds = 3 * (KV_PAGE_TOKENS * 96 * 2 + 3 * 4096);
      set_max_lds(attn_prefill64_kernel<96, 80, true, 1, true>, lds);
      hipLaunchKernelGGL((attn_prefill64_kernel<96, 80, true, 1, true>), grid, block, lds, st, a, d_tr);
    }
    unsigned long long h[32];
    (void)hipMemcpyAsync(h, d_tr, sizeof(h), hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    for (int w = 0; w < 4; ++w)
      fprintf(stderr, "[attn64 trace] d %d wave %d, %llu tiles, cycles per tile: phase A %.0f, phase B %.0f, rescale %.0f, wait + barrier %.0f\n", a.d, w, h[w * 8 + 7],
              (double)h[w * 8] / (double)h[w * 8 + 7], (double)h[w * 8 + 1] / (double)h[w * 8 + 7], (double)h[w * 8 + 2] / (double)h[w * 8 + 7],
              (double)h[w * 8 + 3] / (double)h[w * 8 + 7]);
    return true;
  }
#endif
  if (a.d == 128) ATTN64_LAUNCH(128, 128, false);
  else if (a.v_ones_row) ATTN64_LAUNCH(96, 80, true);
  else ATTN64_LAUNCH(96, 80, false);
#undef ATTN64_LAUNCH
  return true;
}

}  // namespace aha
