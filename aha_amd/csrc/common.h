// Shared device helpers for the gfx950 kernels (wave = 64 lanes; bf16 storage, f32 arithmetic).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Shared device code (gemv_body.h, attn_decode_body.h) is compiled into several kernels that must stay bit-identical
// (tests/test_model_gpu.py::test_decode_fused_launches_equal_launch_per_op).  clang fuses mul+add on its own, and an
// expression with TWO products (`a * b + c * d`) can be fused either way, decided per compilation context: such
// expressions are written with an explicit fmaf in the shared bodies.  (A blanket `#pragma clang fp contract(off)` costs
// 13% of Qwen3-0.6B decode.)

namespace aha {

typedef uint16_t bf16_t;  // raw bf16 bit pattern in HBM / LDS

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;  // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // MFMA 16x16 C/D fragment
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even f32 -> bf16 (what a Candle op does when it materialises a bf16 tensor).  The cast lowers to
// the gfx950 hardware conversion v_cvt_pk_bf16_f32 (RNE), 1 VALU op instead of ~6 integer ops: the softmax of the
// attention kernels is VALU-bound, so this matters.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ bf16_t f2bf(float f) {
  const __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
// value of f after a round trip through bf16
// (through the bit pattern on purpose: `(float)(__bf16)f` may be elided under clang's bf16 excess-precision rules)
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

// low / high bf16 of a packed dword as f32
__device__ __forceinline__ float lo_bf(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_bf(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack_bf(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  const bf16x2_t p = __builtin_convertvector(v, bf16x2_t);
  return __builtin_bit_cast(uint32_t, p);
}

// Butterfly reductions over the 64 lanes (partners xor 32, 16, 8, 4, 2, 1 in that order; every lane ends with the total).
// gfx950 form without the LDS crossbar (ds_bpermute): v_permlane32_swap / v_permlane16_swap for the two cross-row steps,
// DPP row rotations inside a 16-lane row -- after the xor-8 step lanes i and i^8 hold the same value, so a rotation by 4
// delivers exactly the xor-4 partner's value, and so on.  Same pairings and order as the __shfl_xor loop => same bits.
__device__ __forceinline__ float wave_sum(float v) {
  {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x124 /* row_ror:4 */, 0xf, 0xf, false));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x122 /* row_ror:2 */, 0xf, 0xf, false));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x121 /* row_ror:1 */, 0xf, 0xf, false));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
  {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x128, 0xf, 0xf, false)));
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x124, 0xf, 0xf, false)));
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x122, 0xf, 0xf, false)));
  v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x121, 0xf, 0xf, false)));
  return v;
}

// streamed-once 16-byte load (weights in decode: one CU reads each line once -> non-temporal)
__device__ __forceinline__ u32x4_t ld_nt16(const void* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
}
__device__ __forceinline__ u32x4_t ld16(const void* p) { return *reinterpret_cast<const u32x4_t*>(p); }
// The same for an address that did not come from a kernel argument (KV page pointers are 64-bit integers read from the page
// table): without the explicit global address space the compiler emits flat_load, which it cannot count (flat may complete
// out of order with LDS traffic), so every wait becomes vmcnt(0) lgkmcnt(0) and a software pipeline collapses.
typedef const __attribute__((address_space(1))) u32x4_t* gptr16_t;
__device__ __forceinline__ u32x4_t ld_nt16_global(uint64_t addr) {
  return __builtin_nontemporal_load(reinterpret_cast<gptr16_t>(addr));
}
__device__ __forceinline__ u32x4_t ld16_global(uint64_t addr) { return *reinterpret_cast<gptr16_t>(addr); }  // cached (re-read by other blocks)

// ---- activation traffic inside the persistent decode kernel --------------------------------------------------------
// Blocks on different XCDs exchange activations between grid barriers.  The 8 XCD L2s are not coherent with each other
// for ordinary loads/stores, and agent-scope fences (buffer_wbl2 / buffer_inv) cost tens of microseconds per barrier
// when 2048 waves issue them.  Instead every activation access that crosses a barrier is an agent-scope relaxed atomic
// (sc1 load / write-through store): the few KB per phase bypass the non-coherent caches, the weight stream is untouched.
// COH = false (stand-alone kernels): plain accesses, identical code to before.
template <bool COH>
__device__ __forceinline__ u32x4_t act_ld16(const void* p) {
  if (!COH) return ld16(p);
  const uint64_t* q = reinterpret_cast<const uint64_t*>(p);
  const uint64_t a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint64_t b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return u32x4_t{(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
}
template <bool COH>
__device__ __forceinline__ float4 act_ldf4(const float* p) {
  if (!COH) return *reinterpret_cast<const float4*>(p);
  const u32x4_t v = act_ld16<true>(p);
  return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}
template <bool COH>
__device__ __forceinline__ float2 act_ldf2(const float* p) {
  if (!COH) return *reinterpret_cast<const float2*>(p);
  const uint64_t a = __hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float2(__uint_as_float((uint32_t)a), __uint_as_float((uint32_t)(a >> 32)));
}
template <bool COH>
__device__ __forceinline__ float act_ldf(const float* p) {
  if (!COH) return *p;
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool COH>
__device__ __forceinline__ bf16_t act_ld_bf(const bf16_t* p) {
  if (!COH) return *p;
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool COH>
__device__ __forceinline__ void act_st_bf(bf16_t* p, bf16_t v) {
  if (!COH) *p = v;
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool COH>
__device__ __forceinline__ void act_stf(float* p, float v) {
  if (!COH) *p = v;
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int KV_PAGE_TOKENS = 64;  // tokens per KV page

// Slot of token t (0..63) inside a V page.  Per dim the 64 token slots of a page are permuted
// so that the 8 tokens one MFMA lane group needs for the P.V step (two 4-token runs that come out of the S^T = K.Q^T
// accumulator fragments of token sub-tiles 2kk and 2kk+1) are 16 contiguous bytes: t = kk*32 + sub1*16 + G*4 + j
// lives at slot kk*32 + G*8 + sub1*4 + j.
__host__ __device__ __forceinline__ int v_slot(int t) {
  return (t & 32) | (((t >> 2) & 3) << 3) | (((t >> 4) & 1) << 2) | (t & 3);
}

// ---- KV page layout: FRAGMENT-MAJOR ------------------------------------------------------------------------------------------
// One kv head's share of a page is stored as the MFMA operand fragments themselves, each a contiguous 1 KB block whose 16-byte
// piece l is exactly what lane l (G = l >> 4, c = l & 15) of a wave feeds v_mfma_f32_16x16x32_bf16:
//   K block (4 * KS fragments, KS = padded head dim / 32): fragment (sub, k4) = tokens sub*16.., dims k4*32..;
//       piece l = dims k4*32 + G*8 .. +8 of token sub*16 + c
//   V block (2 * DS fragments, DS = padded head dim / 16): fragment (ds, kk) = dims ds*16.., token slots kk*32..;
//       piece l = slots kk*32 + G*8 .. +8 (v_slot order) of dim ds*16 + c
// A wave-wide 16-byte-per-lane load of a fragment is 1 KB contiguous = 8 whole 128-byte lines.  With token-major K rows /
// dim-major V rows the same load touched 16 rows x 64 B -- half lines, twice the address work per byte in the CU's memory
// pipeline -- and the decode attention kernel streamed at 4.5 TB/s instead of 5.5 (131 k context, measured A/B).
// Element index (in bf16 elements from the head's block start):
__host__ __device__ __forceinline__ int kpage_elem(int t, int e, int KS) {
  return ((((t >> 4) * KS + (e >> 5)) * 64 + ((e >> 3) & 3) * 16 + (t & 15)) << 3) + (e & 7);
}
__host__ __device__ __forceinline__ int vpage_elem(int t, int e) {
  const int s = v_slot(t);
  return ((((e >> 4) * 2 + (s >> 5)) * 64 + ((s >> 3) & 3) * 16 + (e & 15)) << 3) + (s & 7);
}

}  // namespace aha
