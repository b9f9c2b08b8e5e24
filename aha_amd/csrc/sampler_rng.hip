// The random draw of the reference's non-greedy sampler, host code behind the C ABI.
//
// Reference call chain: sample_and_push (/root/reference/src/models/common/generate.rs:70-86) -> LogitsProcessor::sample ->
// sample_multinomial on the processor built by get_logit_processor (src/models/common/sample.rs:7-37: from_sampling(seed, ..) /
// new(seed, ..); seed 299792458 in common/generate.rs:408,452 and 34562 in qwen3_asr/generate.rs:134).  The processor is
// candle-transformers 0.9.2, which depends on rand 0.9.2 (Cargo.lock:590-606), so the stream is
//   StdRng::seed_from_u64      rand_core 0.9.5: the u64 is expanded to 32 seed bytes by a PCG32 (XSH-RR) stream
//   StdRng = ChaCha12Rng       rand_chacha 0.9.0: 12 rounds, 64-bit block counter (words 12-13) from 0, stream id 0, a buffer of
//                              four consecutive blocks handed out word by word
//   WeightedIndex<f32>         rand 0.9.2: running f32 sums, UniformFloat<f32>::new(0, total): value = ((u32 >> 9) as mantissa of
//                              [1, 2)) - 1, times scale (total, lowered by ulps until the largest sample stays < total);
//                              index = number of running sums <= the drawn value.
// None of these crates is on disk (PARITY UNPINNED, [unverified] against the crates): restated from their published algorithms,
// checked against the RFC 7539 block test vector (same function, 20 rounds) and against the independent Python restatement
// oracle/rand_stdrng.py (tests/test_sampling_cpu.py).  Plain C++: no device code, no HIP calls.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <new>
#include <vector>

#include "../../include/aha_hip.h"
#include "model.h"

struct aha_rng {
  uint32_t key[8];
  uint64_t counter;
  uint32_t buf[64];
  int index;
};

namespace {

inline uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

void chacha_block(const uint32_t in[16], int rounds, uint32_t out[16]) {
  uint32_t x[16];
  memcpy(x, in, sizeof(x));
#define AHA_QR(a, b, c, d)                      \
  x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);   \
  x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);   \
  x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);    \
  x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
  for (int r = 0; r < rounds / 2; ++r) {
    AHA_QR(0, 4, 8, 12) AHA_QR(1, 5, 9, 13) AHA_QR(2, 6, 10, 14) AHA_QR(3, 7, 11, 15)
    AHA_QR(0, 5, 10, 15) AHA_QR(1, 6, 11, 12) AHA_QR(2, 7, 8, 13) AHA_QR(3, 4, 9, 14)
  }
#undef AHA_QR
  for (int i = 0; i < 16; ++i) out[i] = x[i] + in[i];
}

void refill(aha_rng* r) {
  for (int b = 0; b < 4; ++b) {
    uint32_t st[16] = {0x61707865u, 0x3320646Eu, 0x79622D32u, 0x6B206574u};
    memcpy(st + 4, r->key, 32);
    st[12] = (uint32_t)r->counter;
    st[13] = (uint32_t)(r->counter >> 32);
    st[14] = 0;
    st[15] = 0;
    chacha_block(st, 12, r->buf + 16 * b);
    r->counter += 1;
  }
  r->index = 0;
}

inline uint32_t next_u32(aha_rng* r) {
  if (r->index >= 64) refill(r);
  return r->buf[r->index++];
}

inline float bits_f32(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
inline uint32_t f32_bits(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }

}  // namespace

extern "C" {

int aha_hip_rng_create(uint64_t seed, aha_rng** out) {
  if (!out) return AHA_ERR_INVALID;
  aha_rng* r = new (std::nothrow) aha_rng();
  if (!r) return AHA_ERR_OOM;
  // seed_from_u64: PCG32 stream, state advanced before every 4-byte output
  const uint64_t MUL = 6364136223846793005ull, INC = 11634580027462260723ull;
  uint64_t state = seed;
  for (int i = 0; i < 8; ++i) {
    state = state * MUL + INC;
    const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
    const uint32_t rot = (uint32_t)(state >> 59);
    r->key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));   // to_le_bytes -> little-endian key word: the value itself
  }
  r->counter = 0;
  r->index = 64;
  *out = r;
  return AHA_OK;
}

void aha_hip_rng_destroy(aha_rng* r) { delete r; }

uint32_t aha_hip_rng_next_u32(aha_rng* r) { return r ? next_u32(r) : 0u; }

// Debug / test hook: the ChaCha block function itself on a caller-supplied 16-word state (RFC 7539 test vectors use 20 rounds).
int aha_hip_debug_chacha_block(const uint32_t* state16, int rounds, uint32_t* out16) {
  if (!state16 || !out16 || rounds <= 0 || (rounds & 1)) return AHA_ERR_INVALID;
  chacha_block(state16, rounds, out16);
  return AHA_OK;
}

int aha_hip_rng_weighted_index(aha_rng* r, const float* weights, size_t n, uint32_t* index_out) {
  if (!r || !weights || n == 0 || !index_out) return AHA_ERR_INVALID;
  // WeightedIndex::new -- every weight >= 0 (NaN fails the comparison), running sums of all but the last, total > 0 and finite
  if (!(weights[0] >= 0.0f)) { aha::set_error("WeightedIndex: invalid weight"); return AHA_ERR_INVALID; }
  std::vector<float> cum;
  cum.reserve(n - 1);
  volatile float total = weights[0];   // (volatile: every partial sum is rounded to f32, whatever the host compiler's float mode)
  for (size_t i = 1; i < n; ++i) {
    if (!(weights[i] >= 0.0f)) { aha::set_error("WeightedIndex: invalid weight"); return AHA_ERR_INVALID; }
    cum.push_back((float)total);
    total = total + weights[i];
  }
  const float tot = total;
  if (tot == 0.0f) { aha::set_error("WeightedIndex: all weights are zero"); return AHA_ERR_INVALID; }
  if (!isfinite(tot)) { aha::set_error("WeightedIndex: the weights overflow"); return AHA_ERR_INVALID; }
  // UniformFloat<f32>::new(0, tot)
  const float max_rand = 1.0f - 1.1920928955078125e-07f;   // 1 - 2^-23
  volatile float scale = tot;
  for (;;) {
    volatile float top = scale * max_rand;   // + low (0)
    if (!(top >= tot)) break;
    scale = bits_f32(f32_bits(scale) - 1u);
  }
  // sample: value in [1, 2) from 23 random bits, minus 1, times scale (+ 0)
  const float v12 = bits_f32((next_u32(r) >> 9) | 0x3F800000u);
  volatile float v01 = v12 - 1.0f;
  volatile float chosen = v01 * scale;
  const float ch = chosen;
  // partition_point(|w| w <= chosen): the running sums are non-decreasing
  size_t lo = 0, hi = cum.size();
  while (lo < hi) {
    const size_t mid = lo + (hi - lo) / 2;
    if (cum[mid] <= ch) lo = mid + 1; else hi = mid;
  }
  *index_out = (uint32_t)lo;
  return AHA_OK;
}

}  // extern "C"
