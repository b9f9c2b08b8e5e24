/* Plain-C harness over include/aha_hip.h -- what a non-Python host (the reference's Rust shim)
 * does with the library: load a checkpoint directory, run generate_generic's greedy loop two ways (host-driven
 * forward_initial / forward_step, generate.rs:115-159; and the device-resident aha_hip_decode_greedy), print the tokens.
 *
 *   gcc -O2 -Iinclude examples/c_harness.c -o examples/c_harness -Laha_amd/csrc -laha_hip -Wl,-rpath,$PWD/aha_amd/csrc -lm
 *   examples/c_harness <checkpoint_dir> <max_tokens> <id0> <id1> ...
 * Output: one line "host: t0 t1 ..." and one line "device: t0 t1 ..." (they must be equal), then "candidates: i0 i1 i2 i3"
 * (the 4 largest penalised logits of the prompt's last position: aha_hip_sample_candidates) and "resampled: n" (length of a
 * 44.1 kHz stereo second brought to 16 kHz mono by aha_hip_audio_resample); exit code 0.
 * No torch, no HIP headers: only the C ABI. */
#include <stdint.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "aha_hip.h"

#define CHECK(call)                                                              \
  do {                                                                           \
    int _rc = (call);                                                            \
    if (_rc < 0) {                                                               \
      fprintf(stderr, "%s failed (%d): %s\n", #call, _rc, aha_hip_last_error()); \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

static int is_stop(const uint32_t* stop, int n, uint32_t t) {
  for (int i = 0; i < n; ++i)
    if (stop[i] == t) return 1;
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: %s <checkpoint_dir> <max_tokens> <id0> [id1 ...]\n", argv[0]);
    return 2;
  }
  const char* dir = argv[1];
  const int max_tokens = atoi(argv[2]);
  const size_t n_ids = (size_t)(argc - 3);
  uint32_t* ids = (uint32_t*)malloc(n_ids * sizeof(uint32_t));
  for (size_t i = 0; i < n_ids; ++i) ids[i] = (uint32_t)strtoul(argv[3 + i], NULL, 10);

  aha_model_desc desc;
  CHECK(aha_hip_config_parse(dir, &desc));
  fprintf(stderr, "arch %d, %d layers, hidden %d, vocab %d, %d stop tokens\n", desc.arch, desc.num_hidden_layers,
          desc.hidden_size, desc.vocab_size, desc.n_stop_tokens);

  aha_ctx* ctx = NULL;
  aha_model* model = NULL;
  CHECK(aha_hip_init(0, &ctx));
  CHECK(aha_hip_model_load(ctx, dir, 0, &model));
  uint32_t stop[8];
  const int n_stop = aha_hip_stop_token_ids(model, stop, 8);
  if (n_stop < 0) return 1;

  uint32_t* host_toks = (uint32_t*)malloc((size_t)max_tokens * sizeof(uint32_t));
  uint32_t* dev_toks = (uint32_t*)malloc((size_t)max_tokens * sizeof(uint32_t));
  float* logits = (float*)malloc((size_t)desc.vocab_size * sizeof(float));

  /* (1) the reference's loop: logits come back every step, argmax on the host (first maximal index) */
  int n_host = 0;
  size_t offset = 0;
  CHECK(aha_hip_forward_initial(model, ids, n_ids, 0, NULL, logits, NULL));
  for (;;) {
    uint32_t best = 0;
    for (int v = 1; v < desc.vocab_size; ++v)
      if (logits[v] > logits[best]) best = (uint32_t)v;
    host_toks[n_host++] = best;
    if (n_host >= max_tokens || is_stop(stop, n_stop, best)) break;
    offset = n_ids + (size_t)n_host - 1;
    CHECK(aha_hip_forward_step(model, best, offset, logits, NULL));
  }
  CHECK(aha_hip_clear_cache(model));

  /* (2) device-resident greedy loop: only token ids cross PCIe */
  uint32_t first = 0;
  CHECK(aha_hip_forward_initial(model, ids, n_ids, 0, NULL, NULL, &first));
  int n_dev = 0;
  dev_toks[n_dev++] = first;
  if (max_tokens > 1 && !is_stop(stop, n_stop, first)) {
    const int got = aha_hip_decode_greedy(model, first, n_ids, (size_t)(max_tokens - 1), dev_toks + 1);
    if (got < 0) {
      fprintf(stderr, "decode_greedy failed: %s\n", aha_hip_last_error());
      return 1;
    }
    n_dev += got;
  }

  printf("host:");
  for (int i = 0; i < n_host; ++i) printf(" %u", host_toks[i]);
  printf("\ndevice:");
  for (int i = 0; i < n_dev; ++i) printf(" %u", dev_toks[i]);
  printf("\n");

  /* (3) the sampled path: repeat penalty over the generated tokens + top-4 candidates of the prompt's logits, all on the device */
  CHECK(aha_hip_clear_cache(model));
  CHECK(aha_hip_forward_initial(model, ids, n_ids, 0, NULL, NULL, NULL));
  float cand_val[4], cmax = 0.f, csum = 0.f;
  uint32_t cand_idx[4];
  CHECK(aha_hip_sample_candidates(model, host_toks, (size_t)n_host, 1.1f, 0.6f, 4, cand_val, cand_idx, &cmax, &csum));
  printf("candidates: %u %u %u %u\n", cand_idx[0], cand_idx[1], cand_idx[2], cand_idx[3]);
  if (!(cand_val[0] >= cand_val[1] && cand_val[1] >= cand_val[2] && cand_val[2] >= cand_val[3] && cmax == cand_val[0] && csum >= 1.0f)) {
    fprintf(stderr, "candidate ordering / normaliser inconsistent\n");
    return 1;
  }

  /* (3b) ... and the draw: candle's softmax values of the four candidates, then eight draws of the reference's default-seed stream
   * (LogitsProcessor::from_sampling(299792458, ..), common/generate.rs:408; rand 0.9.2 StdRng + WeightedIndex<f32> behind the ABI) */
  {
    float prs[4];
    const float inv_t = (float)(1.0 / 0.6);
    for (int j = 0; j < 4; ++j) prs[j] = expf((cand_val[j] - cmax) * inv_t) / csum;
    aha_rng* rng = NULL;
    CHECK(aha_hip_rng_create(299792458ull, &rng));
    printf("weights: %a %a %a %a\ndraws:", prs[0], prs[1], prs[2], prs[3]);
    for (int i = 0; i < 8; ++i) {
      uint32_t pos = 0;
      CHECK(aha_hip_rng_weighted_index(rng, prs, 4, &pos));
      printf(" %u", cand_idx[pos]);
    }
    printf("\n");
    aha_hip_rng_destroy(rng);
  }

  /* (4) audio pre-processing: one second of 44.1 kHz stereo -> 16 kHz mono */
  {
    const int frames = 44100;
    float* pcm = (float*)calloc((size_t)frames * 2, sizeof(float));
    for (int i = 0; i < frames; ++i) pcm[2 * i] = pcm[2 * i + 1] = (float)((i % 100) - 50) / 100.0f;
    const int64_t n_out = aha_hip_audio_resample(ctx, pcm, frames, 2, 44100, 16000, NULL, 0);
    float* res = (float*)malloc((size_t)(n_out > 0 ? n_out : 1) * sizeof(float));
    const int64_t n_got = aha_hip_audio_resample(ctx, pcm, frames, 2, 44100, 16000, res, n_out);
    if (n_out != 16000 || n_got != n_out) {
      fprintf(stderr, "audio_resample: %lld / %lld samples: %s\n", (long long)n_out, (long long)n_got, aha_hip_last_error());
      return 1;
    }
    printf("resampled: %lld\n", (long long)n_got);
    free(pcm); free(res);
  }

  aha_hip_model_destroy(model);
  aha_hip_shutdown(ctx);
  free(ids); free(host_toks); free(dev_toks); free(logits);
  return 0;
}
