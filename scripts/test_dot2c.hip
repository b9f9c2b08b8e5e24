// Semantics probe for v_dot2c_f32_bf16 on gfx950 (inline asm; the compiler has no builtin): compares a chain of dot2c
// against fmaf on the same packed bf16 pairs.   hipcc --offload-arch=gfx950 -O3 -o /tmp/t scripts/test_dot2c.hip && /tmp/t
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
__device__ __forceinline__ float dot2c(unsigned a, unsigned b, float acc) {
  asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
  return acc;
}
__device__ __forceinline__ float lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__global__ void k(const unsigned* a, const unsigned* b, float* out, int n) {
  const int t = threadIdx.x;
  float d = 0.f, f = 0.f, d4[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < n; ++i) {
    const unsigned ua = a[i * 64 + t], ub = b[i * 64 + t];
    d = dot2c(ua, ub, d);
    d4[i & 3] = dot2c(ua, ub, d4[i & 3]);
    f = fmaf(lo(ua), lo(ub), f);
    f = fmaf(hi(ua), hi(ub), f);
  }
  out[t] = d;
  out[64 + t] = f;
  out[128 + t] = (d4[0] + d4[1]) + (d4[2] + d4[3]);
  // single op probes
  out[192 + t] = dot2c(a[t], b[t], 1.0f);
  out[256 + t] = fmaf(hi(a[t]), hi(b[t]), fmaf(lo(a[t]), lo(b[t]), 1.0f));
}
static unsigned short f2bf(float x) { unsigned u; memcpy(&u, &x, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
int main() {
  const int n = 256;
  unsigned *ha = (unsigned*)malloc(n * 64 * 4), *hb = (unsigned*)malloc(n * 64 * 4);
  srand(1);
  for (int i = 0; i < n * 64; ++i) {
    auto r = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    ha[i] = f2bf(r()) | ((unsigned)f2bf(r()) << 16);
    hb[i] = f2bf(r() * 0.02f) | ((unsigned)f2bf(r() * 0.02f) << 16);
  }
  unsigned *da, *db; float* dout;
  hipMalloc(&da, n * 64 * 4); hipMalloc(&db, n * 64 * 4); hipMalloc(&dout, 320 * 4);
  hipMemcpy(da, ha, n * 64 * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb, n * 64 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout, n);
  float h[320];
  hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
  double md = 0, m4 = 0, m1 = 0;
  for (int t = 0; t < 64; ++t) {
    md = fmax(md, fabs(h[t] - h[64 + t]));
    m4 = fmax(m4, fabs(h[128 + t] - h[64 + t]));
    m1 = fmax(m1, fabs(h[192 + t] - h[256 + t]));
  }
  printf("chain dot2c vs fmaf: max |diff| %.3e (values ~%.3f); 4-chain: %.3e; single op: %.3e\n", md, h[64], m4, m1);
  printf("lane0: dot2c %.7f fmaf %.7f | single: %.7f vs %.7f\n", h[0], h[64], h[192], h[256]);
  return 0;
}
