// Probe for the gfx950 behaviour described in profiles/r03_simd_coresidency.md: a packed-f32 VALU op with a swapped second source
// (`v_pk_add_f32 vD, vA, vB op_sel:[0,1] op_sel_hi:[1,0]`, vA != vB) computes lanes 16-31 / 48-63 from the wrong half of vB when
// its wave shares a SIMD with a wave of another kernel that runs on a concurrent stream.
//
// Victims: one wave per row of a 513 x 512 f32 matrix, four values per lane, reduced to a per-lane sum by ONE instruction sequence in
// inline asm (fixed registers v40..v43), stored per lane.  Aggressors: the library's GEMM kernels through the C ABI on a second
// stream, launched from a second host thread.  Every victim launch is compared with the result of the idle GPU.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/simd_coresidency_probe.hip -o /tmp/probe -Laha_amd/csrc -laha_hip \
//         -Wl,-rpath,$PWD/aha_amd/csrc && /tmp/probe
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "../include/aha_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// one packed step under test per victim, then the two halves are added (the result only has to be reproducible)
#define VICTIM_ASM(STEP)                                                                                                              \
  asm volatile("v_mov_b32 v40, %[a0]\n v_mov_b32 v41, %[a1]\n v_mov_b32 v42, %[b0]\n v_mov_b32 v43, %[b1]\n s_nop 7\n" STEP          \
               "s_nop 7\n v_mov_b32 %[r], v40\n"                                                                                      \
               : [r] "=&v"(r) : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1) : "v40", "v41", "v42", "v43")
template <int FORM>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ x, float* __restrict__ out, int rows) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float4 q = *reinterpret_cast<const float4*>(x + (size_t)row * 256 + lane * 4);
  const float a0 = q.x, a1 = q.y, b0 = q.z, b1 = q.w;
  float r;
  if (FORM == 0)        // swapped second source, distinct registers: (a0 + b1, a1 + b0), then the two halves
    VICTIM_ASM("v_pk_add_f32 v[40:41], v[40:41], v[42:43] op_sel:[0,1] op_sel_hi:[1,0]\n s_nop 7\n v_add_f32 v40, v40, v41\n");
  else if (FORM == 1)   // mirrored: swapped FIRST source
    VICTIM_ASM("v_pk_add_f32 v[40:41], v[40:41], v[42:43] op_sel:[1,0] op_sel_hi:[0,1]\n s_nop 7\n v_add_f32 v40, v40, v41\n");
  else if (FORM == 2)   // swapped second source, SAME register: (a0 + a1, a1 + a0)
    VICTIM_ASM("v_pk_add_f32 v[40:41], v[40:41], v[40:41] op_sel:[0,1] op_sel_hi:[1,0]\n s_nop 7\n v_add_f32 v40, v40, v42\n s_nop 1\n v_add_f32 v40, v40, v43\n");
  else if (FORM == 3)   // no operand selection
    VICTIM_ASM("v_pk_add_f32 v[40:41], v[40:41], v[42:43]\n s_nop 7\n v_add_f32 v40, v40, v41\n");
  else if (FORM == 4)   // broadcast of the low half of the second source (the matvec kernels' form)
    VICTIM_ASM("v_pk_mul_f32 v[40:41], v[40:41], v[42:43] op_sel_hi:[1,0]\n s_nop 7\n v_add_f32 v40, v40, v41\n");
  else if (FORM == 5)   // scalar adds only
    VICTIM_ASM("v_add_f32 v40, v40, v41\n s_nop 1\n v_add_f32 v40, v40, v42\n s_nop 1\n v_add_f32 v40, v40, v43\n");
  else if (FORM == 6)   // the swapped second source on a multiply
    VICTIM_ASM("v_pk_mul_f32 v[40:41], v[40:41], v[42:43] op_sel:[0,1] op_sel_hi:[1,0]\n s_nop 7\n v_add_f32 v40, v40, v41\n");
  else if (FORM == 7)   // ... on the second source of an fma
    VICTIM_ASM("v_pk_fma_f32 v[40:41], v[40:41], v[42:43], v[40:41] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n s_nop 7\n v_add_f32 v40, v40, v41\n");
  else if (FORM == 8)   // the swap on the THIRD source of an fma
    VICTIM_ASM("v_pk_fma_f32 v[40:41], v[40:41], v[40:41], v[42:43] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n s_nop 7\n v_add_f32 v40, v40, v41\n");
  else if (FORM == 9)   // high element of the second source in BOTH halves
    VICTIM_ASM("v_pk_add_f32 v[40:41], v[40:41], v[42:43] op_sel:[0,1] op_sel_hi:[1,1]\n s_nop 7\n v_add_f32 v40, v40, v41\n");
  else if (FORM == 10)  // high element of the FIRST source in both halves (the matvec kernels' form since round 3)
    VICTIM_ASM("v_pk_fma_f32 v[40:41], v[42:43], v[40:41], v[40:41] op_sel:[1,0,0]\n s_nop 7\n v_add_f32 v40, v40, v41\n");
  else                  // low element of the first source in both halves (the matvec kernels' other form)
    VICTIM_ASM("v_pk_fma_f32 v[40:41], v[42:43], v[40:41], v[40:41] op_sel_hi:[0,1,1]\n s_nop 7\n v_add_f32 v40, v40, v41\n");
  out[(size_t)row * 64 + lane] = r;
}

constexpr int NV = 13;   // victims: 12 instruction forms + the library kernel
int main() {
  const int rows = 513, M = 513, N = 1024, K = 512;
  std::vector<float> hx((size_t)rows * 256);
  srand(1);
  for (auto& v : hx) v = (float)(rand() & 0xffff) / 65536.0f;
  float *dx, *dout;
  void *A, *W, *Cb, *nw, *ny;
  CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dout, (size_t)rows * 512 * 2));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&Cb, (size_t)M * N * 2));
  CK(hipMalloc(&nw, 512 * 2)); CK(hipMalloc(&ny, (size_t)rows * 512 * 2));
  {
    std::vector<uint16_t> t((size_t)N * K);
    for (auto& v : t) v = 0x3c00 + (rand() & 0xff);   // small positive bf16 values
    CK(hipMemcpy(A, t.data(), (size_t)M * K * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, t.data(), (size_t)N * K * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(nw, t.data(), 512 * 2, hipMemcpyHostToDevice));
  }
  hipStream_t sv, sa;
  CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  const char* vn[NV] = {"pk_add, second source swapped, vA != vB", "pk_add, first source swapped", "pk_add, second source swapped, vA == vB",
                        "pk_add, no op_sel", "pk_mul, op_sel_hi:[1,0] broadcast", "scalar adds", "pk_mul, second source swapped",
                        "pk_fma, second source swapped", "pk_fma, third source swapped", "pk_add, op_sel:[0,1] op_sel_hi:[1,1]",
                        "pk_fma, op_sel:[1,0,0] (first source high)", "pk_fma, op_sel_hi:[0,1,1] (first source low)",
                        "library RMSNorm rows (aha_hip_rmsnorm)"};
  const int tiles[3] = {0, 128, 192};
  std::vector<std::vector<uint8_t>> idle(NV);
  for (int ai = 0; ai < 3; ++ai) {
    std::atomic<bool> stop{false};
    std::thread th;
    if (tiles[ai]) {
      aha_hip_debug_gemm_plan(tiles[ai], 1);
      th = std::thread([&] {
        while (!stop) {
          for (int i = 0; i < 20; ++i) aha_hip_gemm(A, W, Cb, M, N, K, K, K, N / 2, nullptr, nullptr, 4 /* gate+up pairs */, sa);
          (void)hipStreamSynchronize(sa);
        }
      });
    }
    for (int v = 0; v < NV; ++v) {
      const size_t bytes = v == NV - 1 ? (size_t)rows * 512 * 2 : (size_t)rows * 64 * 4;
      std::vector<uint8_t> got(bytes);
      int bad = 0, bad_lanes_odd = 0, bad_lanes_even = 0;
      for (int it = 0; it < 3000; ++it) {
        switch (v) {
          case 0: hipLaunchKernelGGL(victim<0>, dim3((rows + 3) / 4), dim3(256), 0, sv, dx, dout, rows); break;
          case 1: hipLaunchKernelGGL(victim<1>, dim3((rows + 3) / 4), dim3(256), 0, sv, dx, dout, rows); break;
          case 2: hipLaunchKernelGGL(victim<2>, dim3((rows + 3) / 4), dim3(256), 0, sv, dx, dout, rows); break;
          case 3: hipLaunchKernelGGL(victim<3>, dim3((rows + 3) / 4), dim3(256), 0, sv, dx, dout, rows); break;
          case 4: hipLaunchKernelGGL(victim<4>, dim3((rows + 3) / 4), dim3(256), 0, sv, dx, dout, rows); break;
          case 5: hipLaunchKernelGGL(victim<5>, dim3((rows + 3) / 4), dim3(256), 0, sv, dx, dout, rows); break;
          case 6: hipLaunchKernelGGL(victim<6>, dim3((rows + 3) / 4), dim3(256), 0, sv, dx, dout, rows); break;
          case 7: hipLaunchKernelGGL(victim<7>, dim3((rows + 3) / 4), dim3(256), 0, sv, dx, dout, rows); break;
          case 8: hipLaunchKernelGGL(victim<8>, dim3((rows + 3) / 4), dim3(256), 0, sv, dx, dout, rows); break;
          case 9: hipLaunchKernelGGL(victim<9>, dim3((rows + 3) / 4), dim3(256), 0, sv, dx, dout, rows); break;
          case 10: hipLaunchKernelGGL(victim<10>, dim3((rows + 3) / 4), dim3(256), 0, sv, dx, dout, rows); break;
          case 11: hipLaunchKernelGGL(victim<11>, dim3((rows + 3) / 4), dim3(256), 0, sv, dx, dout, rows); break;
          default: aha_hip_rmsnorm(A, nw, ny, rows, 512, 1e-6f, sv); break;
        }
        CK(hipMemcpyAsync(got.data(), v == NV - 1 ? ny : (void*)dout, bytes, hipMemcpyDeviceToHost, sv));
        CK(hipStreamSynchronize(sv));
        if (ai == 0 && it == 0) idle[v] = got;
        if (memcmp(idle[v].data(), got.data(), bytes) != 0) {
          ++bad;
          if (v < NV - 1) {
            const float* g = (const float*)got.data(); const float* r = (const float*)idle[v].data();
            for (size_t i = 0; i < (size_t)rows * 64; ++i)
              if (g[i] != r[i]) { if (((i & 63) >> 4) & 1) ++bad_lanes_odd; else ++bad_lanes_even; }
          }
        }
      }
      printf("aggressor %-22s | victim %-42s: %4d of 3000 launches differ", tiles[ai] == 0 ? "none (idle GPU)" : tiles[ai] == 128 ? "128^2 GEMM tile" : "256x192 GEMM tile", vn[v], bad);
      if (bad && v < NV - 1) printf("  (wrong values in lanes 16-31/48-63: %d, in lanes 0-15/32-47: %d)", bad_lanes_odd, bad_lanes_even);
      printf("\n");
      fflush(stdout);
    }
    if (tiles[ai]) { stop = true; th.join(); aha_hip_debug_gemm_plan(0, 0); }
  }
  return 0;
}
