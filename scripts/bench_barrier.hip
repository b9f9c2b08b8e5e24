// Micro-benchmark + correctness check for the grid barrier of the persistent decode kernel (decode_mega.hip).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/bench_barrier scripts/bench_barrier.hip && /tmp/bench_barrier
// Each iteration: every block publishes a value (agent-scope store), barrier, reads two other blocks' values (agent-scope
// loads) and checks them, barrier.  Reports microseconds per barrier and the number of stale reads for each variant.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr unsigned LIMIT = 1u << 20;

__device__ __forceinline__ unsigned ld_coh(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_coh(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned add_coh(unsigned* p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct Bar {
  unsigned* mem;    // [0]: root counter, [32]: error flag, [64 + 32*g]: group counters, [4096 + 32*g]: group flags
  unsigned nblk, done, bid;
  int sleep_n;
};

template <int SLEEP>
__device__ __forceinline__ bool spin_until(const unsigned* p, unsigned target, unsigned* err) {
  unsigned polls = 0;
  while ((int)(ld_coh(p) - target) < 0) {
    __builtin_amdgcn_s_sleep(SLEEP);
    if ((++polls & 63u) == 0) {
      if (ld_coh(err) != 0u) return false;
      if (polls > LIMIT) { st_coh(err, 1u); return false; }
    }
  }
  return true;
}

// MODE 0: flat counter, s_sleep 1.  MODE 1: flat counter, s_sleep 16.  MODE 2: groups of GS arrive at a group counter,
// the last arriver of a group arrives at the root; everybody polls the root.  MODE 3: like 2, but only the group's
// block 0 polls the root and then raises the group's flag, which the others poll.
template <int MODE, int GS>
__device__ __forceinline__ void barrier(Bar& b) {
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's stores have been acknowledged
  __syncthreads();
  ++b.done;
  if (threadIdx.x == 0) {
    unsigned* root = b.mem;
    unsigned* err = b.mem + 32;
    if (MODE <= 1) {
      add_coh(root, 1u);
      if (MODE == 0) spin_until<1>(root, b.done * b.nblk, err); else spin_until<16>(root, b.done * b.nblk, err);
    } else {
      const unsigned g = b.bid / GS, ng = (b.nblk + GS - 1) / GS;
      const unsigned gsize = min((unsigned)GS, b.nblk - g * GS);
      unsigned* gc = b.mem + 64 + 32 * g;
      const unsigned prev = add_coh(gc, 1u);
      if (prev + 1 == b.done * gsize) add_coh(root, 1u);  // last of the group this round
      if (MODE == 2) {
        spin_until<4>(root, b.done * ng, err);
      } else {
        unsigned* gf = b.mem + 4096 + 32 * g;
        if (b.bid % GS == 0) {
          spin_until<1>(root, b.done * ng, err);
          st_coh(gf, b.done);
        } else {
          spin_until<4>(gf, b.done, err);
        }
      }
    }
  }
  __syncthreads();
}

template <int MODE, int GS>
__global__ __launch_bounds__(256) void kern(unsigned* mem, unsigned* data, int iters, unsigned* errs, int do_check) {
  Bar b{mem, gridDim.x, 0u, blockIdx.x, 0};
  const unsigned nblk = gridDim.x, bid = blockIdx.x;
  for (int it = 0; it < iters; ++it) {
    if (do_check == 1) {
      if (threadIdx.x < 64) st_coh(data + bid * 64 + threadIdx.x, (unsigned)it * 1000003u + bid * 64 + threadIdx.x);
    } else if (do_check == 2) {  // false sharing: 2-byte elements of neighbouring blocks share 128-byte lines
      if (threadIdx.x < 4) __hip_atomic_store((unsigned short*)data + bid * 4 + threadIdx.x, (unsigned short)(it * 7 + bid * 4 + threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    barrier<MODE, GS>(b);
    if (do_check == 2) {
      // every block reads the whole vector (like the matvec prologue reads x): 8-byte agent-scope loads
      const unsigned n16 = nblk * 4;
      for (unsigned i = threadIdx.x * 4; i < n16; i += 256 * 4) {
        const unsigned long long v = __hip_atomic_load((const unsigned long long*)((const unsigned short*)data + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int j = 0; j < 4; ++j)
          if ((unsigned short)(v >> (16 * j)) != (unsigned short)(it * 7 + i + j)) atomicAdd(errs, 1u);
      }
    }
    if (do_check == 1) {
      const unsigned o1 = (bid + 1) % nblk, o2 = (bid + nblk / 2 + 3) % nblk;
      if (threadIdx.x < 64) {
        const unsigned v1 = ld_coh(data + o1 * 64 + threadIdx.x), v2 = ld_coh(data + o2 * 64 + threadIdx.x);
        if (v1 != (unsigned)it * 1000003u + o1 * 64 + threadIdx.x) atomicAdd(errs, 1u);
        if (v2 != (unsigned)it * 1000003u + o2 * 64 + threadIdx.x) atomicAdd(errs, 1u);
      }
    }
    barrier<MODE, GS>(b);
  }
}

template <int MODE, int GS>
void run(const char* name, int grid, int iters, int do_check) {
  unsigned *mem, *data, *errs;
  CK(hipMalloc(&mem, 1 << 20));
  CK(hipMalloc(&data, grid * 64 * 4));
  CK(hipMalloc(&errs, 4));
  CK(hipMemset(mem, 0, 1 << 20));
  CK(hipMemset(errs, 0, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((kern<MODE, GS>), dim3(grid), dim3(256), 0, 0, mem, data, 4, errs, do_check);  // warm
  CK(hipDeviceSynchronize());
  CK(hipMemset(mem, 0, 1 << 20));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((kern<MODE, GS>), dim3(grid), dim3(256), 0, 0, mem, data, iters, errs, do_check);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned herr, hflag;
  CK(hipMemcpy(&herr, errs, 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&hflag, mem + 32, 4, hipMemcpyDeviceToHost));
  printf("%-28s grid %4d check %d: %7.2f us/barrier   stale reads %u   timeout %u\n", name, grid, do_check,
         1e3 * ms / (2.0 * iters), herr, hflag);
  fflush(stdout);
  CK(hipFree(mem)); CK(hipFree(data)); CK(hipFree(errs));
}

int main() {
  const int iters = 300;
  for (int grid : {512}) {
    for (int chk : {0, 1, 2}) {
      run<0, 1>("flat sleep1", grid, iters, chk);
      run<1, 1>("flat sleep16", grid, iters, chk);
      run<2, 8>("tree8 poll-root", grid, iters, chk);
      run<2, 16>("tree16 poll-root", grid, iters, chk);
      run<2, 32>("tree32 poll-root", grid, iters, chk);
      run<3, 8>("tree8 group-flag", grid, iters, chk);
      run<3, 16>("tree16 group-flag", grid, iters, chk);
      run<3, 32>("tree32 group-flag", grid, iters, chk);
    }
  }
  return 0;
}
