// Probe: cost of a grid-wide meeting inside one launch (256 workgroups x 512 threads, one per CU), per variant:
//   0 one counter, agent-scope release / acquire (buffer_wbl2 + buffer_inv), poll every ~64 cycles
//   1 one counter, relaxed arrival + relaxed polls (no cache maintenance), poll every ~64 cycles
//   2 as 1, poll every ~1000 cycles
//   3 two levels: 8 counters (workgroup % 8 = its XCD), the last arrival of each bumps a top counter; everybody polls the top one
//   4 as 3, poll every ~1000 cycles
//   5 two levels, and only thread 0 of workgroups < 8 ... (n/a)
// Each workgroup also writes 32 doubles (sc1 stores) before a meeting and reads all 256 x 32 after it (sc1 loads), like the
// constructor passes do, so the figure includes the exchange.  hipcc --offload-arch=gfx950 -O3 -o grid_meet grid_meet.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <typename T> __device__ __forceinline__ void gstore(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ T gload(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int VAR>
__device__ __forceinline__ void meet(unsigned *sync, unsigned round, unsigned nwg) {
  if (VAR != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned target = (round + 1) * nwg;
    if (VAR == 0) {
      __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    } else if (VAR == 1 || VAR == 2) {
      __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (gload(sync) < target) __builtin_amdgcn_s_sleep(VAR == 1 ? 1 : 16);
    } else {
      unsigned *mine = sync + 32 * (1 + (blockIdx.x & 7));          // own cache line per XCD
      const unsigned per = nwg / 8;
      const unsigned old = __hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == (round + 1) * per - 1) __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (gload(sync) < (round + 1) * 8) __builtin_amdgcn_s_sleep(VAR == 3 ? 1 : 16);
    }
  }
  __syncthreads();
}

template <int VAR>
__global__ void __launch_bounds__(512) probe(unsigned *sync, double *part, int rounds, double *out) {
  __shared__ double red[16][32];
  const int t = threadIdx.x, v = t & 31, sl = t >> 5;
  double acc = 0.0;
  for (int r = 0; r < rounds; ++r) {
    if (t < 32) gstore(&part[(size_t)blockIdx.x * 32 + t], (double)(r + blockIdx.x + t));
    meet<VAR>(sync, 2 * r, gridDim.x);
    double a = 0.0;
    for (int b = sl; b < (int)gridDim.x; b += 16) a += gload(&part[(size_t)b * 32 + v]);
    red[sl][v] = a;
    __syncthreads();
    if (t < 32) { double s = 0; for (int k = 0; k < 16; ++k) s += red[k][t]; acc += s; }
    meet<VAR>(sync, 2 * r + 1, gridDim.x);      // (nobody overwrites its row before everybody has read it)
  }
  if (t < 32) out[(size_t)blockIdx.x * 32 + t] = acc;
}

int main() {
  const int nwg = 256, rounds = 50;
  unsigned *sync; double *part, *out;
  hipMalloc(&sync, 4096); hipMalloc(&part, nwg * 32 * 8); hipMalloc(&out, nwg * 32 * 8);
  double *h = (double *)malloc(nwg * 32 * 8);
  // expected: sum over r of sum over b of (r + b + t) = rounds*(nwg*t + nwg(nwg-1)/2) + nwg*rounds(rounds-1)/2
  for (int var = 0; var < 5; ++var) {
    float best = 1e9f; int bad = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(sync, 0, 4096);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      switch (var) {
        case 0: hipLaunchKernelGGL(probe<0>, dim3(nwg), dim3(512), 0, 0, sync, part, rounds, out); break;
        case 1: hipLaunchKernelGGL(probe<1>, dim3(nwg), dim3(512), 0, 0, sync, part, rounds, out); break;
        case 2: hipLaunchKernelGGL(probe<2>, dim3(nwg), dim3(512), 0, 0, sync, part, rounds, out); break;
        case 3: hipLaunchKernelGGL(probe<3>, dim3(nwg), dim3(512), 0, 0, sync, part, rounds, out); break;
        default: hipLaunchKernelGGL(probe<4>, dim3(nwg), dim3(512), 0, 0, sync, part, rounds, out); break;
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
      hipMemcpy(h, out, nwg * 32 * 8, hipMemcpyDeviceToHost);
      for (int b = 0; b < nwg; ++b) for (int t = 0; t < 32; ++t) {
        const double want = (double)rounds * ((double)nwg * t + (double)nwg * (nwg - 1) / 2) + (double)nwg * rounds * (rounds - 1) / 2;
        if (h[b * 32 + t] != want) ++bad;
      }
    }
    printf("variant %d: %7.2f us per meeting (+ exchange), %d wrong sums of %d\n", var, best * 1e3f / (2 * rounds), bad, 3 * nwg * 32);
  }
  return 0;
}
