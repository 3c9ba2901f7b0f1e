// Probe: cost of LDS gathers by access width and address pattern on gfx950 (what bounds the ball-query candidate walk).
// 256 workgroups x 512 threads (one per CU, 2 waves per SIMD), every lane issues ITER x 8 reads at 8 fixed per-lane addresses inside
// a 16 KB block.  Prints cycles per wave-instruction per CU (kernel time x 2.4 GHz / instructions issued on one CU).
//   hipcc --offload-arch=gfx950 -O3 -o lds_gather lds_gather.hip && ./lds_gather
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

enum { B128 = 0, B64 = 1, B32 = 2, R2B32 = 3, BPERM = 4, W16 = 5 };

template <int KIND>
__global__ void __launch_bounds__(512) gather(const unsigned *__restrict__ addr, int iters, float *sink) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 4096 + 64; i += 512) lds[i] = (float)i;
  __syncthreads();
  unsigned a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = addr[u * 512 + threadIdx.x];
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    // eight reads back to back behind ONE wait: the LDS pipe, not a single wave's round trip, is what is timed
    if constexpr (KIND == B128) {
      float4 v[8];
      asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %9\n ds_read_b128 %2, %10\n ds_read_b128 %3, %11\n ds_read_b128 %4, %12\n ds_read_b128 %5, %13\n"
                   "ds_read_b128 %6, %14\n ds_read_b128 %7, %15\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                   : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].w;
    } else if constexpr (KIND == B64 || KIND == R2B32) {
      float2 v[8];
      if constexpr (KIND == B64)
        asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %9\n ds_read_b64 %2, %10\n ds_read_b64 %3, %11\n ds_read_b64 %4, %12\n ds_read_b64 %5, %13\n"
                     "ds_read_b64 %6, %14\n ds_read_b64 %7, %15\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
      else
        asm volatile("ds_read2_b32 %0, %8 offset1:1\n ds_read2_b32 %1, %9 offset1:1\n ds_read2_b32 %2, %10 offset1:1\n ds_read2_b32 %3, %11 offset1:1\n"
                     "ds_read2_b32 %4, %12 offset1:1\n ds_read2_b32 %5, %13 offset1:1\n ds_read2_b32 %6, %14 offset1:1\n ds_read2_b32 %7, %15 offset1:1\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].y;
    } else if constexpr (KIND == B32 || KIND == BPERM) {
      float v[8];
      if constexpr (KIND == B32)
        asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %9\n ds_read_b32 %2, %10\n ds_read_b32 %3, %11\n ds_read_b32 %4, %12\n ds_read_b32 %5, %13\n"
                     "ds_read_b32 %6, %14\n ds_read_b32 %7, %15\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
      else
        asm volatile("ds_bpermute_b32 %0, %8, %16\n ds_bpermute_b32 %1, %9, %16\n ds_bpermute_b32 %2, %10, %16\n ds_bpermute_b32 %3, %11, %16\n ds_bpermute_b32 %4, %12, %16\n"
                     "ds_bpermute_b32 %5, %13, %16\n ds_bpermute_b32 %6, %14, %16\n ds_bpermute_b32 %7, %15, %16\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(acc));
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("ds_write_b16 %0, %1" : : "v"(a[u]), "v"(acc) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  if (acc == 1234.5f) sink[0] = acc;
}

static unsigned rnd(unsigned &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

int main() {
  const int iters = 2000, wgs = 256;
  unsigned *ha = (unsigned *)malloc(8 * 512 * 4), *da;
  float *sink;
  hipMalloc(&da, 8 * 512 * 4);
  hipMalloc(&sink, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  struct { const char *name; int kind; int pattern; } cases[] = {
      {"ds_read_b128  random 16B-aligned", B128, 0}, {"ds_read_b128  quad = lane%8 (rotated)", B128, 1}, {"ds_read_b128  sequential (lane*16)", B128, 2},
      {"ds_read_b128  one address (broadcast)", B128, 3}, {"ds_read_b128  sorted-like (lane*32 + noise)", B128, 4},
      {"ds_read_b64   random 8B-aligned", B64, 0}, {"ds_read_b64   sequential", B64, 2},
      {"ds_read2_b32  random dword pair", R2B32, 5}, {"ds_read2_b32  sequential", R2B32, 2},
      {"ds_read_b32   random", B32, 5}, {"ds_read_b32   sequential", B32, 2},
      {"ds_bpermute_b32 random lanes", BPERM, 6}, {"ds_write_b16  random 2B-aligned", W16, 7}, {"ds_write_b16  own row (tid*52 + k*2)", W16, 8}};
  for (auto &c : cases) {
    unsigned s = 12345;
    for (int u = 0; u < 8; ++u)
      for (int t = 0; t < 512; ++t) {
        const int lane = t & 63;
        unsigned a = 0;
        const unsigned sz = c.kind == B128 ? 16 : c.kind == B64 ? 8 : 4;
        switch (c.pattern) {
          case 0: a = (rnd(s) % (16384 / sz)) * sz; break;
          case 1: a = ((rnd(s) % 128) * 8 + (lane & 7)) * 16; break;
          case 2: a = ((lane + u * 64) * sz) % 16384; break;
          case 3: a = u * 64; break;
          case 4: a = ((lane * 2 + rnd(s) % 3 + u * 128) % 1024) * 16; break;
          case 5: a = (rnd(s) % 4095) * 4; break;
          case 6: a = (rnd(s) % 64) * 4; break;
          case 7: a = (rnd(s) % 8192) * 2; break;
          case 8: a = (t * 52 + (rnd(s) % 16) * 2) % 16384 + 0; break;
        }
        ha[u * 512 + t] = a;
      }
    hipMemcpy(da, ha, 8 * 512 * 4, hipMemcpyHostToDevice);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, 0);
      const size_t shm = (4096 + 64) * 4 + 16384;
      switch (c.kind) {
        case B128: hipLaunchKernelGGL(gather<B128>, dim3(wgs), dim3(512), shm, 0, da, iters, sink); break;
        case B64: hipLaunchKernelGGL(gather<B64>, dim3(wgs), dim3(512), shm, 0, da, iters, sink); break;
        case B32: hipLaunchKernelGGL(gather<B32>, dim3(wgs), dim3(512), shm, 0, da, iters, sink); break;
        case R2B32: hipLaunchKernelGGL(gather<R2B32>, dim3(wgs), dim3(512), shm, 0, da, iters, sink); break;
        case BPERM: hipLaunchKernelGGL(gather<BPERM>, dim3(wgs), dim3(512), shm, 0, da, iters, sink); break;
        default: hipLaunchKernelGGL(gather<W16>, dim3(wgs), dim3(512), shm, 0, da, iters, sink); break;
      }
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double instr_per_cu = 8.0 * iters * 8;       // 8 waves x iters x 8 reads
    printf("%-46s %8.3f ms   %6.1f clk per wave-instruction (2.4 GHz)   %5.1f B/clk/CU\n", c.name, ms, ms * 1e-3 * 2.4e9 / instr_per_cu,
           64.0 * (c.kind == B128 ? 16 : c.kind == B64 || c.kind == R2B32 ? 8 : c.kind == W16 ? 2 : 4) / (ms * 1e-3 * 2.4e9 / instr_per_cu));
  }
  return 0;
}
