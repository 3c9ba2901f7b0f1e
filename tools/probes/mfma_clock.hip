// Probe: shader clock under MFMA load.  Each workgroup (256 threads) issues back-to-back fp32 MFMAs for a fixed
// count; clock64() counts shader cycles, wall_clock64() a constant 100 MHz timer.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) burn(int iters, long long *out, float *sink) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-4f;
  __syncthreads();
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = w1 - w0; }
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
  if (s == 12345.678f) sink[0] = s;
}

int main() {
  long long *d; float *sink; hipMalloc(&d, 4096 * 16); hipMalloc(&sink, 4);
  long long h[8192];
  const int iters = 20000;   // 80000 MFMAs per wave = 5.12 M cycles at 64 cycles each
  for (int wgs : {1, 64, 128, 256, 512, 1024, 2048}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(burn, dim3(wgs), dim3(256), 0, 0, iters, d, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h, d, wgs * 16, hipMemcpyDeviceToHost);
      double cyc = 0, wall = 0; for (int i = 0; i < wgs; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
      cyc /= wgs; wall /= wgs;
      const double mhz = cyc / (wall / 100.0);     // wall ticks at 100 MHz
      const double flops = (double)wgs * 4 * iters * 4 * 4096.0;
      if (rep) printf("wgs=%5d  kernel %8.3f ms  avg shader cycles/WG %.0f  wall %.1f us  => %.0f MHz ; cycles per MFMA per wave %.1f ; %.1f TFLOP/s\n",
             wgs, ms, cyc, wall / 100.0, mhz, cyc / (iters * 4.0), flops / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
