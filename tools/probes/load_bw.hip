// Probe: streaming-read throughput of tile-shaped float4 loads, the way the GEMM / weight-gradient kernels read.
// 512 workgroups x 256 threads; per stage a workgroup reads ROWS x COLS floats of a row-major (R x LD) matrix.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int ROWS, int COLS, int DEPTH>
__global__ void __launch_bounds__(256, 2) rd(const float *__restrict__ a, long long nrows, int ld, int col0, float *sink) {
  constexpr int TPR = COLS / 4, RPP = 256 / TPR, VECS = ROWS / RPP;
  const int c = (threadIdx.x % TPR) * 4 + col0, r = threadIdx.x / TPR;
  float4 acc = make_float4(0, 0, 0, 0);
  const long long step = (long long)gridDim.x * ROWS;
  float4 buf[DEPTH][VECS];
  long long r0 = (long long)blockIdx.x * ROWS;
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d)
#pragma unroll
    for (int p = 0; p < VECS; ++p) {
      const long long rr = min(r0 + d * step + p * RPP + r, nrows - 1);
      buf[d][p] = *reinterpret_cast<const float4 *>(a + rr * ld + c);
    }
  int it = 0;
  for (; r0 < nrows; r0 += step, ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if ((it % DEPTH) == d) {
        const int dn = (d + DEPTH - 1) % DEPTH;
#pragma unroll
        for (int p = 0; p < VECS; ++p) {
          const long long rr = min(r0 + (DEPTH - 1) * step + p * RPP + r, nrows - 1);
          buf[dn][p] = *reinterpret_cast<const float4 *>(a + rr * ld + c);
        }
#pragma unroll
        for (int p = 0; p < VECS; ++p) { acc.x += buf[d][p].x; acc.y += buf[d][p].y; acc.z += buf[d][p].z; acc.w += buf[d][p].w; }
      }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 1234.5f) sink[0] = acc.x;
}

template <int ROWS, int COLS, int DEPTH>
void run(const char *name, const float *a, long long nrows, int ld, float *sink, int wgs) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((rd<ROWS, COLS, DEPTH>), dim3(wgs), dim3(256), 0, 0, a, nrows, ld, 0, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep == 2) printf("%-40s wgs=%4d  %8.1f us  %7.1f GB/s\n", name, wgs, ms * 1e3, (double)nrows * COLS * 4 / (ms * 1e-3) / 1e9);
  }
}

int main() {
  const long long nrows = 262144 * 4; const int ld = 256;      // 1 GiB matrix: beyond the 256 MB Infinity Cache
  float *a, *sink; hipMalloc(&a, (size_t)nrows * ld * 4); hipMalloc(&sink, 4); hipMemset(a, 0, (size_t)nrows * ld * 4);
  for (int wgs : {256, 512, 1024, 2048}) {
    run<32, 128, 2>("32 rows x 128 cols (half rows), depth 2", a, nrows, ld, sink, wgs);
    run<32, 128, 3>("32 rows x 128 cols (half rows), depth 3", a, nrows, ld, sink, wgs);
    run<32, 256, 2>("32 rows x 256 cols (full rows), depth 2", a, nrows, ld, sink, wgs);
    run<128, 32, 2>("128 rows x 32 cols (GEMM chunk), depth 2", a, nrows, ld, sink, wgs);
    run<128, 32, 3>("128 rows x 32 cols (GEMM chunk), depth 3", a, nrows, ld, sink, wgs);
  }
  return 0;
}
