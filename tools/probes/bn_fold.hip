// Probe (round 4): what folding a BatchNorm finalize launch into its consumer is worth inside a replayed hipGraph.
//   chain A (what the stacks do):  producer (row tiles out + one fp64 partial row per workgroup) -> finalize launch (c/8 workgroups
//                                  sum the 512 partial rows, write scale / shift) -> consumer (reads scale / shift, streams the tensor)
//   chain B (fold):                producer (the same + fp64 atomics into `slots` pre-zeroed rows) -> consumer whose workgroups
//                                  each sum the slots, compute scale / shift, WRITE them to the same global arrays (identical
//                                  values from every workgroup), __syncthreads, then run the unchanged body
// 20 triples per graph, one memset of all the slot rows in front (chain B); time per triple.
// hipcc --offload-arch=gfx950 -O3 -o bn_fold bn_fold.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int T = 256;

// rows x c floats, tile = 64 rows; a workgroup walks tiles blockIdx.x, + gridDim.x, ...; column sums per workgroup
template <bool ATOMIC>
__global__ void __launch_bounds__(T) producer(float *y, long long rows, int c, double *partial, int slots, float seed) {
  const int tid = threadIdx.x;
  const int vec = c / 4, tpr = vec, rpp = T / tpr;           // c / 4 threads per row
  const int col = (tid % tpr) * 4, r_in = tid / tpr;
  float s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  const long long tiles = rows / 64;
  for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
    for (int rl = r_in; rl < 64; rl += rpp) {
      const long long r = t * 64 + rl;
      float4 v = make_float4(seed + (float)(r & 7), seed * 0.5f, (float)(col & 3), 1.f);
      *reinterpret_cast<float4 *>(y + r * c + col) = v;
      s0[0] += v.x; s0[1] += v.y; s0[2] += v.z; s0[3] += v.w;
      s1[0] += v.x * v.x; s1[1] += v.y * v.y; s1[2] += v.z * v.z; s1[3] += v.w * v.w;
    }
  }
  // combine the rpp row slots of each column (fixed order), then one row per workgroup / atomics into a slot
  __shared__ double cs[2][16][256];        // [stat][row slot][column]  (rpp <= 16, c <= 256 here)
  for (int e = 0; e < 4; ++e) { cs[0][r_in][col + e] = s0[e]; cs[1][r_in][col + e] = s1[e]; }
  __syncthreads();
  for (int e = tid; e < 2 * c; e += T) {
    const int st = e / c, cc = e - st * c;
    double t = 0.0;
    for (int k = 0; k < rpp; ++k) t += cs[st][k][cc];
    if (ATOMIC) unsafeAtomicAdd(&partial[((long long)(blockIdx.x % slots) * 2 + st) * c + cc], t);
    else partial[((long long)blockIdx.x * 2 + st) * c + cc] = t;
  }
}

__global__ void __launch_bounds__(256) finalize(int c, long long rows, int nblk, const double *partial, float *scale, float *shift) {
  __shared__ double red[32][8][2];
  const int ex = threadIdx.x & 7, sl = threadIdx.x >> 3, ch = blockIdx.x * 8 + ex;
  double a0 = 0, a1 = 0;
  if (ch < c)
    for (int b = sl; b < nblk; b += 512) {
      double v0[16], v1[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        v0[u] = (b + 32 * u < nblk) ? partial[((long long)(b + 32 * u) * 2 + 0) * c + ch] : 0.0;
        v1[u] = (b + 32 * u < nblk) ? partial[((long long)(b + 32 * u) * 2 + 1) * c + ch] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) { a0 += v0[u]; a1 += v1[u]; }
    }
  red[sl][ex][0] = a0; red[sl][ex][1] = a1;
  __syncthreads();
  if (sl == 0 && ch < c) {
    double t0 = 0, t1 = 0;
    for (int k = 0; k < 32; ++k) { t0 += red[k][ex][0]; t1 += red[k][ex][1]; }
    const double mean = t0 / (double)rows;
    double var = t1 / (double)rows - mean * mean; if (var < 0) var = 0;
    const double is = 1.0 / sqrt(var + 1e-5);
    scale[ch] = (float)is; shift[ch] = (float)(-mean * is);
  }
}

template <bool FOLD>
__global__ void __launch_bounds__(T) consumer(const float *y, long long rows, int c, const double *sums, int slots, float *scale, float *shift, float *out) {
  const int tid = threadIdx.x;
  if (FOLD) {
    for (int ch = tid; ch < c; ch += T) {
      double t0 = 0, t1 = 0;
      for (int s = 0; s < slots; ++s) { t0 += sums[((long long)s * 2 + 0) * c + ch]; t1 += sums[((long long)s * 2 + 1) * c + ch]; }
      const double mean = t0 / (double)rows;
      double var = t1 / (double)rows - mean * mean; if (var < 0) var = 0;
      const double is = 1.0 / sqrt(var + 1e-5);
      scale[ch] = (float)is; shift[ch] = (float)(-mean * is);
    }
    __syncthreads();
  }
  const int vec = c / 4, tpr = vec, rpp = T / tpr;
  const int col = (tid % tpr) * 4, r_in = tid / tpr;
  const float4 sc = *reinterpret_cast<const float4 *>(scale + col), sh = *reinterpret_cast<const float4 *>(shift + col);
  float acc = 0.f;
  const long long tiles = rows / 64;
  for (long long t = blockIdx.x; t < tiles; t += gridDim.x)
    for (int rl = r_in; rl < 64; rl += rpp) {
      const float4 v = *reinterpret_cast<const float4 *>(y + (t * 64 + rl) * c + col);
      acc += fmaxf(v.x * sc.x + sh.x, 0.f) + fmaxf(v.y * sc.y + sh.y, 0.f) + fmaxf(v.z * sc.z + sh.z, 0.f) + fmaxf(v.w * sc.w + sh.w, 0.f);
    }
  if (acc == 12345.678f) out[blockIdx.x * T + tid] = acc;
}

int main() {
  const int nwg = 512, reps = 20, replays = 30;
  hipStream_t st; CK(hipStreamCreate(&st));
  for (int c : {64, 128, 256}) {
    const long long rows = 262144LL * 128 / c;                     // 134 MB tensor either way
    float *y, *scale, *shift, *out; double *partial, *sums;
    CK(hipMalloc(&y, rows * c * 4)); CK(hipMalloc(&scale, c * 4)); CK(hipMalloc(&shift, c * 4)); CK(hipMalloc(&out, nwg * T * 4));
    CK(hipMalloc(&partial, (size_t)nwg * 2 * c * 8));
    const int max_slots = 32;
    CK(hipMalloc(&sums, (size_t)reps * max_slots * 2 * c * 8));
    for (int variant = 0; variant < 6; ++variant) {     // 0: chain A; 1-5: chain B with 1 / 2 / 4 / 8 / 16 slot rows
      const int slots = variant == 0 ? 0 : (1 << (variant - 1));
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      if (variant) CK(hipMemsetAsync(sums, 0, (size_t)reps * max_slots * 2 * c * 8, st));
      for (int r = 0; r < reps; ++r) {
        if (variant == 0) {
          hipLaunchKernelGGL(producer<false>, dim3(nwg), dim3(T), 0, st, y, rows, c, partial, 0, 1.f + r);
          hipLaunchKernelGGL(finalize, dim3(c / 8), dim3(256), 0, st, c, rows, nwg, partial, scale, shift);
          hipLaunchKernelGGL(consumer<false>, dim3(nwg), dim3(T), 0, st, y, rows, c, (const double *)nullptr, 0, scale, shift, out);
        } else {
          double *s = sums + (size_t)r * max_slots * 2 * c;
          hipLaunchKernelGGL(producer<true>, dim3(nwg), dim3(T), 0, st, y, rows, c, s, slots, 1.f + r);
          hipLaunchKernelGGL(consumer<true>, dim3(nwg), dim3(T), 0, st, y, rows, c, (const double *)s, slots, scale, shift, out);
        }
      }
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < replays; ++i) CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      float h[2]; CK(hipMemcpy(h, scale, 8, hipMemcpyDeviceToHost));
      char what[64];
      if (variant == 0) snprintf(what, sizeof what, "finalize launch      ");
      else snprintf(what, sizeof what, "fold, %2d slot rows   ", slots);
      printf("c=%3d rows=%7lld %s: %7.2f us per (producer, %sconsumer) triple   scale[0]=%g\n", c, rows, what,
             1000.f * ms / (replays * reps), variant == 0 ? "finalize, " : "", h[0]);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    CK(hipFree(y)); CK(hipFree(scale)); CK(hipFree(shift)); CK(hipFree(out)); CK(hipFree(partial)); CK(hipFree(sums));
  }
  return 0;
}
