// Reduced reproducer of the two-stream hazard (round 6; tools/victim_probe.py, profiles/r06/eager_beside_graph.txt): the segmentation
// fan-feature kernel cut down to gathers + a linear sort key + the 9-element sorting network + stores (no atan2f, no tie branch, K = 9).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Irepsurf_amd/csrc -Iinclude \
//         -x hip -c tools/victim/fan_sort_only.hip -o /tmp/f.o && hipcc --offload-arch=gfx950 -shared -o tools/victim/libfan_slp.so /tmp/f.o build/rs_lib.cpp.o
//   the same with -fno-slp-vectorize -> tools/victim/libfan_noslp.so;   FAN_LIB=libfan_slp.so python tools/victim_probe.py 20000 60 4096 8 9
// Beside the replaying network graph: 434-610 of 20 000 launches wrote other values in lanes 48..63 of a wave with the SLP-vectorized build
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 with op_sel / SGPR operands in the key arithmetic), 0 of 20 000 without; alone: 0 either way.
#include "rs_common.h"
#include "umbrella_fan.h"
namespace {
constexpr int SG_THREADS = 256;
// first cloud c with q < ends[c]  (the reference walks linearly: knnquery_cuda_kernel.cu:51-62)
__device__ __forceinline__ int cloud_of(int q, const int *__restrict__ ends, int b) {
  int lo = 0, hi = b - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (q < ends[mid]) hi = mid; else lo = mid + 1;
  }
  return lo;
}

template <int K, bool ROT>
__global__ void __launch_bounds__(SG_THREADS)
fan_packed_kernel(int m, int b, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                  const int *__restrict__ knn_idx, const int *__restrict__ new_offset,
                  const float *__restrict__ inv_sign, float *__restrict__ feat) {
  const int q = blockIdx.x * SG_THREADS + threadIdx.x;
  if (q >= m) return;
  const float qx = new_xyz[q * 3 + 0], qy = new_xyz[q * 3 + 1], qz = new_xyz[q * 3 + 2];
  float ox[K], oy[K], oz[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const int p = knn_idx[(size_t)q * K + j];
    ox[j] = xyz[(size_t)p * 3 + 0] - qx; oy[j] = xyz[(size_t)p * 3 + 1] - qy; oz[j] = xyz[(size_t)p * 3 + 2] - qz;
  }
  const float flip = inv_sign ? inv_sign[cloud_of(q, new_offset, b)] : 1.f;

  {
    constexpr int G = K;
    float *orow = feat + (size_t)q * (K * 10);
    float key[G], kx[G], ky[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      kx[j] = rs_fma(oz[j], -0.5f, rs_fma(oy[j], 0.7071f, ox[j] * 0.5f));
      ky[j] = rs_fma(oz[j], 0.5f, rs_fma(oy[j], 0.7071f, ox[j] * -0.5f));
      key[j] = rs_fma(kx[j], 0.37f, ky[j] * 0.21f) + 0.5f;
      orow[j] = key[j];                 // stage A: keys before the sort
      orow[9 + j] = ox[j];              // stage A: loaded offsets (x)
    }
#pragma unroll
    for (int round = 0; round < G; ++round) {
#pragma unroll
      for (int j = (round & 1); j + 1 < G; j += 2) {
        const bool sw = (key[j + 1] - key[j]) < 0.f;
        const float tk = key[j], tx = ox[j], ty = oy[j], tz = oz[j];
        key[j] = sw ? key[j + 1] : tk; ox[j] = sw ? ox[j + 1] : tx; oy[j] = sw ? oy[j + 1] : ty; oz[j] = sw ? oz[j + 1] : tz;
        key[j + 1] = sw ? tk : key[j + 1]; ox[j + 1] = sw ? tx : ox[j + 1]; oy[j + 1] = sw ? ty : oy[j + 1]; oz[j + 1] = sw ? tz : oz[j + 1];
        const float ux_ = kx[j], uy_ = ky[j];
        kx[j] = sw ? kx[j + 1] : ux_; ky[j] = sw ? ky[j + 1] : uy_;
        kx[j + 1] = sw ? ux_ : kx[j + 1]; ky[j + 1] = sw ? uy_ : ky[j + 1];
      }
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
      orow[18 + j] = key[j];            // stage B: keys after the sort
      orow[27 + j] = ox[j]; orow[36 + j] = oy[j]; orow[45 + j] = oz[j];
    }
#pragma unroll
    for (int j = 54; j < 90; ++j) orow[j] = 0.f;
  }

}

}
extern "C" int rs_umbrella_fan_offset(int m, int k, int b, int rotate, const float *xyz, const float *new_xyz,
                                      const int *knn_idx, const int *new_offset, const float *inv_sign,
                                      float *feat, void *stream) {
  RS_REQUIRE(m >= 0 && b >= 0, "rs_umbrella_fan_offset: negative size");
  if (m == 0 || b == 0) return RS_OK;
  RS_REQUIRE(k == 5 || k == 9 || k == 13 || k == 17,
             "rs_umbrella_fan_offset: k=%d not built (group_size+1 must be 5, 9, 13 or 17)", k);
  RS_REQUIRE(xyz && new_xyz && knn_idx && new_offset && feat, "rs_umbrella_fan_offset: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(rs_cdiv(m, SG_THREADS)), block(SG_THREADS);
#define RS_LAUNCH_FAN(K)                                                                                      \
  do {                                                                                                        \
    if (rotate) hipLaunchKernelGGL((fan_packed_kernel<K, true>), grid, block, 0, st, m, b, xyz, new_xyz,      \
                                   knn_idx, new_offset, inv_sign, feat);                                      \
    else hipLaunchKernelGGL((fan_packed_kernel<K, false>), grid, block, 0, st, m, b, xyz, new_xyz, knn_idx,   \
                            new_offset, inv_sign, feat);                                                      \
  } while (0)
  switch (k) {
    case 9: RS_LAUNCH_FAN(9); break;
    default: return 1;
  }
#undef RS_LAUNCH_FAN
  RS_CHECK_LAUNCH("rs_umbrella_fan_offset");
  return RS_OK;
}
