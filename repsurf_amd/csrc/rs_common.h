// rs_common.h — shared device/host helpers for librepsurf_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/repsurf_hip.h"

#define RS_WAVE 64

// ---- host-side error plumbing ------------------------------------------------
void rs_set_error(const char *fmt, ...);

#define RS_REQUIRE(cond, ...)                 \
  do {                                        \
    if (!(cond)) {                            \
      rs_set_error(__VA_ARGS__);              \
      return RS_ERR_ARG;                      \
    }                                         \
  } while (0)

#define RS_CHECK_LAUNCH(name)                                                   \
  do {                                                                          \
    hipError_t _e = hipGetLastError();                                          \
    if (_e != hipSuccess) {                                                     \
      rs_set_error("%s: launch failed: %s", name, hipGetErrorString(_e));       \
      return RS_ERR_HIP_BASE + (int)_e;                                         \
    }                                                                           \
  } while (0)

static inline int rs_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// LDS a workgroup may ask for on this device (rs_lib.cpp).  Several kernels size their dynamic LDS for MI355X's 160 KB: a launch that asks
// for more than the device has fails HERE, with a message, instead of at dispatch.
int rs_lds_limit(void);
#define RS_REQUIRE_LDS(bytes, name)                                                                                          \
  RS_REQUIRE((size_t)(bytes) <= (size_t)rs_lds_limit(), "%s: needs %zu bytes of LDS per workgroup, the device offers %d", name, \
             (size_t)(bytes), rs_lds_limit())

// ---- exact-arithmetic helpers -------------------------------------------------
// The geometry translation units are compiled with -ffp-contract=off, so a*b+c
// below is two roundings; rs_fma is the only fused form.
__device__ __forceinline__ float rs_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// square_distance of classification/modules/pointnet2_utils.py:15-25 for ONE pair:
//   dot = fma(qz,pz, fma(qy,py, qx*px))         (torch.matmul, K=3, sequential-k FMA chain)
//   d   = ((-2*dot) + |q|^2) + |p|^2            (two in-place adds, each rounded)
// -2*dot is exact, so (-2*dot)+qq == fma(-2,dot,qq).
__device__ __forceinline__ float rs_sqdist_expanded(float qx, float qy, float qz, float qq,
                                                    float px, float py, float pz, float pp) {
  float dot = rs_fma(qz, pz, rs_fma(qy, py, qx * px));
  return rs_fma(-2.0f, dot, qq) + pp;
}
// torch.sum(p ** 2, -1): ((x*x + y*y) + z*z)
__device__ __forceinline__ float rs_sqnorm(float x, float y, float z) { return (x * x + y * y) + z * z; }

// ---- wave64 cross-lane reductions (DPP inside rows of 16, gfx950 permlane swaps across rows) ----
template <int CTRL>
__device__ __forceinline__ unsigned rs_dpp(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
#define RS_DPP_QUAD_XOR1 0xB1   // quad_perm [1,0,3,2]
#define RS_DPP_QUAD_XOR2 0x4E   // quad_perm [2,3,0,1]
#define RS_DPP_ROW_HALF_MIRROR 0x141
#define RS_DPP_ROW_MIRROR 0x140

// max over the 64 lanes, result in every lane
__device__ __forceinline__ unsigned rs_wave_max_u32(unsigned v) {
  v = max(v, rs_dpp<RS_DPP_QUAD_XOR1>(v));
  v = max(v, rs_dpp<RS_DPP_QUAD_XOR2>(v));
  v = max(v, rs_dpp<RS_DPP_ROW_HALF_MIRROR>(v));
  v = max(v, rs_dpp<RS_DPP_ROW_MIRROR>(v));
  auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  v = max((unsigned)r[0], (unsigned)r[1]);
  auto s = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  v = max((unsigned)s[0], (unsigned)s[1]);
  return v;
}
__device__ __forceinline__ unsigned rs_wave_min_u32(unsigned v) {
  v = min(v, rs_dpp<RS_DPP_QUAD_XOR1>(v));
  v = min(v, rs_dpp<RS_DPP_QUAD_XOR2>(v));
  v = min(v, rs_dpp<RS_DPP_ROW_HALF_MIRROR>(v));
  v = min(v, rs_dpp<RS_DPP_ROW_MIRROR>(v));
  auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  v = min((unsigned)r[0], (unsigned)r[1]);
  auto s = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  v = min((unsigned)s[0], (unsigned)s[1]);
  return v;
}
// float sum over the 64 lanes (fixed butterfly order => deterministic), result in every lane
__device__ __forceinline__ float rs_wave_sum_f32(float x) {
  unsigned v = __float_as_uint(x);
  v = __float_as_uint(__uint_as_float(v) + __uint_as_float(rs_dpp<RS_DPP_QUAD_XOR1>(v)));
  v = __float_as_uint(__uint_as_float(v) + __uint_as_float(rs_dpp<RS_DPP_QUAD_XOR2>(v)));
  v = __float_as_uint(__uint_as_float(v) + __uint_as_float(rs_dpp<RS_DPP_ROW_HALF_MIRROR>(v)));
  v = __float_as_uint(__uint_as_float(v) + __uint_as_float(rs_dpp<RS_DPP_ROW_MIRROR>(v)));
  auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  v = __float_as_uint(__uint_as_float((unsigned)r[0]) + __uint_as_float((unsigned)r[1]));
  auto s = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  v = __float_as_uint(__uint_as_float((unsigned)s[0]) + __uint_as_float((unsigned)s[1]));
  return __uint_as_float(v);
}
__device__ __forceinline__ int rs_lane() { return (int)(threadIdx.x & 63); }
// number of set bits of `mask` below this lane
__device__ __forceinline__ int rs_mbcnt(unsigned long long mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
__device__ __forceinline__ float rs_uniform(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ int rs_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// XCD-aware block remap: hardware places block i on XCD i % 8 (MI355X_MICROARCH.md).  Given a
// grid of `groups * per_group` blocks where the blocks of one group share data (one cloud),
// return (group, member) such that all members of a group run on one XCD / one L2.
__device__ __forceinline__ void rs_xcd_remap(int bid, int groups, int per_group, int &group, int &member) {
  const int total = groups * per_group;
  if ((groups & 7) == 0) {
    const int xcd = bid & 7;          // which L2
    const int slot = bid >> 3;        // position inside that XCD's stream
    const int gpx = groups >> 3;      // groups per XCD
    group = xcd * gpx + slot / per_group;
    member = slot % per_group;
  } else {
    group = bid / per_group;
    member = bid % per_group;
  }
  (void)total;
}

// csrc/knn_wide.hip: the nsample > 64 route of rs_knnquery / rs_knnquery_offset (one wave per query, repeated minimum)
void rs_launch_knn_wide_dense(int b, int n, int m, int nsample, const float *xyz, const float *new_xyz, int *idx,
                              float *dist2, hipStream_t st);
void rs_launch_knn_wide_packed(int m, int nsample, int b, const float *xyz, const float *new_xyz, const int *offset,
                               const int *new_offset, int *idx, float *dist2, hipStream_t st);
