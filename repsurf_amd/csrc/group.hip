// group.hip — row gathers / grouped feature assembly and their scatter-add backwards (gfx950).
//
// The reference moves point data as (b, c, n) and wraps every gather in two transposes
// (classification/modules/pointnet2_utils.py:31-35); its kernels read one float per thread with
// an n-strided address (classification/modules/pointops/src/grouping/grouping_cuda_kernel.cu:60-76).
// Here data stays channels-last: a gathered row is one contiguous 12..1 KB segment, consecutive
// lanes read consecutive channels of it (coalesced) and write consecutive channels of the
// output tile, which is exactly the (rows, channels) operand layout of the shared-MLP GEMM.
// rs_group_features fuses the three gathers, the centre subtraction, xyz2sphere and the concat
// of sample_and_group (classification/modules/repsurface_utils.py:36-57) into one pass.
// These kernels are HBM-bound: bytes = idx + gathered rows (L2-resident, re-read) + the output
// tile written once.
#include "rs_common.h"
#include <math.h>

#define RS_PI_F 3.14159274101257324f
#define RS_TWO_PI_F 6.28318548202514648f

namespace {

constexpr int GR_THREADS = 256;

// out[r, :] = points[cloud(r), idx[r], :]   rows = b * per_cloud
__global__ void __launch_bounds__(GR_THREADS)
gather_rows_kernel(long long rows, int per_cloud, int n, int c, const float *__restrict__ points,
                   const int *__restrict__ idx, float *__restrict__ out) {
  const long long total = rows * c;
  for (long long e = (long long)blockIdx.x * GR_THREADS + threadIdx.x; e < total;
       e += (long long)gridDim.x * GR_THREADS) {
    const long long r = e / c;
    const int ch = (int)(e - r * c);
    const long long cloud = r / per_cloud;
    out[e] = points[(cloud * n + idx[r]) * c + ch];
  }
}

// grad_points[cloud(r), idx[r], :] += grad_out[r, :]
__global__ void __launch_bounds__(GR_THREADS)
gather_rows_bwd_kernel(long long rows, int per_cloud, int n, int c, const float *__restrict__ grad_out,
                       const int *__restrict__ idx, float *__restrict__ grad_points) {
  const long long total = rows * c;
  for (long long e = (long long)blockIdx.x * GR_THREADS + threadIdx.x; e < total;
       e += (long long)gridDim.x * GR_THREADS) {
    const long long r = e / c;
    const int ch = (int)(e - r * c);
    const long long cloud = r / per_cloud;
    atomicAdd(grad_points + (cloud * n + idx[r]) * c + ch, grad_out[e]);
  }
}

// Head channels [offset(3), polar(3)?, normal(cn)] : one thread per (row, channel)
__global__ void __launch_bounds__(GR_THREADS)
group_head_kernel(long long rows, int m, int nsample, int n, int cn, int cpos, int ctot,
                  const float *__restrict__ center, const float *__restrict__ new_center,
                  const float *__restrict__ normal, const int *__restrict__ idx,
                  float *__restrict__ out) {
  const int ch_head = cpos + cn;
  const long long total = rows * ch_head;
  for (long long e = (long long)blockIdx.x * GR_THREADS + threadIdx.x; e < total;
       e += (long long)gridDim.x * GR_THREADS) {
    const long long r = e / ch_head;
    const int ch = (int)(e - r * ch_head);
    const long long g = r / nsample;          // (cloud, s)
    const long long cloud = g / m;
    const long long src = cloud * n + idx[r];
    float v;
    if (ch < cpos) {
      const float dx = center[src * 3 + 0] - new_center[g * 3 + 0];
      const float dy = center[src * 3 + 1] - new_center[g * 3 + 1];
      const float dz = center[src * 3 + 2] - new_center[g * 3 + 2];
      if (ch < 3) v = (ch == 0) ? dx : (ch == 1 ? dy : dz);
      else {
        const float rho = sqrtf(rs_sqnorm(dx, dy, dz));
        if (ch == 3) v = rho;
        else if (ch == 4) v = (rho == 0.f) ? 0.f : acosf(dz / rho) / RS_PI_F;
        else v = atan2f(dy, dx) / RS_TWO_PI_F + 0.5f;
      }
    } else {
      v = normal[src * cn + (ch - cpos)];
    }
    out[r * ctot + ch] = v;
  }
}

// Feature channels: out[r, c0 + :] = feature[src(r), :]   VEC floats per thread
template <int VEC>
__global__ void __launch_bounds__(GR_THREADS)
group_tail_kernel(long long rows, int per_cloud_rows, int n, int cf, int c0, int ctot,
                  const float *__restrict__ feature, const int *__restrict__ idx,
                  float *__restrict__ out) {
  const int vpr = cf / VEC;
  const long long total = rows * vpr;
  for (long long e = (long long)blockIdx.x * GR_THREADS + threadIdx.x; e < total;
       e += (long long)gridDim.x * GR_THREADS) {
    const long long r = e / vpr;
    const int v = (int)(e - r * vpr);
    const long long cloud = r / per_cloud_rows;
    const float *s = feature + (cloud * n + idx[r]) * cf + v * VEC;
    float *d = out + r * ctot + c0 + v * VEC;
    if (VEC == 4) *reinterpret_cast<float4 *>(d) = *reinterpret_cast<const float4 *>(s);
    else *d = *s;
  }
}

// backward of the gathered channels [c0, c0+cw) of a (rows, ctot) gradient tile.
// A ball-query row lists its real neighbours once (ascending indices) and then repeats the FIRST one
// up to nsample (classification/modules/pointnet2_utils.py:92-94) — 70-90 % of the slots at the model's radii.
// One thread owns (group, channel): gradients of all slots that point at the first neighbour are summed in
// a register and leave as ONE atomic; only the remaining distinct neighbours cost an atomic each.
__global__ void __launch_bounds__(GR_THREADS)
group_scatter_kernel(long long groups, int nsample, int groups_per_cloud, int n, int cw, int c0, int ctot,
                     const float *__restrict__ grad_out, const int *__restrict__ idx,
                     float *__restrict__ grad_src) {
  const long long total = groups * cw;
  for (long long e = (long long)blockIdx.x * GR_THREADS + threadIdx.x; e < total;
       e += (long long)gridDim.x * GR_THREADS) {
    const long long g = e / cw;
    const int ch = (int)(e - g * cw);
    const long long cloud = g / groups_per_cloud;
    const int *row = idx + g * nsample;
    const float *go = grad_out + g * nsample * ctot + c0 + ch;
    float *dst = grad_src + cloud * n * cw + ch;
    const int first = row[0];
    float acc = go[0];
    for (int k = 1; k < nsample; ++k) {
      const int p = row[k];
      const float v = go[(long long)k * ctot];
      if (p == first) acc += v;
      else atomicAdd(dst + (long long)p * cw, v);
    }
    atomicAdd(dst + (long long)first * cw, acc);
  }
}

// group_all: row (cloud, j) = [center, polar(center)?, normal, feature]
__global__ void __launch_bounds__(GR_THREADS)
group_all_kernel(long long rows, int cn, int cf, int cpos, int ctot, const float *__restrict__ center,
                 const float *__restrict__ normal, const float *__restrict__ feature,
                 float *__restrict__ out) {
  const long long total = rows * ctot;
  for (long long e = (long long)blockIdx.x * GR_THREADS + threadIdx.x; e < total;
       e += (long long)gridDim.x * GR_THREADS) {
    const long long r = e / ctot;
    const int ch = (int)(e - r * ctot);
    float v;
    if (ch < cpos) {
      const float x = center[r * 3 + 0], y = center[r * 3 + 1], z = center[r * 3 + 2];
      if (ch < 3) v = (ch == 0) ? x : (ch == 1 ? y : z);
      else {
        const float rho = sqrtf(rs_sqnorm(x, y, z));
        if (ch == 3) v = rho;
        else if (ch == 4) v = (rho == 0.f) ? 0.f : acosf(z / rho) / RS_PI_F;
        else v = atan2f(y, x) / RS_TWO_PI_F + 0.5f;
      }
    } else if (ch < cpos + cn) {
      v = normal[r * cn + (ch - cpos)];
    } else {
      v = feature[r * cf + (ch - cpos - cn)];
    }
    out[e] = v;
  }
}

inline int grid_for(long long work_items) {
  long long blocks = (work_items + GR_THREADS - 1) / GR_THREADS;
  const long long cap = 256LL * 8;     // 8 workgroups per CU, grid-stride beyond that
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

}  // namespace

extern "C" int rs_gather_rows(int b, int n, int m, int c, const float *points, const int *idx,
                              float *out, void *stream) {
  RS_REQUIRE(b >= 0 && n >= 0 && m >= 0 && c >= 0, "rs_gather_rows: negative size");
  const long long rows = (long long)b * m;
  if (rows == 0 || c == 0) return RS_OK;
  RS_REQUIRE(points && idx && out, "rs_gather_rows: null pointer");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(rows * c)), dim3(GR_THREADS), 0, (hipStream_t)stream,
                     rows, m, n, c, points, idx, out);
  RS_CHECK_LAUNCH("rs_gather_rows");
  return RS_OK;
}

extern "C" int rs_gather_rows_backward(int b, int n, int m, int c, const float *grad_out, const int *idx,
                                       float *grad_points, void *stream) {
  RS_REQUIRE(b >= 0 && n >= 0 && m >= 0 && c >= 0, "rs_gather_rows_backward: negative size");
  const long long rows = (long long)b * m;
  if (rows == 0 || c == 0) return RS_OK;
  RS_REQUIRE(grad_out && idx && grad_points, "rs_gather_rows_backward: null pointer");
  hipLaunchKernelGGL(gather_rows_bwd_kernel, dim3(grid_for(rows * c)), dim3(GR_THREADS), 0,
                     (hipStream_t)stream, rows, m, n, c, grad_out, idx, grad_points);
  RS_CHECK_LAUNCH("rs_gather_rows_backward");
  return RS_OK;
}

extern "C" int rs_group_rows(int b, int n, int m, int nsample, int c, const float *points,
                             const int *idx, float *out, void *stream) {
  return rs_gather_rows(b, n, m * nsample, c, points, idx, out, stream);
}
extern "C" int rs_group_rows_backward(int b, int n, int m, int nsample, int c, const float *grad_out,
                                      const int *idx, float *grad_points, void *stream) {
  return rs_gather_rows_backward(b, n, m * nsample, c, grad_out, idx, grad_points, stream);
}

extern "C" int rs_group_features(int b, int n, int m, int nsample, int cn, int cf, int polar,
                                 const float *center, const float *new_center, const float *normal,
                                 const float *feature, const int *idx, float *out, void *stream) {
  RS_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 0 && cn >= 0 && cf >= 0, "rs_group_features: negative size");
  const long long rows = (long long)b * m * nsample;
  if (rows == 0) return RS_OK;
  RS_REQUIRE(center && new_center && idx && out, "rs_group_features: null pointer");
  RS_REQUIRE(cn == 0 || normal, "rs_group_features: normal is NULL but cn=%d", cn);
  RS_REQUIRE(cf == 0 || feature, "rs_group_features: feature is NULL but cf=%d", cf);
  const int cpos = polar ? 6 : 3, ctot = cpos + cn + cf;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(group_head_kernel, dim3(grid_for(rows * (cpos + cn))), dim3(GR_THREADS), 0, st, rows, m,
                     nsample, n, cn, cpos, ctot, center, new_center, normal, idx, out);
  if (cf > 0) {
    const bool vec = (cf % 4 == 0) && ((cpos + cn) % 4 == 0) && (ctot % 4 == 0);
    if (vec)
      hipLaunchKernelGGL(group_tail_kernel<4>, dim3(grid_for(rows * (cf / 4))), dim3(GR_THREADS), 0, st, rows,
                         m * nsample, n, cf, cpos + cn, ctot, feature, idx, out);
    else
      hipLaunchKernelGGL(group_tail_kernel<1>, dim3(grid_for(rows * cf)), dim3(GR_THREADS), 0, st, rows,
                         m * nsample, n, cf, cpos + cn, ctot, feature, idx, out);
  }
  RS_CHECK_LAUNCH("rs_group_features");
  return RS_OK;
}

extern "C" int rs_group_features_backward(int b, int n, int m, int nsample, int cn, int cf, int polar,
                                          const float *grad_out, const int *idx, float *grad_normal,
                                          float *grad_feature, void *stream) {
  RS_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 0 && cn >= 0 && cf >= 0, "rs_group_features_backward: negative size");
  const long long rows = (long long)b * m * nsample;
  if (rows == 0) return RS_OK;
  RS_REQUIRE(grad_out && idx, "rs_group_features_backward: null pointer");
  const int cpos = polar ? 6 : 3, ctot = cpos + cn + cf;
  hipStream_t st = (hipStream_t)stream;
  const long long groups = (long long)b * m;
  if (grad_normal && cn > 0)
    hipLaunchKernelGGL(group_scatter_kernel, dim3(grid_for(groups * cn)), dim3(GR_THREADS), 0, st, groups, nsample,
                       m, n, cn, cpos, ctot, grad_out, idx, grad_normal);
  if (grad_feature && cf > 0)
    hipLaunchKernelGGL(group_scatter_kernel, dim3(grid_for(groups * cf)), dim3(GR_THREADS), 0, st, groups, nsample,
                       m, n, cf, cpos + cn, ctot, grad_out, idx, grad_feature);
  RS_CHECK_LAUNCH("rs_group_features_backward");
  return RS_OK;
}

extern "C" int rs_group_all_features(int b, int n, int cn, int cf, int polar, const float *center,
                                     const float *normal, const float *feature, float *out, void *stream) {
  RS_REQUIRE(b >= 0 && n >= 0 && cn >= 0 && cf >= 0, "rs_group_all_features: negative size");
  const long long rows = (long long)b * n;
  if (rows == 0) return RS_OK;
  RS_REQUIRE(center && out, "rs_group_all_features: null pointer");
  RS_REQUIRE(cn == 0 || normal, "rs_group_all_features: normal is NULL but cn=%d", cn);
  RS_REQUIRE(cf == 0 || feature, "rs_group_all_features: feature is NULL but cf=%d", cf);
  const int cpos = polar ? 6 : 3, ctot = cpos + cn + cf;
  hipLaunchKernelGGL(group_all_kernel, dim3(grid_for(rows * ctot)), dim3(GR_THREADS), 0, (hipStream_t)stream,
                     rows, cn, cf, cpos, ctot, center, normal, feature, out);
  RS_CHECK_LAUNCH("rs_group_all_features");
  return RS_OK;
}
