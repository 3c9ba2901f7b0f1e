// interp.hip — three nearest neighbours + inverse-distance interpolation (gfx950).
//
// rs_three_nn / rs_three_interpolate(+backward) replace the `nearestneighbor` / `interpolation`
// operators of the reference (classification/modules/pointops/src/interpolation/
// interpolation_cuda_kernel.cu:134-195, 90-114; Python side pointops.py:86-147) on channels-last
// data.  Distances use direct differences ((dx*dx + dy*dy) + dz*dz, products rounded
// separately); a candidate replaces a kept neighbour only when strictly closer, so equal
// distances keep the lower index -- the order the reference's sequential scan produces.
#include "rs_common.h"
#include <math.h>

namespace {

constexpr int IT_THREADS = 256;
constexpr int IT_TILE = 2048;

__global__ void __launch_bounds__(IT_THREADS)
three_nn_kernel(int b, int n, int m, int blocks_per_cloud, const float *__restrict__ unknown,
                const float *__restrict__ known, float *__restrict__ dist2, int *__restrict__ idx) {
  __shared__ float4 tile[IT_TILE];
  int cloud, chunk;
  rs_xcd_remap(blockIdx.x, b, blocks_per_cloud, cloud, chunk);
  const int q = chunk * IT_THREADS + threadIdx.x;
  const int qc = min(q, n - 1);
  const float *u = unknown + ((size_t)cloud * n + qc) * 3;
  const float ux = u[0], uy = u[1], uz = u[2];
  const float *kp = known + (size_t)cloud * m * 3;
  float d1 = INFINITY, d2 = INFINITY, d3 = INFINITY;
  int i1 = 0, i2 = 0, i3 = 0;
  for (int t0 = 0; t0 < m; t0 += IT_TILE) {
    const int tn = min(IT_TILE, m - t0);
    __syncthreads();
    for (int p = threadIdx.x; p < tn; p += IT_THREADS)
      tile[p] = make_float4(kp[(t0 + p) * 3 + 0], kp[(t0 + p) * 3 + 1], kp[(t0 + p) * 3 + 2], 0.f);
    __syncthreads();
    for (int p = 0; p < tn; ++p) {
      const float4 c = tile[p];
      const float dx = ux - c.x, dy = uy - c.y, dz = uz - c.z;
      const float d = (dx * dx + dy * dy) + dz * dz;
      const int k = t0 + p;
      if (d < d1) { d3 = d2; i3 = i2; d2 = d1; i2 = i1; d1 = d; i1 = k; }
      else if (d < d2) { d3 = d2; i3 = i2; d2 = d; i2 = k; }
      else if (d < d3) { d3 = d; i3 = k; }
    }
  }
  if (q < n) {
    const size_t o = ((size_t)cloud * n + q) * 3;
    dist2[o + 0] = d1; dist2[o + 1] = d2; dist2[o + 2] = d3;
    idx[o + 0] = i1; idx[o + 1] = i2; idx[o + 2] = i3;
  }
}

// out[b, j, :] = sum_t w[b,j,t] * points[b, idx[b,j,t], :]
// Both kernels take FOUR elements per trip with all index / weight loads of the four first, then the gathers, then the stores /
// atomics: an element is a chain of two dependent round trips (idx -> the rows it names), a thread of the widest decoder level sees
// 32 of them, and one after the other that is what the launch took (46 us for the backward of 65 536 x 256).  32-bit index
// arithmetic while the element count allows (two 64-bit divisions per element otherwise).
__global__ void __launch_bounds__(IT_THREADS)
interp_fwd_kernel(long long rows, int n, int m, int c, const float *__restrict__ points,
                  const int *__restrict__ idx, const float *__restrict__ weight, float *__restrict__ out,
                  const float *__restrict__ add, int relu, const float *__restrict__ ps, const float *__restrict__ pt,
                  const float *__restrict__ as, const float *__restrict__ at) {
  // ps / pt, as / at (round 4, optional): per-channel affine maps of the gathered rows and of `add` -- the BatchNorm of the two
  // Linear layers in front of the interpolation (repsurface_utils.py:256-270) applied on the fly, z = fma(scale, y, shift) as the
  // BatchNorm pass computes it, instead of two passes that materialise the normalised tensors
  // add / relu (round 4): out = relu(interpolated + add) in the same launch -- the feature-propagation stage's skip connection and
  // activation (segmentation/modules/repsurface_utils.py:266-270) were two framework kernels behind this one, 3 + 2 passes over the tensor
  const long long total = rows * c, stride = (long long)gridDim.x * IT_THREADS;
  if (total < (1LL << 31) && rows * 3 < (1LL << 31)) {
    const unsigned tot = (unsigned)total, st = (unsigned)stride, cu = (unsigned)c, nu = (unsigned)n;
    for (unsigned e0 = blockIdx.x * IT_THREADS + threadIdx.x; e0 < tot; e0 += 4 * st) {
      unsigned ch[4], cloud[4]; int id[4][3]; float w[4][3], sk[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned e = e0 + k * st < tot ? e0 + k * st : e0, r = e / cu;
        ch[k] = e - r * cu; cloud[k] = r / nu;
        sk[k] = add ? add[e] : 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) { id[k][t] = idx[r * 3 + t]; w[k][t] = weight[r * 3 + t]; }
      }
      float v[4][3];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float *base = points + (long long)cloud[k] * m * c + ch[k];
#pragma unroll
        for (int t = 0; t < 3; ++t) v[k][t] = base[(long long)id[k][t] * c];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (e0 + k * st < tot) {
          if (ps) {
            const float s_ = ps[ch[k]], t_ = pt[ch[k]];
#pragma unroll
            for (int t = 0; t < 3; ++t) v[k][t] = fmaf(s_, v[k][t], t_);
          }
          float o = (w[k][0] * v[k][0] + w[k][1] * v[k][1]) + w[k][2] * v[k][2];
          if (add) o += as ? fmaf(as[ch[k]], sk[k], at[ch[k]]) : sk[k];
          out[e0 + k * st] = relu ? fmaxf(o, 0.f) : o;
        }
    }
    return;
  }
  for (long long e = (long long)blockIdx.x * IT_THREADS + threadIdx.x; e < total; e += stride) {
    const long long r = e / c;
    const int ch = (int)(e - r * c);
    const long long cloud = r / n;
    const float *base = points + cloud * m * c + ch;
    float z0 = base[(long long)idx[r * 3 + 0] * c], z1 = base[(long long)idx[r * 3 + 1] * c], z2 = base[(long long)idx[r * 3 + 2] * c];
    if (ps) { z0 = fmaf(ps[ch], z0, pt[ch]); z1 = fmaf(ps[ch], z1, pt[ch]); z2 = fmaf(ps[ch], z2, pt[ch]); }
    const float v0 = weight[r * 3 + 0] * z0, v1 = weight[r * 3 + 1] * z1, v2 = weight[r * 3 + 2] * z2;
    float o = (v0 + v1) + v2;
    if (add) o += as ? fmaf(as[ch], add[e], at[ch]) : add[e];
    out[e] = relu ? fmaxf(o, 0.f) : o;
  }
}

__global__ void __launch_bounds__(IT_THREADS)
interp_bwd_kernel(long long rows_arg, const int *__restrict__ rows_dev, int n, int m, int c, const float *__restrict__ grad_out,
                  const int *__restrict__ idx, const float *__restrict__ weight,
                  float *__restrict__ grad_points, const float *__restrict__ fwd_out, float *__restrict__ grad_add,
                  const float *__restrict__ sy, const float *__restrict__ smean, const float *__restrict__ sinvstd,
                  double *__restrict__ partial, int partial_blocks) {
  // partial != NULL (round 4): the BatchNorm-backward sums of the skip branch -- per channel {sum g, sum g * yhat}, yhat = (sy - mean)
  // * invstd, g the masked gradient -- leave with this pass (partial (partial_blocks, 2, c); the launcher makes the grid stride a
  // multiple of c, c <= 256: a thread meets one channel only), instead of a second pass over g and sy (rs_pool_max_backward)
  __shared__ double red[IT_THREADS][2];
  double acc0 = 0.0, acc1 = 0.0;
  const long long rows = rows_dev ? min(rows_arg, (long long)*rows_dev) : rows_arg;      // (rows beyond a device count: not read, not scattered, not summed)
  // fwd_out != NULL: the forward ended in a ReLU -- the incoming gradient counts where fwd_out > 0; grad_add != NULL: the masked
  // gradient is also written out (the skip connection's gradient), one pass instead of threshold_backward + this kernel
  const long long total = rows * c, stride = (long long)gridDim.x * IT_THREADS;
  if (total < (1LL << 31) && rows * 3 < (1LL << 31)) {
    const unsigned tot = (unsigned)total, st = (unsigned)stride, cu = (unsigned)c, nu = (unsigned)n;
    for (unsigned e0 = blockIdx.x * IT_THREADS + threadIdx.x; e0 < tot; e0 += 4 * st) {
      unsigned ch[4], cloud[4]; int id[4][3]; float w[4][3], g[4]; bool ok[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ok[k] = e0 + k * st < tot;
        const unsigned e = ok[k] ? e0 + k * st : e0, r = e / cu;
        ch[k] = e - r * cu; cloud[k] = r / nu;
        g[k] = grad_out[e];
        if (fwd_out) g[k] = fwd_out[e] > 0.f ? g[k] : 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) { id[k][t] = idx[r * 3 + t]; w[k][t] = weight[r * 3 + t]; }
      }
      float yh[4] = {0.f, 0.f, 0.f, 0.f};
      if (partial) {
#pragma unroll
        for (int k = 0; k < 4; ++k) yh[k] = sy[ok[k] ? e0 + k * st : e0];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (!ok[k]) continue;
        if (grad_add) grad_add[e0 + k * st] = g[k];
        if (partial) { acc0 += (double)g[k]; acc1 += (double)(g[k] * ((yh[k] - smean[ch[k]]) * sinvstd[ch[k]])); }
        if (grad_points) {      // (NULL: only the masked gradient and the sums -- the coarse rows' gradient is gathered, interp_gather_bwd_kernel)
          float *base = grad_points + (long long)cloud[k] * m * c + ch[k];
#pragma unroll
          for (int t = 0; t < 3; ++t) atomicAdd(base + (long long)id[k][t] * c, g[k] * w[k][t]);
        }
      }
    }
    if (partial) {      // threads of one channel meet in LDS (fixed order), one partial row per workgroup
      red[threadIdx.x][0] = acc0; red[threadIdx.x][1] = acc1;
      __syncthreads();
      if ((int)threadIdx.x < c) {
        double t0 = 0.0, t1 = 0.0;
        for (int k = threadIdx.x; k < IT_THREADS; k += c) { t0 += red[k][0]; t1 += red[k][1]; }
        const int chn = (int)((blockIdx.x * (unsigned)IT_THREADS + threadIdx.x) % cu);
        partial[((long long)blockIdx.x * 2 + 0) * c + chn] = t0;
        partial[((long long)blockIdx.x * 2 + 1) * c + chn] = t1;
        for (int pb = blockIdx.x + gridDim.x; pb < partial_blocks; pb += gridDim.x) {      // rows no workgroup owns read as zero
          partial[((long long)pb * 2 + 0) * c + chn] = 0.0;
          partial[((long long)pb * 2 + 1) * c + chn] = 0.0;
        }
      }
    }
    return;
  }
  for (long long e = (long long)blockIdx.x * IT_THREADS + threadIdx.x; e < total; e += stride) {
    const long long r = e / c;
    const int ch = (int)(e - r * c);
    const long long cloud = r / n;
    float *base = grad_points + cloud * m * c + ch;
    float g = grad_out[e];
    if (fwd_out) g = fwd_out[e] > 0.f ? g : 0.f;
    if (grad_add) grad_add[e] = g;
#pragma unroll
    for (int t = 0; t < 3; ++t) atomicAdd(base + (long long)idx[r * 3 + t] * c, g * weight[r * 3 + t]);
  }
}

// The interpolation's backward towards the coarse rows as a GATHER (round 4): csr_off / csr_edges = rs_inverse_index of idx (edge
// e = 3 n + t reads coarse row idx[e]), built with the geometry.  Per coarse row and channel, the weighted sum over the edges that
// read it, ascending -- every element written once: no atomics, no zero fill, one summation order.  g: the masked gradient (N, c)
// when a pass already made it (skip branch), else it is taken as grad_out where fwd_out > 0.  partial != NULL: the BatchNorm-backward
// sums of the coarse branch {sum d, sum d * yhat} leave with this pass (grid stride a multiple of c, c <= 256, as in interp_bwd_kernel).
__global__ void __launch_bounds__(IT_THREADS)
interp_gather_bwd_kernel(unsigned total, int c, const float *__restrict__ g, const float *__restrict__ grad_out,
                         const float *__restrict__ fwd_out, const float *__restrict__ weight, const int *__restrict__ csr_off,
                         const int *__restrict__ csr_edges, float *__restrict__ grad_points, const float *__restrict__ sy,
                         const float *__restrict__ smean, const float *__restrict__ sinvstd, double *__restrict__ partial,
                         int partial_blocks) {
  __shared__ double red[IT_THREADS][2];
  double acc0 = 0.0, acc1 = 0.0;
  const unsigned st = gridDim.x * IT_THREADS, cu = (unsigned)c;
  for (unsigned e0 = blockIdx.x * IT_THREADS + threadIdx.x; e0 < total; e0 += st) {
    const unsigned row = e0 / cu, ch = e0 - row * cu;
    const int lo = csr_off[row], hi = csr_off[row + 1];
    float acc = 0.f;
    int j = lo;
    auto term = [&](int e) {
      const unsigned n = (unsigned)e / 3u;
      const float w = weight[e];
      float v;
      if (g) v = g[n * cu + ch];
      else { v = grad_out[n * cu + ch]; if (fwd_out) v = fwd_out[n * cu + ch] > 0.f ? v : 0.f; }
      return w * v;
    };
    for (; j + 4 <= hi; j += 4) {                    // four edges in flight, the sum keeps the list's order
      const float t0 = term(csr_edges[j]), t1 = term(csr_edges[j + 1]), t2 = term(csr_edges[j + 2]), t3 = term(csr_edges[j + 3]);
      acc = (((acc + t0) + t1) + t2) + t3;
    }
    for (; j < hi; ++j) acc += term(csr_edges[j]);
    grad_points[e0] = acc;
    if (partial) { acc0 += (double)acc; acc1 += (double)(acc * ((sy[e0] - smean[ch]) * sinvstd[ch])); }
  }
  if (partial) {
    red[threadIdx.x][0] = acc0; red[threadIdx.x][1] = acc1;
    __syncthreads();
    if ((int)threadIdx.x < c) {
      double t0 = 0.0, t1 = 0.0;
      for (int k = threadIdx.x; k < IT_THREADS; k += c) { t0 += red[k][0]; t1 += red[k][1]; }
      const int chn = (int)((blockIdx.x * (unsigned)IT_THREADS + threadIdx.x) % cu);
      partial[((long long)blockIdx.x * 2 + 0) * c + chn] = t0;
      partial[((long long)blockIdx.x * 2 + 1) * c + chn] = t1;
      for (int pb = blockIdx.x + gridDim.x; pb < partial_blocks; pb += gridDim.x) {
        partial[((long long)pb * 2 + 0) * c + chn] = 0.0;
        partial[((long long)pb * 2 + 1) * c + chn] = 0.0;
      }
    }
  }
}

inline int grid_for(long long work_items) {
  long long blocks = (work_items + IT_THREADS - 1) / IT_THREADS;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

}  // namespace

extern "C" int rs_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                           int *idx, void *stream) {
  RS_REQUIRE(b >= 0 && n >= 0 && m >= 0, "rs_three_nn: negative size");
  if (b == 0 || n == 0) return RS_OK;
  RS_REQUIRE(m >= 3, "rs_three_nn: needs at least 3 known points (m=%d)", m);
  RS_REQUIRE(unknown && known && dist2 && idx, "rs_three_nn: null pointer");
  const int bpc = rs_cdiv(n, IT_THREADS);
  hipLaunchKernelGGL(three_nn_kernel, dim3(b * bpc), dim3(IT_THREADS), 0, (hipStream_t)stream, b, n, m, bpc,
                     unknown, known, dist2, idx);
  RS_CHECK_LAUNCH("rs_three_nn");
  return RS_OK;
}

extern "C" int rs_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                                    const float *weight, float *out, void *stream) {
  RS_REQUIRE(b >= 0 && c >= 0 && m >= 0 && n >= 0, "rs_three_interpolate: negative size");
  const long long rows = (long long)b * n;
  if (rows == 0 || c == 0) return RS_OK;
  RS_REQUIRE(points && idx && weight && out, "rs_three_interpolate: null pointer");
  hipLaunchKernelGGL(interp_fwd_kernel, dim3(grid_for(rows * c)), dim3(IT_THREADS), 0, (hipStream_t)stream,
                     rows, n, m, c, points, idx, weight, out, (const float *)nullptr, 0, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr);
  RS_CHECK_LAUNCH("rs_three_interpolate");
  return RS_OK;
}

extern "C" int rs_three_interpolate_fused(int b, int c, int m, int n, const float *points, const int *idx, const float *weight,
                                          const float *add, int relu, float *out, void *stream) {
  RS_REQUIRE(b >= 0 && c >= 0 && m >= 0 && n >= 0, "rs_three_interpolate_fused: negative size");
  const long long rows = (long long)b * n;
  if (rows == 0 || c == 0) return RS_OK;
  RS_REQUIRE(points && idx && weight && out, "rs_three_interpolate_fused: null pointer");
  hipLaunchKernelGGL(interp_fwd_kernel, dim3(grid_for(rows * c)), dim3(IT_THREADS), 0, (hipStream_t)stream,
                     rows, n, m, c, points, idx, weight, out, add, relu, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr);
  RS_CHECK_LAUNCH("rs_three_interpolate_fused");
  return RS_OK;
}

extern "C" int rs_three_interpolate_fused_backward_dev(int b, int c, int n, int m, const float *grad_out, const float *fwd_out,
                                                       const int *idx, const float *weight, float *grad_points, float *grad_add,
                                                       const int *rows_dev, void *stream) {
  RS_REQUIRE(b >= 0 && c >= 0 && m >= 0 && n >= 0, "rs_three_interpolate_fused_backward: negative size");
  RS_REQUIRE(!rows_dev || b == 1, "rs_three_interpolate_fused_backward_dev: a device row count needs a packed batch (b = 1)");
  const long long rows = (long long)b * n;
  if (rows == 0 || c == 0) return RS_OK;
  RS_REQUIRE(grad_out && idx && weight && grad_points, "rs_three_interpolate_fused_backward: null pointer");
  hipLaunchKernelGGL(interp_bwd_kernel, dim3(grid_for(rows * c)), dim3(IT_THREADS), 0, (hipStream_t)stream,
                     rows, rows_dev, n, m, c, grad_out, idx, weight, grad_points, fwd_out, grad_add, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, (double *)nullptr, 0);
  RS_CHECK_LAUNCH("rs_three_interpolate_fused_backward");
  return RS_OK;
}

extern "C" int rs_three_interpolate_fused_backward(int b, int c, int n, int m, const float *grad_out, const float *fwd_out,
                                                   const int *idx, const float *weight, float *grad_points, float *grad_add,
                                                   void *stream) {
  return rs_three_interpolate_fused_backward_dev(b, c, n, m, grad_out, fwd_out, idx, weight, grad_points, grad_add, nullptr, stream);
}

extern "C" int rs_three_interpolate_backward(int b, int c, int n, int m, const float *grad_out,
                                             const int *idx, const float *weight, float *grad_points,
                                             void *stream) {
  RS_REQUIRE(b >= 0 && c >= 0 && m >= 0 && n >= 0, "rs_three_interpolate_backward: negative size");
  const long long rows = (long long)b * n;
  if (rows == 0 || c == 0) return RS_OK;
  RS_REQUIRE(grad_out && idx && weight && grad_points, "rs_three_interpolate_backward: null pointer");
  hipLaunchKernelGGL(interp_bwd_kernel, dim3(grid_for(rows * c)), dim3(IT_THREADS), 0, (hipStream_t)stream,
                     rows, (const int *)nullptr, n, m, c, grad_out, idx, weight, grad_points, (const float *)nullptr, (float *)nullptr, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, (double *)nullptr, 0);
  RS_CHECK_LAUNCH("rs_three_interpolate_backward");
  return RS_OK;
}


/* Feature propagation, first layers (segmentation/modules/repsurface_utils.py:256-270): out = relu(interpolate(BN_f(points)) +
 * BN_s(add)) with both BatchNorms given as per-channel (scale, shift) and applied on the fly to the raw Linear outputs `points`
 * (b, m, c) and `add` (b, n, c) -- z = fma(scale, y, shift), the arithmetic of the BatchNorm pass it replaces. */
extern "C" int rs_three_interpolate_affine(int b, int c, int m, int n, const float *points, const float *pscale, const float *pshift,
                                           const int *idx, const float *weight, const float *add, const float *ascale,
                                           const float *ashift, int relu, float *out, void *stream) {
  RS_REQUIRE(b >= 0 && c >= 0 && m >= 0 && n >= 0, "rs_three_interpolate_affine: negative size");
  const long long rows = (long long)b * n;
  if (rows == 0 || c == 0) return RS_OK;
  RS_REQUIRE(points && idx && weight && out, "rs_three_interpolate_affine: null pointer");
  RS_REQUIRE((pscale == nullptr) == (pshift == nullptr) && (ascale == nullptr) == (ashift == nullptr), "rs_three_interpolate_affine: scale and shift come together");
  RS_REQUIRE(!ascale || add, "rs_three_interpolate_affine: an affine map of `add` needs `add`");
  hipLaunchKernelGGL(interp_fwd_kernel, dim3(grid_for(rows * c)), dim3(IT_THREADS), 0, (hipStream_t)stream,
                     rows, n, m, c, points, idx, weight, out, add, relu, pscale, pshift, ascale, ashift);
  RS_CHECK_LAUNCH("rs_three_interpolate_affine");
  return RS_OK;
}

/* Its backward: g = grad_out * (fwd_out > 0) is written to grad_add (the gradient at BN_s's output = at the interpolation's sum),
 * scattered with the weights into grad_points (zeroed by the caller: the gradient at BN_f's output), and BN_s's backward sums
 * {sum g, sum g * (add - mean) * invstd} per channel leave as partial (partial_blocks, 2, c) doubles (c <= 256). */
extern "C" int rs_three_interpolate_affine_backward(int b, int c, int n, int m, const float *grad_out, const float *fwd_out,
                                                    const int *idx, const float *weight, float *grad_points, float *grad_add,
                                                    const float *add, const float *add_mean, const float *add_invstd,
                                                    double *partial, int partial_blocks, const int *rows_dev, void *stream) {
  RS_REQUIRE(b >= 0 && c >= 0 && m >= 0 && n >= 0, "rs_three_interpolate_affine_backward: negative size");
  RS_REQUIRE(!rows_dev || b == 1, "rs_three_interpolate_affine_backward: a device row count needs a packed batch (b = 1)");
  const long long rows = (long long)b * n;
  if (rows == 0 || c == 0) return RS_OK;
  RS_REQUIRE(grad_out && idx && weight && grad_add && add && add_mean && add_invstd && partial && partial_blocks > 0,
             "rs_three_interpolate_affine_backward: null pointer (grad_points may be NULL: rs_three_interpolate_backward_csr gathers it)");
  RS_REQUIRE(c <= IT_THREADS && rows * c < (1LL << 31) && rows * 3 < (1LL << 31),
             "rs_three_interpolate_affine_backward: c=%d / %lld elements outside the fused-sums form (c <= 256, < 2^31 elements)", c, rows * c);
  int g = grid_for(rows * c);
  if (g > partial_blocks) g = partial_blocks;
  int a = c, bb = IT_THREADS;                      // grid * 256 must be a multiple of c: a thread then meets one channel only
  while (bb) { const int t = a % bb; a = bb; bb = t; }
  const int step = c / a;
  g = g / step * step;
  RS_REQUIRE(g >= step && g >= 1, "rs_three_interpolate_affine_backward: partial_blocks=%d below %d", partial_blocks, step);
  hipLaunchKernelGGL(interp_bwd_kernel, dim3(g), dim3(IT_THREADS), 0, (hipStream_t)stream, rows, rows_dev, n, m, c, grad_out, idx, weight,
                     grad_points, fwd_out, grad_add, add, add_mean, add_invstd, partial, partial_blocks);
  RS_CHECK_LAUNCH("rs_three_interpolate_affine_backward");
  return RS_OK;
}

/* The gradient of the interpolated rows (m_rows, c) as a gather over rs_inverse_index of idx (per = 3): grad_points WRITTEN (no zero
 * fill, no atomics, ascending sums).  g (n, c): the masked gradient if a pass made it (rs_three_interpolate_affine_backward with
 * grad_points = NULL), else NULL: grad_out where fwd_out > 0 (fwd_out NULL: no ReLU).  partial (optional): the BatchNorm-backward sums
 * of the rows' BatchNorm {sum d, sum d * (y - mean) * invstd} with y (m_rows, c) their raw Linear output (c <= 256). */
extern "C" int rs_three_interpolate_backward_csr(long long m_rows, int c, const float *g, const float *grad_out, const float *fwd_out,
                                                 const float *weight, const int *csr_off, const int *csr_edges, float *grad_points,
                                                 const float *y, const float *mean, const float *invstd, double *partial,
                                                 int partial_blocks, void *stream) {
  RS_REQUIRE(m_rows >= 0 && c >= 0, "rs_three_interpolate_backward_csr: negative size");
  if (m_rows == 0 || c == 0) return RS_OK;
  RS_REQUIRE((g || grad_out) && weight && csr_off && csr_edges && grad_points, "rs_three_interpolate_backward_csr: null pointer");
  RS_REQUIRE(m_rows * c < (1LL << 31), "rs_three_interpolate_backward_csr: more than 2^31 elements");
  int gr = grid_for(m_rows * c);
  if (partial) {
    RS_REQUIRE(y && mean && invstd && partial_blocks > 0 && c <= IT_THREADS, "rs_three_interpolate_backward_csr: the sums need y, mean, invstd and c <= 256");
    if (gr > partial_blocks) gr = partial_blocks;
    int a = c, bb = IT_THREADS;
    while (bb) { const int t = a % bb; a = bb; bb = t; }
    const int step = c / a;
    gr = gr / step * step;
    RS_REQUIRE(gr >= step && gr >= 1, "rs_three_interpolate_backward_csr: partial_blocks=%d below %d", partial_blocks, step);
  }
  hipLaunchKernelGGL(interp_gather_bwd_kernel, dim3(gr), dim3(IT_THREADS), 0, (hipStream_t)stream, (unsigned)(m_rows * c), c, g, grad_out,
                     fwd_out, weight, csr_off, csr_edges, grad_points, y, mean, invstd, partial, partial_blocks);
  RS_CHECK_LAUNCH("rs_three_interpolate_backward_csr");
  return RS_OK;
}
