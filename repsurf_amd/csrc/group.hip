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
gather_rows_bwd_kernel(long long rows_arg, const int *__restrict__ rows_dev, int per_cloud, int n, int c, const float *__restrict__ grad_out,
                       const int *__restrict__ idx, float *__restrict__ grad_points) {
  const long long rows = rows_dev ? min(rows_arg, (long long)*rows_dev) : rows_arg;
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
group_head_kernel(long long rows, int m, int nsample, int n, int cn, int cpos, int pw, int ldo,
                  const float *__restrict__ center, const float *__restrict__ new_center,
                  const float *__restrict__ normal, const int *__restrict__ idx,
                  float *__restrict__ out) {
  const int ch_head = pw + cn;                 // [offset / polar (cpos) | zero padding up to pw | normal (cn)]
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
    } else if (ch < pw) {
      v = 0.f;
    } else {
      v = normal[src * cn + (ch - pw)];
    }
    out[r * ldo + ch] = v;
  }
}

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
group_scatter_kernel(long long groups_arg, const int *__restrict__ groups_dev, int nsample, int groups_per_cloud, int n, int cw, int c0, int ctot,
                     const float *__restrict__ grad_out, const int *__restrict__ idx,
                     float *__restrict__ grad_src) {
  const long long groups = groups_dev ? min(groups_arg, (long long)*groups_dev) : groups_arg;      // (groups beyond a device count are not read)
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

// ---- compacted groups ---------------------------------------------------------------------------
// A ball-query row holds cnt[g] distinct neighbours followed by nsample - cnt[g] copies of the first one
// (classification/modules/pointnet2_utils.py:92-94).  The shared MLP maps identical rows to identical
// outputs, so the grouped operand is built for the DISTINCT slots only; slot 0 carries the multiplicity
// nsample - cnt + 1 for the BatchNorm sums.  Rows of group g live at [offsets[g], offsets[g+1]); the total
// stays on the device (offsets[groups]) and is read by the consuming kernels, never by the host.

// single-workgroup exclusive scan: out[i] = sum_{j<i} in[j], out[n] = total
__global__ void __launch_bounds__(1024)
exclusive_scan_kernel(int n, const int *__restrict__ in, int *__restrict__ out) {
  __shared__ int part[1024];
  const int t = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int lo = min(n, t * per), hi = min(n, lo + per);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += in[i];
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {        // Hillis-Steele inclusive scan of the 1024 partials
    const int v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - s;                             // exclusive prefix of this thread's chunk
  for (int i = lo; i < hi; ++i) { out[i] = run; run += in[i]; }
  if (t == 1023) out[n] = part[1023];
}

// Every thread owns 16 consecutive counts of a 16384-count block (four 16-byte loads issued together), scans them in
// registers, the wave scans its 64 totals with shuffles and the 16 wave totals go through LDS -- one barrier and one memory
// latency per block instead of the twenty barriers / sixteen latencies of the generic kernel (22 us -> ~3 us at 16384
// counts; the scan sits between ball query and the first grouped GEMM on the critical path).  Longer inputs (configs[4]:
// 32768 groups; the cell histogram of a whole scene) walk the blocks with a running carry.
__global__ void __launch_bounds__(1024)
exclusive_scan16_kernel(int n, const int *__restrict__ in, int *__restrict__ out) {
  __shared__ int wsum[2][16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int carry = 0;
  for (int b0 = 0, it = 0; b0 < n; b0 += 16384, ++it) {
    const int base = b0 + t * 16;
    int v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = base + 4 * q;
      if (i + 4 <= n) {
        const int4 x = *reinterpret_cast<const int4 *>(in + i);
        v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * q + e] = (i + e < n) ? in[i + e] : 0;
      }
    }
    int tot = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) { const int x = v[e]; v[e] = tot; tot += x; }     // exclusive inside the thread
    int inc = tot;                                                                  // inclusive scan over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(inc, off, 64);
      if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[it & 1][wave] = inc;       // two buffers: the next block's totals cannot overtake this block's readers
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int x = wsum[it & 1][w]; if (w < wave) wbase += x; total += x; }
    const int pre = carry + wbase + inc - tot;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = base + 4 * q;
      if (i + 4 <= n) {
        *reinterpret_cast<int4 *>(out + i) = make_int4(pre + v[4 * q], pre + v[4 * q + 1], pre + v[4 * q + 2], pre + v[4 * q + 3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (i + e < n) out[i + e] = pre + v[4 * q + e];
      }
    }
    carry += total;
  }
  if (t == 0) out[n] = carry;
}

__global__ void __launch_bounds__(GR_THREADS)
compact_index_kernel(long long groups, int nsample, int groups_per_cloud, int n, const int *__restrict__ idx,
                     const int *__restrict__ cnt, const int *__restrict__ offsets, int *__restrict__ grp,
                     int *__restrict__ slot, int *__restrict__ src, float *__restrict__ mult) {
  const long long total = groups * nsample;
  for (long long e = (long long)blockIdx.x * GR_THREADS + threadIdx.x; e < total; e += (long long)gridDim.x * GR_THREADS) {
    const long long g = e / nsample;
    const int k = (int)(e - g * nsample);
    const int c = cnt[g];
    if (k >= c) continue;
    const long long u = offsets[g] + k;
    grp[u] = (int)g;
    slot[u] = k;
    src[u] = (int)((g / groups_per_cloud) * n + idx[e]);
    mult[u] = (k == 0) ? (float)(nsample - c + 1) : 1.f;
  }
}

// X[u, :] = [center[src]-new_center[g] (3), polar (3)?, normal[src] (cn), feature[src] (cf)] for u < *rows_dev
// fps_idx != NULL: the same launch also gathers the centres' own normal rows, new_normal[g, :] = normal[cloud(g) * n + fps_idx[g], :]
// (index_points(normal, fps_idx), repsurface_utils.py:31) -- one launch less on the critical path per stage
__global__ void __launch_bounds__(GR_THREADS)
compact_features_kernel(const int *__restrict__ rows_dev, int cn, int cf, int cpos, int ctot,
                        const float *__restrict__ center, const float *__restrict__ new_center,
                        const float *__restrict__ normal, const float *__restrict__ feature,
                        const int *__restrict__ grp, const int *__restrict__ src, float *__restrict__ out,
                        long long groups, int m, int n, const int *__restrict__ fps_idx, float *__restrict__ new_normal) {
  const long long total = (long long)(*rows_dev) * ctot;
  if (fps_idx) {
    const long long extra = groups * cn;
    for (long long e = (long long)blockIdx.x * GR_THREADS + threadIdx.x; e < extra; e += (long long)gridDim.x * GR_THREADS) {
      const long long g = e / cn;
      const int ch = (int)(e - g * cn);
      new_normal[e] = normal[((g / m) * n + fps_idx[g]) * cn + ch];
    }
  }
  for (long long e = (long long)blockIdx.x * GR_THREADS + threadIdx.x; e < total; e += (long long)gridDim.x * GR_THREADS) {
    const long long u = e / ctot;
    const int ch = (int)(e - u * ctot);
    const long long sp = src[u];
    float v;
    if (ch < cpos) {
      const long long g = grp[u];
      const float dx = center[sp * 3 + 0] - new_center[g * 3 + 0];
      const float dy = center[sp * 3 + 1] - new_center[g * 3 + 1];
      const float dz = center[sp * 3 + 2] - new_center[g * 3 + 2];
      if (ch < 3) v = (ch == 0) ? dx : (ch == 1 ? dy : dz);
      else {
        const float rho = sqrtf(rs_sqnorm(dx, dy, dz));
        if (ch == 3) v = rho;
        else if (ch == 4) v = (rho == 0.f) ? 0.f : acosf(dz / rho) / RS_PI_F;
        else v = atan2f(dy, dx) / RS_TWO_PI_F + 0.5f;
      }
    } else if (ch < cpos + cn) {
      v = normal[sp * cn + (ch - cpos)];
    } else {
      v = feature[sp * cf + (ch - cpos - cn)];
    }
    out[e] = v;
  }
}

// ---- the gather with four channels per thread, the scatter with four elements in flight (round 4) ------------------------------------------------------------------
// One element per thread meant one integer division, one dependent index load and one 4-byte access per float: the 48 k x 144 rows of
// the second stage took 13 grid-stride trips per thread, each a chain of two memory latencies -- 32 us for 28 MB (1 TB/s) forward,
// 40 us backward.  Here a thread owns a 16-byte chunk of a row: the feature part is one float4 load + one float4 store (forward) or
// one float4 load + four atomics (backward); the first four chunks (position, polar, normal channels) are assembled by the same
// scalar expressions as before (identical values).  Needs 16-byte aligned rows: (cpos + cn) % 4 == 0 and cf % 4 == 0.
__device__ __forceinline__ float head_channel(int ch, int cpos, int cn, long long sp, long long g, const float *__restrict__ center,
                                              const float *__restrict__ new_center, const float *__restrict__ normal) {
  if (ch >= cpos) return normal[sp * cn + (ch - cpos)];
  const float dx = center[sp * 3 + 0] - new_center[g * 3 + 0];
  const float dy = center[sp * 3 + 1] - new_center[g * 3 + 1];
  const float dz = center[sp * 3 + 2] - new_center[g * 3 + 2];
  if (ch < 3) return (ch == 0) ? dx : (ch == 1 ? dy : dz);
  const float rho = sqrtf(rs_sqnorm(dx, dy, dz));
  if (ch == 3) return rho;
  if (ch == 4) return (rho == 0.f) ? 0.f : acosf(dz / rho) / RS_PI_F;
  return atan2f(dy, dx) / RS_TWO_PI_F + 0.5f;
}

__global__ void __launch_bounds__(GR_THREADS)
compact_features4_kernel(const int *__restrict__ rows_dev, int cn, int cf, int cpos, int ctot,
                         const float *__restrict__ center, const float *__restrict__ new_center,
                         const float *__restrict__ normal, const float *__restrict__ feature,
                         const int *__restrict__ grp, const int *__restrict__ src, float *__restrict__ out,
                         long long groups, int m, int n, const int *__restrict__ fps_idx, float *__restrict__ new_normal) {
  const int q = ctot >> 2, qh = (cpos + cn) >> 2;                 // chunks per row, of which the head
  const long long total = (long long)(*rows_dev) * q;
  if (fps_idx) {
    const long long extra = groups * cn;
    for (long long e = (long long)blockIdx.x * GR_THREADS + threadIdx.x; e < extra; e += (long long)gridDim.x * GR_THREADS) {
      const long long g = e / cn;
      const int ch = (int)(e - g * cn);
      new_normal[e] = normal[((g / m) * n + fps_idx[g]) * cn + ch];
    }
  }
  for (long long e = (long long)blockIdx.x * GR_THREADS + threadIdx.x; e < total; e += (long long)gridDim.x * GR_THREADS) {
    const long long u = e / q;
    const int j = (int)(e - u * q);
    const long long sp = src[u];
    float4 v;
    if (j >= qh) {
      v = *reinterpret_cast<const float4 *>(feature + sp * cf + 4 * (j - qh));
    } else {
      const long long g = grp[u];
      v.x = head_channel(4 * j + 0, cpos, cn, sp, g, center, new_center, normal);
      v.y = head_channel(4 * j + 1, cpos, cn, sp, g, center, new_center, normal);
      v.z = head_channel(4 * j + 2, cpos, cn, sp, g, center, new_center, normal);
      v.w = head_channel(4 * j + 3, cpos, cn, sp, g, center, new_center, normal);
    }
    reinterpret_cast<float4 *>(out)[e] = v;
  }
}

// grad_normal[src[u], :] += grad_out[u, cpos : cpos+cn], grad_feature[src[u], :] += grad_out[u, cpos+cn : ctot] -- one atomic per
// distinct neighbour and channel, both tensors in ONE launch; fps_idx != NULL: also the backward of the centre-row gather
// of compact_features_kernel, grad_normal[cloud(g) * n + fps_idx[g], :] += grad_new_normal[g * ldg + :]
// Backward: still ONE float per lane and atomic instruction -- consecutive lanes add to consecutive addresses, a wave's atomic covers two
// full cache lines (a float4 per lane put every line under four separate atomic instructions: measured 1.5x slower) -- but FOUR
// independent (index, value) pairs per trip, so a thread's 13 grid-stride trips of two dependent memory latencies become 4.
__global__ void __launch_bounds__(GR_THREADS)
compact_scatter4_kernel(const int *__restrict__ rows_dev, int cn, int cf, int cpos, int ctot, const float *__restrict__ grad_out,
                        const int *__restrict__ src, float *__restrict__ grad_normal, float *__restrict__ grad_feature,
                        long long groups, int m, int n, const int *__restrict__ fps_idx, const float *__restrict__ grad_new_normal,
                        long long ldg) {
  const int c0 = grad_normal ? 0 : cn, cw = (grad_normal ? cn : 0) + (grad_feature ? cf : 0);     // channels [c0, c0 + cw) behind cpos
  const long long total = (long long)(*rows_dev) * cw, step = (long long)gridDim.x * GR_THREADS;
  for (long long e = (long long)blockIdx.x * GR_THREADS + threadIdx.x; e < total; e += 4 * step) {
    long long sp[4];
    int ch[4];
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long ei = e + i * step, ec = ei < total ? ei : e;
      const long long u = ec / cw;
      ch[i] = c0 + (int)(ec - u * cw);
      v[i] = grad_out[u * ctot + cpos + ch[i]];
      sp[i] = src[u];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (e + i * step >= total) continue;
      if (ch[i] < cn) atomicAdd(grad_normal + sp[i] * cn + ch[i], v[i]);
      else atomicAdd(grad_feature + sp[i] * cf + (ch[i] - cn), v[i]);
    }
  }
  if (fps_idx) {
    const long long extra = groups * cn;
    for (long long e = (long long)blockIdx.x * GR_THREADS + threadIdx.x; e < extra; e += (long long)gridDim.x * GR_THREADS) {
      const long long g = e / cn;
      const int ch = (int)(e - g * cn);
      atomicAdd(grad_normal + ((g / m) * n + fps_idx[g]) * cn + ch, grad_new_normal[g * ldg + ch]);
    }
  }
}

// ---- the compacted grouping's backward as a GATHER (round 4) ---------------------------------------------------------------------
// The scatter above is one atomic per distinct neighbour and channel into zero-filled targets: 6.6 M atomics + a fill for the second
// stage of the classification step, 20 us per launch.  Which compacted rows name which source point is geometry: compact_csr_kernel
// (geometry stage, one workgroup per cloud: LDS histogram of src -> scan -> fill, lists sorted so that the sums below have a fixed
// order) inverts `src` into csr_off (points + 1) / csr_rows (the compacted rows of point p: csr_rows[csr_off[p] .. csr_off[p + 1])),
// and centre_of[p] = the group whose centre p is (or -1).  The backward then reads, per source point and channel, the rows that
// name it and writes every element of the targets exactly once: no atomics, no fill, bit-reproducible gradients.
constexpr int CSR_THREADS = 1024;

__global__ void __launch_bounds__(CSR_THREADS)
compact_csr_kernel(int n, int m, const int *__restrict__ src, const int *__restrict__ offsets, const int *__restrict__ fps_idx,
                   int *__restrict__ csr_off, int *__restrict__ centre_of, int *__restrict__ csr_rows) {
  extern __shared__ int cnt[];                       // n counters, then n cursors
  __shared__ int wsum[CSR_THREADS / 64];
  int *cur = cnt + n;
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = offsets[(long long)c * m], r1 = offsets[(long long)(c + 1) * m];       // the compacted rows of this cloud
  const long long p0 = (long long)c * n;
  for (int p = tid; p < n; p += CSR_THREADS) { cnt[p] = 0; centre_of[p0 + p] = -1; }
  __syncthreads();
  for (int u = r0 + tid; u < r1; u += CSR_THREADS) atomicAdd(&cnt[src[u] - (int)p0], 1);
  __syncthreads();
  // exclusive scan of the n counters: consecutive chunks per thread, wave scan, wave totals through LDS
  const int per = (n + CSR_THREADS - 1) / CSR_THREADS;
  int run = 0;
  for (int k = 0; k < per; ++k) { const int p = tid * per + k; if (p < n) run += cnt[p]; }
  int inc = run;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = inc - run;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  for (int k = 0; k < per; ++k) {
    const int p = tid * per + k;
    if (p < n) { const int v = cnt[p]; cur[p] = base; csr_off[p0 + p] = r0 + base; base += v; }
  }
  if (c == (int)gridDim.x - 1 && tid == 0) csr_off[p0 + n] = r1;
  __syncthreads();
  for (int u = r0 + tid; u < r1; u += CSR_THREADS) {
    const int pos = atomicAdd(&cur[src[u] - (int)p0], 1);
    csr_rows[r0 + pos] = u;
  }
  __syncthreads();      // (global writes of this workgroup, read back by this workgroup below)
  // every list in ascending row order: the sums of the backward then have one order, whatever the atomics' order was
  for (int p = tid; p < n; p += CSR_THREADS) {
    const int lo = r0 + (cur[p] - cnt[p]), hi = r0 + cur[p];
    for (int i = lo + 1; i < hi; ++i) {
      const int v = csr_rows[i];
      int j = i - 1;
      while (j >= lo && csr_rows[j] > v) { csr_rows[j + 1] = csr_rows[j]; --j; }
      csr_rows[j + 1] = v;
    }
  }
  // centre_of[p]: the group whose centre p is, -1 for none.  FPS repeats a row when a cloud holds fewer distinct points than it is asked
  // to pick (padded clouds): the point is then the centre of SEVERAL groups, every one of which hands it a centre-row gradient.  The
  // lowest such group g wins (atomicMin: deterministic) and the entry becomes -2 - g, which tells the backward to look the others up in
  // fps_idx (ascending: a fixed summation order).  (one racing writer used to win and the other groups' gradients were lost.)
  if (fps_idx) {
    __syncthreads();
    for (int p = tid; p < n; p += CSR_THREADS) { cnt[p] = 0; cur[p] = 0x7fffffff; }       // (the counters / cursors are spent: shared flags, lowest group)
    __syncthreads();
    for (int g = tid; g < m; g += CSR_THREADS) atomicMin(&cur[fps_idx[(long long)c * m + g]], c * m + g);
    __syncthreads();
    for (int g = tid; g < m; g += CSR_THREADS) { const int p = fps_idx[(long long)c * m + g]; if (cur[p] != c * m + g) cnt[p] = 1; }
    __syncthreads();
    for (int p = tid; p < n; p += CSR_THREADS)
      if (cur[p] != 0x7fffffff) centre_of[p0 + p] = cnt[p] ? -2 - cur[p] : cur[p];
  }
}

// The same inversion for ANY gather of packed batches (round 4): edge e reads source row src[e]; the edges of cloud c are
// [per * edge_ends[c-1], per * edge_ends[c]) (`per` edges per query row: nsample of a grouping, 3 of an interpolation), its source
// rows [point_ends[c-1], point_ends[c]).  csr_off (P + 1), csr_edges (E): the edges that read each source row, ascending.
__global__ void __launch_bounds__(CSR_THREADS)
inverse_index_kernel(int per, const int *__restrict__ src, const int *__restrict__ edge_ends, const int *__restrict__ point_ends,
                     int lds_points, int *__restrict__ csr_off, int *__restrict__ csr_edges, int *__restrict__ overflow) {
  extern __shared__ int cnt[];                       // lds_points counters, then lds_points cursors
  __shared__ int wsum[CSR_THREADS / 64];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p0 = c ? point_ends[c - 1] : 0, n = point_ends[c] - p0;
  const int e0 = (c ? edge_ends[c - 1] : 0) * per, e1 = edge_ends[c] * per;
  if (n > lds_points) {                              // (workgroup-uniform) a cloud the counters cannot hold: the caller falls back
    if (tid == 0) atomicAdd(overflow, 1);
    return;
  }
  int *cur = cnt + lds_points;
  for (int p = tid; p < n; p += CSR_THREADS) cnt[p] = 0;
  __syncthreads();
  for (int e = e0 + tid; e < e1; e += CSR_THREADS) atomicAdd(&cnt[src[e] - p0], 1);
  __syncthreads();
  const int chunk = (n + CSR_THREADS - 1) / CSR_THREADS;
  int run = 0;
  for (int k = 0; k < chunk; ++k) { const int p = tid * chunk + k; if (p < n) run += cnt[p]; }
  int inc = run;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = inc - run;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  for (int k = 0; k < chunk; ++k) {
    const int p = tid * chunk + k;
    if (p < n) { const int v = cnt[p]; cur[p] = base; csr_off[p0 + p] = e0 + base; base += v; }
  }
  if (c == (int)gridDim.x - 1 && tid == 0) csr_off[p0 + n] = e1;
  __syncthreads();
  for (int e = e0 + tid; e < e1; e += CSR_THREADS) {
    const int pos = atomicAdd(&cur[src[e] - p0], 1);
    csr_edges[e0 + pos] = e;
  }
  __syncthreads();
  for (int p = tid; p < n; p += CSR_THREADS) {       // ascending lists: one summation order, whatever the atomics' order was
    const int lo = e0 + (cur[p] - cnt[p]), hi = e0 + cur[p];
    for (int i = lo + 1; i < hi; ++i) {
      const int v = csr_edges[i];
      int j = i - 1;
      while (j >= lo && csr_edges[j] > v) { csr_edges[j + 1] = csr_edges[j]; --j; }
      csr_edges[j + 1] = v;
    }
  }
}

__global__ void __launch_bounds__(GR_THREADS)
compact_gather_bwd_kernel(long long points, int cn, int cf, int cpos, int ctot, const float *__restrict__ grad_out,
                          const int *__restrict__ csr_off, const int *__restrict__ csr_rows, const int *__restrict__ centre_of,
                          float *__restrict__ grad_normal, float *__restrict__ grad_feature,
                          const float *__restrict__ grad_new_normal, long long ldg, const int *__restrict__ fps_idx, int n, int m) {
  const int c0 = grad_normal ? 0 : cn, cw = (grad_normal ? cn : 0) + (grad_feature ? cf : 0);     // channels [c0, c0 + cw) behind cpos
  const long long total = points * cw;
  for (long long e = (long long)blockIdx.x * GR_THREADS + threadIdx.x; e < total; e += (long long)gridDim.x * GR_THREADS) {
    const long long p = e / cw;
    const int ch = c0 + (int)(e - p * cw);
    const int lo = csr_off[p], hi = csr_off[p + 1];
    const float *col = grad_out + cpos + ch;
    float acc = 0.f;
    int j = lo;
    for (; j + 4 <= hi; j += 4) {                    // four rows in flight; the sum keeps the list's order
      const float v0 = col[(long long)csr_rows[j] * ctot], v1 = col[(long long)csr_rows[j + 1] * ctot];
      const float v2 = col[(long long)csr_rows[j + 2] * ctot], v3 = col[(long long)csr_rows[j + 3] * ctot];
      acc = (((acc + v0) + v1) + v2) + v3;
    }
    for (; j < hi; ++j) acc += col[(long long)csr_rows[j] * ctot];
    if (ch < cn) {
      if (grad_new_normal) {
        const int g = centre_of[p];
        if (g >= 0) acc += grad_new_normal[(long long)g * ldg + ch];
        else if (g < -1 && fps_idx) {                   // the centre of several groups (FPS repeated the row): all of them, ascending
          const int g0 = -2 - g, gend = (g0 / m + 1) * m, pl = (int)(p - (long long)(g0 / m) * n);
          for (int g2 = g0; g2 < gend; ++g2)
            if (fps_idx[g2] == pl) acc += grad_new_normal[(long long)g2 * ldg + ch];
        } else if (g < -1) acc += grad_new_normal[(long long)(-2 - g) * ldg + ch];
      }
      grad_normal[p * cn + ch] = acc;
    } else {
      grad_feature[p * cf + (ch - cn)] = acc;
    }
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

extern "C" int rs_gather_rows_backward_dev(int b, int n, int m, int c, const float *grad_out, const int *idx,
                                           float *grad_points, const int *rows_dev, void *stream) {
  RS_REQUIRE(b >= 0 && n >= 0 && m >= 0 && c >= 0, "rs_gather_rows_backward: negative size");
  RS_REQUIRE(!rows_dev || b == 1, "rs_gather_rows_backward_dev: a device row count needs a packed batch (b = 1)");
  const long long rows = (long long)b * m;
  if (rows == 0 || c == 0) return RS_OK;
  RS_REQUIRE(grad_out && idx && grad_points, "rs_gather_rows_backward: null pointer");
  hipLaunchKernelGGL(gather_rows_bwd_kernel, dim3(grid_for(rows * c)), dim3(GR_THREADS), 0,
                     (hipStream_t)stream, rows, rows_dev, m, n, c, grad_out, idx, grad_points);
  RS_CHECK_LAUNCH("rs_gather_rows_backward");
  return RS_OK;
}

extern "C" int rs_gather_rows_backward(int b, int n, int m, int c, const float *grad_out, const int *idx,
                                       float *grad_points, void *stream) {
  return rs_gather_rows_backward_dev(b, n, m, c, grad_out, idx, grad_points, nullptr, stream);
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
                                 const float *feature, const int *idx, float *out, int pos_pad, int ldo, void *stream) {
  RS_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 0 && cn >= 0 && cf >= 0 && pos_pad >= 0, "rs_group_features: negative size");
  const long long rows = (long long)b * m * nsample;
  if (rows == 0) return RS_OK;
  RS_REQUIRE(center && new_center && idx && out, "rs_group_features: null pointer");
  RS_REQUIRE(cn == 0 || normal, "rs_group_features: normal is NULL but cn=%d", cn);
  RS_REQUIRE(cf == 0 || feature, "rs_group_features: feature is NULL but cf=%d", cf);
  const int cpos = polar ? 6 : 3, pw = cpos + pos_pad, ctot = pw + cn + cf;
  if (ldo <= 0) ldo = ctot;
  RS_REQUIRE(ldo >= ctot, "rs_group_features: row stride %d < %d channels", ldo, ctot);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(group_head_kernel, dim3(grid_for(rows * (pw + cn))), dim3(GR_THREADS), 0, st, rows, m,
                     nsample, n, cn, cpos, pw, ldo, center, new_center, normal, idx, out);
  if (cf > 0) {
    const bool vec = (cf % 4 == 0) && ((pw + cn) % 4 == 0) && (ldo % 4 == 0) && (((uintptr_t)out | (uintptr_t)feature) % 16 == 0);
    if (vec)
      hipLaunchKernelGGL(group_tail_kernel<4>, dim3(grid_for(rows * (cf / 4))), dim3(GR_THREADS), 0, st, rows,
                         m * nsample, n, cf, pw + cn, ldo, feature, idx, out);
    else
      hipLaunchKernelGGL(group_tail_kernel<1>, dim3(grid_for(rows * cf)), dim3(GR_THREADS), 0, st, rows,
                         m * nsample, n, cf, pw + cn, ldo, feature, idx, out);
  }
  RS_CHECK_LAUNCH("rs_group_features");
  return RS_OK;
}

extern "C" int rs_group_features_backward_dev(int, int, int, int, int, int, int, const float *, const int *, float *, float *, int, int, const int *, void *);

extern "C" int rs_group_features_backward(int b, int n, int m, int nsample, int cn, int cf, int polar,
                                          const float *grad_out, const int *idx, float *grad_normal,
                                          float *grad_feature, int pos_pad, int ldo, void *stream) {
  return rs_group_features_backward_dev(b, n, m, nsample, cn, cf, polar, grad_out, idx, grad_normal, grad_feature, pos_pad, ldo, nullptr, stream);
}

extern "C" int rs_group_features_backward_dev(int b, int n, int m, int nsample, int cn, int cf, int polar,
                                              const float *grad_out, const int *idx, float *grad_normal,
                                              float *grad_feature, int pos_pad, int ldo, const int *groups_dev, void *stream) {
  RS_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 0 && cn >= 0 && cf >= 0 && pos_pad >= 0, "rs_group_features_backward: negative size");
  RS_REQUIRE(!groups_dev || b == 1, "rs_group_features_backward_dev: a device group count needs a packed batch (b = 1)");
  const long long rows = (long long)b * m * nsample;
  if (rows == 0) return RS_OK;
  RS_REQUIRE(grad_out && idx, "rs_group_features_backward: null pointer");
  const int pw = (polar ? 6 : 3) + pos_pad, ctot = pw + cn + cf;
  if (ldo <= 0) ldo = ctot;
  RS_REQUIRE(ldo >= ctot, "rs_group_features_backward: row stride %d < %d channels", ldo, ctot);
  hipStream_t st = (hipStream_t)stream;
  const long long groups = (long long)b * m;
  if (grad_normal && cn > 0)
    hipLaunchKernelGGL(group_scatter_kernel, dim3(grid_for(groups * cn)), dim3(GR_THREADS), 0, st, groups, groups_dev, nsample,
                       m, n, cn, pw, ldo, grad_out, idx, grad_normal);
  if (grad_feature && cf > 0)
    hipLaunchKernelGGL(group_scatter_kernel, dim3(grid_for(groups * cf)), dim3(GR_THREADS), 0, st, groups, groups_dev, nsample,
                       m, n, cf, pw + cn, ldo, grad_out, idx, grad_feature);
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

extern "C" int rs_exclusive_scan(int n, const int *in, int *out, void *stream) {
  RS_REQUIRE(n >= 0, "rs_exclusive_scan: negative size");
  RS_REQUIRE(in && out, "rs_exclusive_scan: null pointer");
  if ((((size_t)in | (size_t)out) & 15) == 0)
    hipLaunchKernelGGL(exclusive_scan16_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, in, out);
  else
    hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, in, out);
  RS_CHECK_LAUNCH("rs_exclusive_scan");
  return RS_OK;
}

// offsets (groups + 1), grp / slot / src / mult (capacity = groups * nsample): the bookkeeping of the compacted groups from the
// ball query's (idx, cnt) alone -- no features involved, so a pipelined step builds it in the geometry stage
extern "C" int rs_compact_index(int b, int n, int m, int nsample, const int *idx, const int *cnt, int *offsets, int *grp, int *slot,
                                int *src, float *mult, void *stream) {
  RS_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 0, "rs_compact_index: negative size");
  const long long groups = (long long)b * m;
  RS_REQUIRE(offsets, "rs_compact_index: null pointer");
  int rc = rs_exclusive_scan((int)groups, cnt ? cnt : offsets, offsets, stream);
  if (rc != RS_OK || groups == 0 || nsample == 0) return rc;
  RS_REQUIRE(idx && cnt && grp && slot && src && mult, "rs_compact_index: null pointer");
  hipLaunchKernelGGL(compact_index_kernel, dim3(grid_for(groups * nsample)), dim3(GR_THREADS), 0, (hipStream_t)stream, groups, nsample, m, n,
                     idx, cnt, offsets, grp, slot, src, mult);
  RS_CHECK_LAUNCH("rs_compact_index");
  return RS_OK;
}

// have_index != 0: offsets / grp / slot / src / mult were built by rs_compact_index (only the feature rows are written here).
// fps_idx, new_normal (optional, both or neither): new_normal (b*m, cn) = normal rows of the centres themselves, same launch.
extern "C" int rs_group_features_compact(int b, int n, int m, int nsample, int cn, int cf, int polar,
                                         const float *center, const float *new_center, const float *normal,
                                         const float *feature, const int *idx, const int *cnt, const int *offsets,
                                         float *out, float *mult, int *grp, int *slot, int *src, int have_index,
                                         const int *fps_idx, float *new_normal, void *stream) {
  RS_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 0 && cn >= 0 && cf >= 0, "rs_group_features_compact: negative size");
  const long long groups = (long long)b * m;
  if (groups == 0 || nsample == 0) return RS_OK;
  RS_REQUIRE(center && new_center && idx && cnt && offsets && out && mult && grp && slot && src, "rs_group_features_compact: null pointer");
  RS_REQUIRE(cn == 0 || normal, "rs_group_features_compact: normal is NULL but cn=%d", cn);
  RS_REQUIRE(cf == 0 || feature, "rs_group_features_compact: feature is NULL but cf=%d", cf);
  RS_REQUIRE((fps_idx == nullptr) == (new_normal == nullptr), "rs_group_features_compact: fps_idx and new_normal come together");
  const int cpos = polar ? 6 : 3, ctot = cpos + cn + cf;
  hipStream_t st = (hipStream_t)stream;
  if (!have_index)
    hipLaunchKernelGGL(compact_index_kernel, dim3(grid_for(groups * nsample)), dim3(GR_THREADS), 0, st, groups, nsample, m, n,
                       idx, cnt, offsets, grp, slot, src, mult);
  const bool vec4 = ((cpos + cn) & 3) == 0 && (cf & 3) == 0 && (((uintptr_t)out | (uintptr_t)feature) & 15) == 0;
  if (vec4)
    hipLaunchKernelGGL(compact_features4_kernel, dim3(grid_for(groups * nsample * (ctot / 4) / 4 + 1)), dim3(GR_THREADS), 0, st,
                       offsets + groups, cn, cf, cpos, ctot, center, new_center, normal, feature, grp, src, out,
                       groups, m, n, cn > 0 ? fps_idx : nullptr, new_normal);
  else
    hipLaunchKernelGGL(compact_features_kernel, dim3(grid_for(groups * nsample * ctot / 4 + 1)), dim3(GR_THREADS), 0, st,
                       offsets + groups, cn, cf, cpos, ctot, center, new_center, normal, feature, grp, src, out,
                       groups, m, n, cn > 0 ? fps_idx : nullptr, new_normal);
  RS_CHECK_LAUNCH("rs_group_features_compact");
  return RS_OK;
}

// fps_idx / grad_new_normal (optional, both or neither; rows ldg floats apart): backward of the centre-row gather, same launch.
extern "C" int rs_group_features_compact_backward(long long capacity, const int *rows_dev, int cn, int cf, int polar,
                                                  const float *grad_out, const int *src, float *grad_normal,
                                                  float *grad_feature, int b, int n, int m, const int *fps_idx,
                                                  const float *grad_new_normal, long long ldg, void *stream) {
  RS_REQUIRE(capacity >= 0 && cn >= 0 && cf >= 0, "rs_group_features_compact_backward: negative size");
  if (capacity == 0) return RS_OK;
  RS_REQUIRE(rows_dev && grad_out && src, "rs_group_features_compact_backward: null pointer");
  RS_REQUIRE((fps_idx == nullptr) == (grad_new_normal == nullptr), "rs_group_features_compact_backward: fps_idx and grad_new_normal come together");
  RS_REQUIRE(!fps_idx || (grad_normal && ldg >= cn && b >= 0 && m >= 0), "rs_group_features_compact_backward: the centre rows add into grad_normal");
  const int cpos = polar ? 6 : 3, ctot = cpos + cn + cf;
  if (cn == 0) grad_normal = nullptr;
  if (cf == 0) grad_feature = nullptr;
  if (!grad_normal && !grad_feature) return RS_OK;
  const int cw = (grad_normal ? cn : 0) + (grad_feature ? cf : 0);
  hipLaunchKernelGGL(compact_scatter4_kernel, dim3(grid_for(capacity * cw / 16 + 1)), dim3(GR_THREADS), 0, (hipStream_t)stream, rows_dev, cn, cf,
                     cpos, ctot, grad_out, src, grad_normal, grad_feature, (long long)b * m, m, n, fps_idx, grad_new_normal, ldg);
  RS_CHECK_LAUNCH("rs_group_features_compact_backward");
  return RS_OK;
}

/* The inverse of `src` (rs_compact_index), for the gather form of the backward: csr_off (b*n + 1), csr_rows (capacity),
 * centre_of (b*n; the group whose centre a point is, -1 otherwise; fps_idx may be NULL: all -1).  Geometry only. */
extern "C" int rs_compact_csr(int b, int n, int m, const int *src, const int *offsets, const int *fps_idx, int *csr_off,
                              int *centre_of, int *csr_rows, void *stream) {
  RS_REQUIRE(b >= 0 && n >= 0 && m >= 0, "rs_compact_csr: negative size");
  if (b == 0 || n == 0) return RS_OK;
  RS_REQUIRE(src && offsets && csr_off && centre_of && csr_rows, "rs_compact_csr: null pointer");
  RS_REQUIRE(n <= 16384, "rs_compact_csr: %d points per cloud exceed the 16 384 the workgroup's counters hold", n);
  RS_REQUIRE_LDS((size_t)2 * n * sizeof(int), "rs_compact_csr");
  hipLaunchKernelGGL(compact_csr_kernel, dim3(b), dim3(CSR_THREADS), (size_t)2 * n * sizeof(int), (hipStream_t)stream, n, m, src,
                     offsets, fps_idx, csr_off, centre_of, csr_rows);
  RS_CHECK_LAUNCH("rs_compact_csr");
  return RS_OK;
}

/* rs_group_features_compact_backward without atomics: every element of grad_normal (b*n, cn) / grad_feature (b*n, cf) is WRITTEN
 * (no zero fill by the caller) as the sum over the compacted rows that name the point, in ascending row order, plus -- with
 * grad_new_normal -- the centre row's gradient of the group the point is the centre of. */
extern "C" int rs_group_features_compact_backward_csr(int b, int n, int cn, int cf, int polar, const float *grad_out,
                                                      const int *csr_off, const int *csr_rows, const int *centre_of,
                                                      float *grad_normal, float *grad_feature, const float *grad_new_normal,
                                                      long long ldg, const int *fps_idx, int m, void *stream) {
  RS_REQUIRE(b >= 0 && n >= 0 && cn >= 0 && cf >= 0, "rs_group_features_compact_backward_csr: negative size");
  RS_REQUIRE(!grad_new_normal || (fps_idx && m > 0), "rs_group_features_compact_backward_csr: the centre rows need fps_idx (b, m)");
  if (b == 0 || n == 0) return RS_OK;
  RS_REQUIRE(grad_out && csr_off && csr_rows && centre_of, "rs_group_features_compact_backward_csr: null pointer");
  RS_REQUIRE(!grad_new_normal || (grad_normal && ldg >= cn), "rs_group_features_compact_backward_csr: the centre rows add into grad_normal");
  const int cpos = polar ? 6 : 3, ctot = cpos + cn + cf;
  if (cn == 0) grad_normal = nullptr;
  if (cf == 0) grad_feature = nullptr;
  if (!grad_normal && !grad_feature) return RS_OK;
  const int cw = (grad_normal ? cn : 0) + (grad_feature ? cf : 0);
  const long long points = (long long)b * n;
  hipLaunchKernelGGL(compact_gather_bwd_kernel, dim3(grid_for(points * cw)), dim3(GR_THREADS), 0, (hipStream_t)stream, points, cn, cf,
                     cpos, ctot, grad_out, csr_off, csr_rows, centre_of, grad_normal, grad_feature, grad_new_normal, ldg, fps_idx, n, m);
  RS_CHECK_LAUNCH("rs_group_features_compact_backward_csr");
  return RS_OK;
}

/* Inverse of a gather index over a packed batch (round 4): src (E = per * edge_ends[b-1] edges, global source rows), edge_ends /
 * point_ends (b): running ends of the query rows (x per = edges) and of the source rows per cloud -> csr_off (P + 1), csr_edges (E):
 * the edges reading each source row, ascending.  overflow (1 int, zeroed by the caller): the number of clouds with more source rows
 * than `max_points` (the counters of one workgroup: <= 16384) -- their part of the output is then undefined: use the scatter. */
extern "C" int rs_inverse_index(int b, int per, int max_points, const int *src, const int *edge_ends, const int *point_ends,
                                int *csr_off, int *csr_edges, int *overflow, void *stream) {
  RS_REQUIRE(b >= 0 && per > 0, "rs_inverse_index: bad size");
  if (b == 0) return RS_OK;
  RS_REQUIRE(src && edge_ends && point_ends && csr_off && csr_edges && overflow, "rs_inverse_index: null pointer");
  RS_REQUIRE(max_points > 0 && max_points <= 16384, "rs_inverse_index: max_points=%d outside 1..16384", max_points);
  RS_REQUIRE_LDS((size_t)2 * max_points * sizeof(int), "rs_inverse_index");
  hipLaunchKernelGGL(inverse_index_kernel, dim3(b), dim3(CSR_THREADS), (size_t)2 * max_points * sizeof(int), (hipStream_t)stream, per, src,
                     edge_ends, point_ends, max_points, csr_off, csr_edges, overflow);
  RS_CHECK_LAUNCH("rs_inverse_index");
  return RS_OK;
}

/* rs_group_features_backward as a gather over rs_inverse_index of the grouping index (b = 1 packed layout: `points` source rows):
 * grad_normal (points, cn) / grad_feature (points, cf) WRITTEN as the sums over the grouped rows that read each point, ascending --
 * no zero fill, no atomics.  c0: first gathered column of a grouped row (cpos + pad), ldo: its row pitch. */
extern "C" int rs_group_features_backward_csr(long long points, int cn, int cf, int c0, int ldo, const float *grad_out,
                                              const int *csr_off, const int *csr_edges, float *grad_normal, float *grad_feature,
                                              void *stream) {
  RS_REQUIRE(points >= 0 && cn >= 0 && cf >= 0 && c0 >= 0 && ldo >= c0 + cn + cf, "rs_group_features_backward_csr: bad size");
  if (points == 0) return RS_OK;
  RS_REQUIRE(grad_out && csr_off && csr_edges, "rs_group_features_backward_csr: null pointer");
  if (cn == 0) grad_normal = nullptr;
  if (cf == 0) grad_feature = nullptr;
  if (!grad_normal && !grad_feature) return RS_OK;
  const int cw = (grad_normal ? cn : 0) + (grad_feature ? cf : 0);
  hipLaunchKernelGGL(compact_gather_bwd_kernel, dim3(grid_for(points * cw)), dim3(GR_THREADS), 0, (hipStream_t)stream, points, cn, cf,
                     c0, ldo, grad_out, csr_off, csr_edges, (const int *)nullptr, grad_normal, grad_feature, (const float *)nullptr, 0LL,
                     (const int *)nullptr, 0, 1);
  RS_CHECK_LAUNCH("rs_group_features_backward_csr");
  return RS_OK;
}
