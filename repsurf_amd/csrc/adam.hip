// adam.hip — torch.optim.Adam step over a list of fp32 parameter tensors in one launch per 80 tensors (gfx950).
//
// The reference trains with torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=decay_rate)
// (classification/tool/train_cls_scanobjectnn.py:179-185).  The model has ~70 small parameter tensors (1.5 M floats);
// the framework's multi-tensor Adam spends ~75 us on them (two launches whose workgroups each take one 64 K chunk of
// one tensor).  The update is 7 streams x 6 MB = 42 MB of HBM traffic, ~8 us of bandwidth: here a workgroup takes
// 2048 consecutive elements of one tensor (two 16-B loads per stream per thread), the table of tensors rides in the
// kernel arguments, and hyper-parameters / step count live in device memory so a captured hipGraph follows the
// learning-rate schedule and counts its own replays.
#include "rs_common.h"
#include <math.h>

namespace {

constexpr int AD_THREADS = 256;
constexpr int AD_CHUNK = 2048;            // elements per workgroup

struct AdamTable {
  float *p[RS_ADAM_MAX];
  const float *g[RS_ADAM_MAX];
  float *m[RS_ADAM_MAX];
  float *v[RS_ADAM_MAX];
  int n[RS_ADAM_MAX];
  int blk_end[RS_ADAM_MAX];               // running number of workgroups up to and including tensor i
  int count;
};

static_assert(sizeof(AdamTable) + 32 <= 4096, "the table rides in the kernel arguments (4 KB)");

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, float wd, float omb1, float b2, float omb2,
                                         float eps, float step_size, float bc2_sqrt) {
  g = fmaf(wd, p, g);                              // grad.add(param, alpha=weight_decay)
  m = fmaf(g - m, omb1, m);                        // exp_avg.lerp_(grad, 1 - beta1)
  v = fmaf(omb2, g * g, v * b2);                   // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p = fmaf(-step_size, m / denom, p);              // param.addcdiv_(exp_avg, denom, value=-step_size)
}

__global__ void __launch_bounds__(AD_THREADS)
adam_kernel(AdamTable T, double *hyper, int *__restrict__ step, int *__restrict__ done, int advance) {
  // the tensor of this workgroup: first entry whose running block count exceeds blockIdx.x.  A bisection: every probe is a
  // dependent scalar load from the kernel-argument segment, and the big tensors (the head, the last stage) sit at the END of the
  // table -- the linear walk cost their workgroups ~70 round trips before the first operand load was issued.
  int lo = 0, hi = T.count - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((int)blockIdx.x >= T.blk_end[mid]) lo = mid + 1; else hi = mid;
  }
  const int ti = lo;
  const int blk0 = ti ? T.blk_end[ti - 1] : 0;
  const int base = ((int)blockIdx.x - blk0) * AD_CHUNK;
  const int n = T.n[ti];
  float *__restrict__ p = T.p[ti];
  const float *__restrict__ g = T.g[ti];
  float *__restrict__ m = T.m[ti];
  float *__restrict__ v = T.v[ti];

  // the operands are requested first: the bias-correction powers below (fp64, ~300 instructions) run under the loads
  constexpr int U = AD_CHUNK / (AD_THREADS * 4);
  const bool vec = ((((size_t)p | (size_t)g | (size_t)m | (size_t)v) & 15) == 0);
  float4 pv[U], gv[U], mv[U], vv[U];
  bool full[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e = base + (u * AD_THREADS + (int)threadIdx.x) * 4;
    full[u] = vec && e + 4 <= n;
    if (full[u]) {
      pv[u] = *reinterpret_cast<const float4 *>(p + e); gv[u] = *reinterpret_cast<const float4 *>(g + e);
      mv[u] = *reinterpret_cast<const float4 *>(m + e); vv[u] = *reinterpret_cast<const float4 *>(v + e);
    }
  }
  // hyper-parameters are doubles, like the Python floats the framework derives its scalars from: 1 - beta2 formed in
  // fp32 from a rounded beta2 = 0.999f is off by 5e-5 relative.
  const double lr = hyper[0], beta1 = hyper[1], beta2 = hyper[2];
  const float eps = (float)hyper[3], wd = (float)hyper[4];
  const float b2 = (float)beta2, omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
  const int t = *step + 1;                          // this update's step number (1-based)
  // bias corrections 1 - beta1^t and sqrt(1 - beta2^t): ~330 fp64 instructions per THREAD for two numbers the whole launch
  // shares (a thread's own update is ~100 fp32 instructions).  The last workgroup of a step leaves the pair for step t + 1 in
  // hyper[5..9] = {t + 1, 1 - beta1^(t+1), sqrt(1 - beta2^(t+1)), beta1, beta2}; a launch that finds its own step number and betas
  // there (a wave-uniform test) uses them, any other (first step, changed betas, loaded state) computes them as before.
  double denom1, s2;
  if (hyper[5] == (double)t && hyper[8] == beta1 && hyper[9] == beta2) { denom1 = hyper[6]; s2 = hyper[7]; }
  else { denom1 = 1.0 - pow(beta1, (double)t); s2 = sqrt(1.0 - pow(beta2, (double)t)); }
  const float step_size = (float)(lr / denom1);
  const float bc2_sqrt = (float)s2;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e = base + (u * AD_THREADS + (int)threadIdx.x) * 4;
    if (e >= n) break;
    if (full[u]) {
      adam_one(pv[u].x, gv[u].x, mv[u].x, vv[u].x, wd, omb1, b2, omb2, eps, step_size, bc2_sqrt);
      adam_one(pv[u].y, gv[u].y, mv[u].y, vv[u].y, wd, omb1, b2, omb2, eps, step_size, bc2_sqrt);
      adam_one(pv[u].z, gv[u].z, mv[u].z, vv[u].z, wd, omb1, b2, omb2, eps, step_size, bc2_sqrt);
      adam_one(pv[u].w, gv[u].w, mv[u].w, vv[u].w, wd, omb1, b2, omb2, eps, step_size, bc2_sqrt);
      *reinterpret_cast<float4 *>(p + e) = pv[u];
      *reinterpret_cast<float4 *>(m + e) = mv[u];
      *reinterpret_cast<float4 *>(v + e) = vv[u];
    } else {
      for (int i = e; i < min(e + 4, n); ++i) {
        float ps = p[i], ms = m[i], vs = v[i];
        adam_one(ps, g[i], ms, vs, wd, omb1, b2, omb2, eps, step_size, bc2_sqrt);
        p[i] = ps; m[i] = ms; v[i] = vs;
      }
    }
  }
  // the last workgroup to finish advances the step counter (every workgroup has read it by then: a workgroup's read of *step
  // feeds its stores, which precede the barrier and the counter increment).  No __threadfence(): at agent scope it is an L2
  // write-back + invalidate (buffer_wbl2 / buffer_inv) -- 771 of them made the launch 33 us for 41 MB -- and nothing written
  // here is read before the next launch.
  if (advance) {
    __syncthreads();
    if (threadIdx.x == 0) {
      if (atomicAdd(done, 1) == (int)gridDim.x - 1) {
        *done = 0; *step = t;
        hyper[6] = 1.0 - pow(beta1, (double)(t + 1)); hyper[7] = sqrt(1.0 - pow(beta2, (double)(t + 1)));
        hyper[8] = beta1; hyper[9] = beta2;
        hyper[5] = (double)(t + 1);
      }
    }
  }
}

}  // namespace

extern "C" int rs_adam_step(const rs_adam_table *t, double *hyper, int *step, int *done, int advance, void *stream) {
  RS_REQUIRE(t && hyper && step && done, "rs_adam_step: null pointer");
  RS_REQUIRE(t->count > 0 && t->count <= RS_ADAM_MAX, "rs_adam_step: count=%d (1..%d)", t->count, RS_ADAM_MAX);
  AdamTable T;
  int blocks = 0;
  for (int i = 0; i < t->count; ++i) {
    RS_REQUIRE(t->p[i] && t->g[i] && t->m[i] && t->v[i] && t->n[i] > 0, "rs_adam_step: tensor %d: null pointer or empty", i);
    T.p[i] = t->p[i]; T.g[i] = t->g[i]; T.m[i] = t->m[i]; T.v[i] = t->v[i]; T.n[i] = t->n[i];
    blocks += rs_cdiv(t->n[i], AD_CHUNK);
    T.blk_end[i] = blocks;
  }
  T.count = t->count;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(AD_THREADS), 0, (hipStream_t)stream, T, hyper, step, done, advance);
  RS_CHECK_LAUNCH("rs_adam_step");
  return RS_OK;
}

// ---- device wall-clock stamp (measurement aid, round 6) -----------------------------------------------------------------------------
// HIP events cannot be recorded inside a replayed hipGraph on this runtime (hipEventRecordWithFlags(.., hipEventRecordExternal) under
// capture: hipErrorInvalidValue, tools/probe_graph_events.py), so bench.py brackets a launch INSIDE the step's replayed graph with two
// of these one-thread launches: each stores the constant-rate wall clock (s_memrealtime, rs_timestamp_khz() ticks per millisecond).
namespace {
__global__ void stamp_kernel(long long *dst) {
  if (threadIdx.x == 0) *dst = (long long)wall_clock64();
}
}  // namespace

extern "C" int rs_timestamp(long long *dst, void *stream) {
  RS_REQUIRE(dst, "rs_timestamp: null pointer");
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dst);
  RS_CHECK_LAUNCH("rs_timestamp");
  return RS_OK;
}

extern "C" int rs_timestamp_khz(void) {
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) return 100000;   // gfx9: 100 MHz
  return khz;
}
