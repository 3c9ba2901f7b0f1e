// head.hip — the classifier head of the RepSurf-U networks on a batch of <= 64 rows (gfx950).
//
//   classfier = Linear(C0,C1)-BN1d-ReLU-Dropout(p)-Linear(C1,C2)-BN1d-ReLU-Dropout(p)-Linear(C2,classes), log_softmax
//   (classification/models/repsurf/repsurf_ssg_umb.py:32-41,56-57) and SmoothClsLoss (classification/util/utils.py:55-69).
//
// With B = 32 rows the head is 0.04 GFLOP but ~45 framework kernels forward + backward (6 % of the training step).
// Here a workgroup OWNS a few output columns of a layer for ALL rows: the BatchNorm statistics of a column (over the
// batch rows) are then local to the workgroup, so Linear + BatchNorm(train) + ReLU + Dropout is ONE kernel, and the
// backward of a layer (data gradient from the next layer, Dropout/ReLU/BatchNorm backward, weight gradient) is ONE
// kernel as well.  Seven launches replace the 45.
//
// Dropout: the reference uses the framework's device generator, which no other implementation can reproduce bit for
// bit (parity tests run with dropout disabled, like SURVEY.md §8a row 12 prescribes).  The mask here is a counter-based
// hash of (seed, step counter, layer, element); the step counter lives on the device and is advanced by the forward
// output kernel, so a replayed hipGraph draws fresh masks every step.  Backward recomputes the mask (nothing stored).
#include "rs_common.h"
#include <math.h>

namespace {

constexpr int HD_THREADS = 256;
constexpr int HD_CW = 4;          // columns owned by a workgroup
constexpr int HD_RB = 32;         // rows per register block
constexpr int HD_KC = 256;        // reduction chunk staged in LDS
constexpr int HD_MAXR = 64;

__device__ __forceinline__ unsigned hd_hash(unsigned seed, unsigned step, unsigned layer, unsigned e) {
  unsigned x = e * 0x9E3779B1u ^ (seed + step * 0x7F4A7C15u + layer * 0x94D049BBu);
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
// keep-mask of Dropout(p): keep when the 24-bit uniform is >= p
__device__ __forceinline__ bool hd_keep(unsigned seed, unsigned step, unsigned layer, unsigned e, float p) {
  return (float)(hd_hash(seed, step, layer, e) >> 8) * (1.0f / 16777216.0f) >= p;
}

// y[r][c] = sum_k x[r][k] * w[n0 + c][k] for the HD_CW owned columns and all R rows -> ys[r][c] in LDS.
// x chunks are staged transposed (xs[k][r], conflict-free), thread (r, ks) walks 32 k of the chunk.
__device__ __forceinline__ void hd_owned_dot(int R, int K, int ncols, int n0, const float *__restrict__ x, int ldx,
                                             const float *__restrict__ w, int ldw, float *xs, float *ws, float *red,
                                             float *ys) {
  const int tid = threadIdx.x, r = tid & 31, ks = tid >> 5;
  for (int rb = 0; rb < R; rb += HD_RB) {
    float acc[HD_CW];
#pragma unroll
    for (int c = 0; c < HD_CW; ++c) acc[c] = 0.f;
    for (int k0 = 0; k0 < K; k0 += HD_KC) {
      __syncthreads();
#pragma unroll
      for (int rr = 0; rr < HD_RB; ++rr) {                        // coalesced: thread t loads column k0 + t of row rb + rr
        const int k = k0 + tid, row = rb + rr;
        xs[tid * (HD_RB + 1) + rr] = (k < K && row < R) ? x[(long long)row * ldx + k] : 0.f;
      }
#pragma unroll
      for (int c = 0; c < HD_CW; ++c) {
        const int k = k0 + tid;
        ws[c * HD_KC + tid] = (k < K && n0 + c < ncols) ? w[(long long)(n0 + c) * ldw + k] : 0.f;
      }
      __syncthreads();
#pragma unroll 8
      for (int kk = 0; kk < 32; ++kk) {
        const int k = ks * 32 + kk;
        const float xv = xs[k * (HD_RB + 1) + r];
#pragma unroll
        for (int c = 0; c < HD_CW; ++c) acc[c] = fmaf(xv, ws[c * HD_KC + k], acc[c]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < HD_CW; ++c) red[(ks * HD_RB + r) * HD_CW + c] = acc[c];
    __syncthreads();
    if (tid < HD_RB * HD_CW) {
      const int rr = tid / HD_CW, c = tid % HD_CW;
      float t = 0.f;
#pragma unroll
      for (int s = 0; s < 8; ++s) t += red[(s * HD_RB + rr) * HD_CW + c];
      if (rb + rr < R) ys[(rb + rr) * HD_CW + c] = t;
    }
  }
  __syncthreads();
}

struct HeadLayer {
  const float *x; int ldx;             // layer input (R, K)
  const float *w; const float *b;      // (N, K), (N)
  const float *gamma, *beta;           // BatchNorm affine (N)
  float *running_mean, *running_var;   // updated in place when not NULL
  float momentum, eps, drop_p;
  float *y;                            // (R, N) pre-BatchNorm output (bias included), kept for backward
  float *h;                            // (R, N) layer output after BN, ReLU, Dropout
  float *mean, *invstd;                // (N) batch statistics, kept for backward
  unsigned seed; const int *step; int layer;
  int R, K, N;
};

__global__ void __launch_bounds__(HD_THREADS)
head_layer_fwd_kernel(HeadLayer L) {
  __shared__ float xs[HD_KC * (HD_RB + 1)];
  __shared__ float ws[HD_CW * HD_KC];
  __shared__ float red[8 * HD_RB * HD_CW];
  __shared__ float ys[HD_MAXR * HD_CW];
  const int n0 = blockIdx.x * HD_CW;
  hd_owned_dot(L.R, L.K, L.N, n0, L.x, L.ldx, L.w, L.K, xs, ws, red, ys);
  // one wave per owned column, lane = row: the batch statistics are a wave reduction
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = n0 + wave;
  if (n >= L.N) return;
  const bool rok = lane < L.R;
  const float v = rok ? ys[lane * HD_CW + wave] + L.b[n] : 0.f;
  const float mean = rs_wave_sum_f32(v) / (float)L.R;
  const float d = rok ? v - mean : 0.f;
  const float var = rs_wave_sum_f32(d * d) / (float)L.R;
  const float invstd = 1.0f / sqrtf(var + L.eps);
  if (lane == 0) {
    L.mean[n] = mean; L.invstd[n] = invstd;
    if (L.running_mean) {       // nn.BatchNorm1d: running = (1-m) running + m batch, variance unbiased
      L.running_mean[n] = (1.f - L.momentum) * L.running_mean[n] + L.momentum * mean;
      const float unb = L.R > 1 ? var * (float)L.R / (float)(L.R - 1) : var;
      L.running_var[n] = (1.f - L.momentum) * L.running_var[n] + L.momentum * unb;
    }
  }
  if (rok) {
    const float z = d * invstd * L.gamma[n] + L.beta[n];
    float a = fmaxf(z, 0.f);
    if (L.drop_p > 0.f)
      a = hd_keep(L.seed, (unsigned)*L.step, (unsigned)L.layer, (unsigned)(lane * L.N + n), L.drop_p) ? a / (1.f - L.drop_p) : 0.f;
    L.y[(long long)lane * L.N + n] = v;
    L.h[(long long)lane * L.N + n] = a;
  }
}

// logits = h . W^T + b, log_softmax over the classes; one workgroup, thread (row r, class j)
struct HeadOut {
  const float *h; const float *w; const float *b;   // (R, K), (classes, K), (classes)
  float *logp;                                       // (R, classes) log-probabilities (the model's output)
  int *step;                                         // device step counter, advanced here (one launch per forward)
  int R, K, classes;
};

__global__ void __launch_bounds__(1024)
head_out_fwd_kernel(HeadOut O) {
  extern __shared__ float sm[];                      // h (R, K+1) | w (classes, K+1) | logits (R, classes)
  const int ldk = O.K + 1;
  float *hs = sm, *wsm = sm + O.R * ldk, *lg = wsm + O.classes * ldk;
  const int tid = threadIdx.x;
  for (int e = tid; e < O.R * O.K; e += blockDim.x) { const int r = e / O.K, k = e - r * O.K; hs[r * ldk + k] = O.h[e]; }
  for (int e = tid; e < O.classes * O.K; e += blockDim.x) { const int j = e / O.K, k = e - j * O.K; wsm[j * ldk + k] = O.w[e]; }
  __syncthreads();
  const int total = O.R * O.classes;
  for (int e = tid; e < total; e += blockDim.x) {
    const int r = e / O.classes, j = e - r * O.classes;
    const float *hr = hs + r * ldk, *wj = wsm + j * ldk;
    float t = 0.f;
#pragma unroll 8
    for (int k = 0; k < O.K; ++k) t = fmaf(hr[k], wj[k], t);
    lg[e] = t + O.b[j];
  }
  __syncthreads();
  for (int r = tid; r < O.R; r += blockDim.x) {
    float mx = -INFINITY;
    for (int j = 0; j < O.classes; ++j) mx = fmaxf(mx, lg[r * O.classes + j]);
    float s = 0.f;
    for (int j = 0; j < O.classes; ++j) s += expf(lg[r * O.classes + j] - mx);
    const float lse = mx + logf(s);
    for (int j = 0; j < O.classes; ++j) O.logp[(long long)r * O.classes + j] = lg[r * O.classes + j] - lse;
  }
  if (tid == 0 && O.step) *O.step += 1;
}

// backward of the output layer: dlogits = dlogp - softmax * sum_j dlogp;  dW3 = dlogits^T h, db3 = colsum(dlogits)
struct HeadOutBwd {
  const float *dlogp, *logp, *h;       // (R, classes), (R, classes), (R, K)
  float *dlogits;                      // (R, classes) out
  float *dw, *db;                      // (classes, K), (classes)
  int R, K, classes;
};

__global__ void __launch_bounds__(1024)
head_out_bwd_kernel(HeadOutBwd O) {
  extern __shared__ float dl[];                      // (R, classes)
  const int tid = threadIdx.x;
  for (int r = tid; r < O.R; r += blockDim.x) {
    float s = 0.f;
    for (int j = 0; j < O.classes; ++j) s += O.dlogp[(long long)r * O.classes + j];
    for (int j = 0; j < O.classes; ++j) {
      const float g = O.dlogp[(long long)r * O.classes + j] - expf(O.logp[(long long)r * O.classes + j]) * s;
      dl[r * O.classes + j] = g;
      O.dlogits[(long long)r * O.classes + j] = g;
    }
  }
  __syncthreads();
  for (int e = tid; e < O.classes * O.K; e += blockDim.x) {
    const int j = e / O.K, k = e - j * O.K;
    float t = 0.f;
    for (int r = 0; r < O.R; ++r) t = fmaf(dl[r * O.classes + j], O.h[(long long)r * O.K + k], t);
    O.dw[e] = t;
  }
  for (int j = tid; j < O.classes; j += blockDim.x) {
    float t = 0.f;
    for (int r = 0; r < O.R; ++r) t += dl[r * O.classes + j];
    O.db[j] = t;
  }
}

// backward of a hidden layer for the owned columns:
//   dh[r][n] = sum_j dz_next[r][j] * w_next[j][n]            (data gradient of the NEXT layer, reduction over its N2 columns)
//   Dropout / ReLU backward (mask recomputed), BatchNorm(train) backward over the rows -> dz[r][n], dgamma, dbeta
//   dW[n][k] = sum_r dz[r][n] * x[r][k]                        (this layer's weight gradient, rows n0..n0+3)
struct HeadLayerBwd {
  const float *dz_next; int n2;        // (R, N2)
  const float *w_next;                 // (N2, N)
  const float *y, *mean, *invstd, *gamma, *beta;
  const float *x; int ldx;             // this layer's input (R, K)
  float *dz;                           // (R, N) out
  float *dw, *dgamma, *dbeta;          // (N, K), (N), (N)
  float drop_p; unsigned seed; const int *step; int layer; int step_back;   // the mask of THIS step: counter - step_back
  int R, K, N;
};

__global__ void __launch_bounds__(HD_THREADS)
head_layer_bwd_kernel(HeadLayerBwd L) {
  extern __shared__ float sm[];                      // dz_next (R, n2+1) | owned columns of w_next (n2, 4)
  __shared__ float dzs[HD_MAXR * HD_CW];
  const int tid = threadIdx.x, n0 = blockIdx.x * HD_CW;
  const int wave = tid >> 6, lane = tid & 63, n = n0 + wave;
  const int ld2 = L.n2 + 1;
  float *dn = sm, *wn = sm + L.R * ld2;
  for (int e = tid; e < L.R * L.n2; e += HD_THREADS) { const int r = e / L.n2, j = e - r * L.n2; dn[r * ld2 + j] = L.dz_next[e]; }
  for (int e = tid; e < L.n2 * HD_CW; e += HD_THREADS) {
    const int j = e / HD_CW, c = e - j * HD_CW;
    wn[e] = (n0 + c < L.N) ? L.w_next[(long long)j * L.N + n0 + c] : 0.f;
  }
  __syncthreads();
  const bool rok = lane < L.R && n < L.N;
  // dh for (row = lane, column = n0 + wave): reduction over the next layer's columns
  float dh = 0.f;
  if (rok) {
    const float *dzr = dn + lane * ld2;
#pragma unroll 8
    for (int j = 0; j < L.n2; ++j) dh = fmaf(dzr[j], wn[j * HD_CW + wave], dh);
  }
  float yv = 0.f, xhat = 0.f, g = 0.f;
  if (rok) {
    yv = L.y[(long long)lane * L.N + n];
    xhat = (yv - L.mean[n]) * L.invstd[n];
    const float z = xhat * L.gamma[n] + L.beta[n];
    g = dh;
    if (L.drop_p > 0.f)
      g = hd_keep(L.seed, (unsigned)(*L.step - L.step_back), (unsigned)L.layer, (unsigned)(lane * L.N + n), L.drop_p)
              ? g / (1.f - L.drop_p) : 0.f;
    g = z > 0.f ? g : 0.f;                       // gradient w.r.t. the BatchNorm output
  }
  const float sum_g = rs_wave_sum_f32(g), sum_gx = rs_wave_sum_f32(g * xhat);
  if (n < L.N) {
    if (lane == 0) { L.dgamma[n] = sum_gx; L.dbeta[n] = sum_g; }
    const float inv_r = 1.f / (float)L.R;
    const float dzv = rok ? L.gamma[n] * L.invstd[n] * (g - sum_g * inv_r - xhat * sum_gx * inv_r) : 0.f;
    dzs[lane * HD_CW + wave] = dzv;
    if (rok) L.dz[(long long)lane * L.N + n] = dzv;
  } else {
    dzs[lane * HD_CW + wave] = 0.f;
  }
  __syncthreads();
  // weight gradient rows n0 .. n0+3: thread per k (coalesced reads of x rows and writes of dW rows)
  for (int k = tid; k < L.K; k += HD_THREADS) {
    float acc[HD_CW];
#pragma unroll
    for (int c = 0; c < HD_CW; ++c) acc[c] = 0.f;
#pragma unroll 8
    for (int r = 0; r < L.R; ++r) {
      const float xv = L.x[(long long)r * L.ldx + k];
#pragma unroll
      for (int c = 0; c < HD_CW; ++c) acc[c] = fmaf(dzs[r * HD_CW + c], xv, acc[c]);
    }
#pragma unroll
    for (int c = 0; c < HD_CW; ++c)
      if (n0 + c < L.N) L.dw[(long long)(n0 + c) * L.K + k] = acc[c];
  }
}

// dx[r][k] = sum_n dz[r][n] * w[n][k]: gradient that leaves the head (into the last abstraction stage).
// Workgroup = 32 consecutive k for all rows; W and dz go through LDS in tiles of 64 n (coalesced float loads);
// thread (kl = tid & 31, rg = tid >> 5) accumulates the rows rg, rg + 8, ...
__global__ void __launch_bounds__(HD_THREADS)
head_dx_kernel(int R, int N, int K, const float *__restrict__ dz, const float *__restrict__ w, float *__restrict__ dx) {
  __shared__ float wsm[64 * 33];
  __shared__ float dzs[HD_MAXR * 65];
  const int tid = threadIdx.x, kl = tid & 31, rg = tid >> 5;
  const int kb = blockIdx.x * 32;
  float acc[HD_MAXR / 8];
#pragma unroll
  for (int i = 0; i < HD_MAXR / 8; ++i) acc[i] = 0.f;
  for (int nb = 0; nb < N; nb += 64) {
    __syncthreads();
    for (int e = tid; e < 64 * 32; e += HD_THREADS) {
      const int nn = e >> 5, kk = e & 31;
      wsm[nn * 33 + kk] = (nb + nn < N && kb + kk < K) ? w[(long long)(nb + nn) * K + kb + kk] : 0.f;
    }
    for (int e = tid; e < R * 64; e += HD_THREADS) {
      const int r = e >> 6, nn = e & 63;
      dzs[r * 65 + nn] = (nb + nn < N) ? dz[(long long)r * N + nb + nn] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int nn = 0; nn < 64; ++nn) {
      const float wv = wsm[nn * 33 + kl];
#pragma unroll
      for (int i = 0; i < HD_MAXR / 8; ++i) {
        const int r = rg + 8 * i;
        acc[i] = fmaf(r < R ? dzs[r * 65 + nn] : 0.f, wv, acc[i]);
      }
    }
  }
  if (kb + kl < K)
#pragma unroll
    for (int i = 0; i < HD_MAXR / 8; ++i) {
      const int r = rg + 8 * i;
      if (r < R) dx[(long long)r * K + kb + kl] = acc[i];
    }
}

// SmoothClsLoss forward + the gradient it sends back: loss = -mean_r sum_j soft[r][j] * logp[r][j],
// soft = 1 - eps on the label, eps / (classes - 1) elsewhere; dlogp = -soft / R (times the incoming scalar gradient later)
__global__ void __launch_bounds__(256)
smooth_loss_kernel(int R, int classes, float eps, const float *__restrict__ logp, const long long *__restrict__ target,
                   float *__restrict__ loss, float *__restrict__ dlogp) {
  __shared__ float part[256];
  const int tid = threadIdx.x;
  float acc = 0.f;
  for (int e = tid; e < R * classes; e += 256) {
    const int r = e / classes, j = e - r * classes;
    const float soft = (long long)j == target[r] ? 1.f - eps : eps / (float)(classes - 1);
    acc = fmaf(soft, logp[e], acc);
    dlogp[e] = -soft / (float)R;
  }
  part[tid] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) part[tid] += part[tid + off];
    __syncthreads();
  }
  if (tid == 0) loss[0] = -part[0] / (float)R;
}

}  // namespace

extern "C" int rs_head_layer_forward(const rs_head_layer *l, void *stream) {
  RS_REQUIRE(l && l->x && l->w && l->b && l->gamma && l->beta && l->y && l->h && l->mean && l->invstd && l->step,
             "rs_head_layer_forward: null pointer");
  RS_REQUIRE(l->R > 0 && l->R <= HD_MAXR && l->K > 0 && l->N > 0, "rs_head_layer_forward: rows=%d (1..%d), K=%d, N=%d", l->R, HD_MAXR, l->K, l->N);
  HeadLayer L;
  L.x = l->x; L.ldx = l->ldx; L.w = l->w; L.b = l->b; L.gamma = l->gamma; L.beta = l->beta;
  L.running_mean = l->running_mean; L.running_var = l->running_var;
  L.momentum = l->momentum; L.eps = l->eps; L.drop_p = l->drop_p;
  L.y = l->y; L.h = l->h; L.mean = l->mean; L.invstd = l->invstd;
  L.seed = l->seed; L.step = l->step; L.layer = l->layer; L.R = l->R; L.K = l->K; L.N = l->N;
  hipLaunchKernelGGL(head_layer_fwd_kernel, dim3(rs_cdiv(l->N, HD_CW)), dim3(HD_THREADS), 0, (hipStream_t)stream, L);
  RS_CHECK_LAUNCH("rs_head_layer_forward");
  return RS_OK;
}

extern "C" int rs_head_output_forward(int rows, int k, int classes, const float *h, const float *w, const float *b,
                                      float *logp, int *step, void *stream) {
  RS_REQUIRE(rows > 0 && rows <= HD_MAXR && k > 0 && classes > 0 && classes <= 256, "rs_head_output_forward: bad size");
  RS_REQUIRE(h && w && b && logp, "rs_head_output_forward: null pointer");
  HeadOut O{h, w, b, logp, step, rows, k, classes};
  const size_t lds = sizeof(float) * ((size_t)(rows + classes) * (k + 1) + (size_t)rows * classes);
  RS_REQUIRE(lds <= 150 * 1024, "rs_head_output_forward: rows=%d, classes=%d, k=%d do not fit the LDS staging", rows, classes, k);
  hipLaunchKernelGGL(head_out_fwd_kernel, dim3(1), dim3(1024), lds, (hipStream_t)stream, O);
  RS_CHECK_LAUNCH("rs_head_output_forward");
  return RS_OK;
}

extern "C" int rs_head_output_backward(int rows, int k, int classes, const float *dlogp, const float *logp, const float *h,
                                       float *dlogits, float *dw, float *db, void *stream) {
  RS_REQUIRE(rows > 0 && rows <= HD_MAXR && k > 0 && classes > 0 && classes <= 256, "rs_head_output_backward: bad size");
  RS_REQUIRE(dlogp && logp && h && dlogits && dw && db, "rs_head_output_backward: null pointer");
  HeadOutBwd O{dlogp, logp, h, dlogits, dw, db, rows, k, classes};
  hipLaunchKernelGGL(head_out_bwd_kernel, dim3(1), dim3(1024), sizeof(float) * (size_t)rows * classes, (hipStream_t)stream, O);
  RS_CHECK_LAUNCH("rs_head_output_backward");
  return RS_OK;
}

extern "C" int rs_head_layer_backward(const rs_head_layer_bwd *l, void *stream) {
  RS_REQUIRE(l && l->dz_next && l->w_next && l->y && l->mean && l->invstd && l->gamma && l->beta && l->x && l->dz && l->dw &&
             l->dgamma && l->dbeta && l->step, "rs_head_layer_backward: null pointer");
  RS_REQUIRE(l->R > 0 && l->R <= HD_MAXR && l->K > 0 && l->N > 0 && l->n2 > 0, "rs_head_layer_backward: bad size");
  HeadLayerBwd L;
  L.dz_next = l->dz_next; L.n2 = l->n2; L.w_next = l->w_next; L.y = l->y; L.mean = l->mean; L.invstd = l->invstd;
  L.gamma = l->gamma; L.beta = l->beta; L.x = l->x; L.ldx = l->ldx; L.dz = l->dz; L.dw = l->dw; L.dgamma = l->dgamma;
  L.dbeta = l->dbeta; L.drop_p = l->drop_p; L.seed = l->seed; L.step = l->step; L.layer = l->layer; L.step_back = l->step_back;
  L.R = l->R; L.K = l->K; L.N = l->N;
  const size_t lds = sizeof(float) * ((size_t)l->R * (l->n2 + 1) + (size_t)l->n2 * HD_CW);
  RS_REQUIRE(lds <= 120 * 1024, "rs_head_layer_backward: rows=%d x n2=%d do not fit the LDS staging", l->R, l->n2);
  hipLaunchKernelGGL(head_layer_bwd_kernel, dim3(rs_cdiv(l->N, HD_CW)), dim3(HD_THREADS), lds, (hipStream_t)stream, L);
  RS_CHECK_LAUNCH("rs_head_layer_backward");
  return RS_OK;
}

extern "C" int rs_head_input_backward(int rows, int n, int k, const float *dz, const float *w, float *dx, void *stream) {
  RS_REQUIRE(rows > 0 && rows <= HD_MAXR && n > 0 && k > 0, "rs_head_input_backward: bad size");
  RS_REQUIRE(dz && w && dx, "rs_head_input_backward: null pointer");
  hipLaunchKernelGGL(head_dx_kernel, dim3(rs_cdiv(k, 32)), dim3(HD_THREADS), 0, (hipStream_t)stream, rows, n, k, dz, w, dx);
  RS_CHECK_LAUNCH("rs_head_input_backward");
  return RS_OK;
}

extern "C" int rs_smooth_cls_loss(int rows, int classes, float eps, const float *logp, const long long *target,
                                  float *loss, float *dlogp, void *stream) {
  RS_REQUIRE(rows > 0 && classes > 1, "rs_smooth_cls_loss: bad size");
  RS_REQUIRE(logp && target && loss && dlogp, "rs_smooth_cls_loss: null pointer");
  hipLaunchKernelGGL(smooth_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rows, classes, eps, logp, target, loss, dlogp);
  RS_CHECK_LAUNCH("rs_smooth_cls_loss");
  return RS_OK;
}
