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

constexpr int HD_THREADS = 512;   // 8 waves: each takes 1/8 of the reduction dimension of the workgroup's 32-column block
constexpr int HD_WAVES = HD_THREADS / 64;
constexpr int HD_COLS = 32;       // columns owned by a workgroup (one MFMA tile wide)
constexpr int HD_MAXR = 64;       // rows (batch) supported: two 32-row MFMA blocks
typedef float hd_f16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned hd_hash(unsigned seed, unsigned step, unsigned layer, unsigned e) {
  unsigned x = e * 0x9E3779B1u ^ (seed + step * 0x7F4A7C15u + layer * 0x94D049BBu);
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
// keep-mask of Dropout(p): keep when the 24-bit uniform is >= p
__device__ __forceinline__ bool hd_keep(unsigned seed, unsigned step, unsigned layer, unsigned e, float p) {
  return (float)(hd_hash(seed, step, layer, e) >> 8) * (1.0f / 16777216.0f) >= p;
}

// 4 consecutive floats of a row, zero beyond `avail`; one 16-B load when the row is aligned (vec)
__device__ __forceinline__ float4 hd_load4(const float *__restrict__ p, int avail, bool vec) {
  if (avail <= 0) return make_float4(0.f, 0.f, 0.f, 0.f);
  if (vec && avail >= 4) return *reinterpret_cast<const float4 *>(p);
  float4 v;
  v.x = p[0]; v.y = avail > 1 ? p[1] : 0.f; v.z = avail > 2 ? p[2] : 0.f; v.w = avail > 3 ? p[3] : 0.f;
  return v;
}
__device__ __forceinline__ float hd_get(const float4 &v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }
// row of accumulator register i in the 32x32 MFMA tile, for lane half h = lane >> 5 (column = lane & 31)
__device__ __forceinline__ int hd_drow(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// acc[rb] (32 rows x 32 columns) += A[rb*32 + i][k] * B[c0 + j][k] over k in [kbeg, kend): both operands k-contiguous
// (x . W^T).  v_mfma_f32_32x32x2_f32 takes, per step, lane (i = lane & 31, h = lane >> 5) -> A[i][k_h], B[i][k_h]; the order
// of the k's is free as long as both operands agree, so a lane feeds 4 consecutive k (one 16-byte fragment per operand) to 4 steps.
// The operands reach the lanes through a wave-private LDS buffer.  (Round 2 loaded the fragments straight from global memory:
// a load instruction then has every lane on a different row -- 64 cache lines per instruction, 16 bytes used of each, and the next
// three instructions come back for the same lines; with 8 waves x 2 operands x 64 lines in flight (128 KB against 32 KB of L1) they
// were fetched from L2 again each time: 8 x the bytes, 5 us per 32-k trip, 25 us for the 32 x 1024 -> 512 layer whose weights
// are 2 MB; 15.7 us now.)  Here 8 (4)
// adjacent lanes read one whole 128-byte (64-byte) row segment, the chunk is written to LDS as planes of [k / 4][row] float4
// (row XOR plane: the minimum of 4 bank passes per write), and the fragment read of lane (c, h) for the k group 2 g + h is one
// conflict-free ds_read_b128.  `stage`: (1 + RB) * CK * 32 floats.
template <int RB>
__device__ __forceinline__ void hd_mma_nt_lds(const float *__restrict__ A, int lda, int R, const float *__restrict__ B, int ldb,
                                              int ncols, int c0, int K, int kbeg, int kend, hd_f16 (&acc)[RB], float *stage) {
  constexpr int CK = RB == 1 ? 32 : 16;              // k per chunk (the stage aliases the 8 KB partial-tile slot of the wave)
  constexpr int KG = CK / 4, RPI = 64 / KG, NB = 32 / RPI;     // 16-byte groups per row, rows per load instruction, instructions per 32 rows
  const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  const int seg = lane % KG, lr = lane / KG;
  const bool vec = (((lda | ldb | K) & 3) == 0) && ((((size_t)A | (size_t)B) & 15) == 0);
  float4 *Ws = reinterpret_cast<float4 *>(stage), *As = Ws + KG * 32;
  const int kmax = min(kend, K);
  // PF chunks are requested together, then committed and multiplied one after the other: a wave's whole k range of the
  // classifier's layers (1 024 / 8 waves = 4 chunks) is ONE round trip to the weights instead of one per chunk -- the MFMAs of a
  // chunk (0.4 us) are far too short to cover the next chunk's HBM latency.
  constexpr int PF = RB == 1 ? 4 : 2;
  float4 wv[PF][NB], av[PF][RB][NB];
  // whole chunks of aligned rows (the classifier: K = 1 024 / 512): plain 16-byte loads, rows / columns beyond the end clamped
  // to valid ones (their products land in accumulator rows / columns nobody reads).  Left to hd_load4's partial-row handling
  // the compiler scalarised EVERY load of this loop into predicated 4-byte loads (no dwordx4 in the ISA at all).
  const bool fast = vec && kend <= K && ((kend - kbeg) % CK) == 0;
  auto fetch = [&](int k0, float4 (&w_)[NB], float4 (&a_)[RB][NB]) {
    if (fast) {
#pragma unroll
      for (int p = 0; p < NB; ++p) {
        const int row = lr + RPI * p, col = min(c0 + row, ncols - 1), kq = k0 + 4 * seg;
        w_[p] = *reinterpret_cast<const float4 *>(B + (long long)col * ldb + kq);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
          a_[rb][p] = *reinterpret_cast<const float4 *>(A + (long long)min(rb * 32 + row, R - 1) * lda + kq);
      }
      return;
    }
#pragma unroll
    for (int p = 0; p < NB; ++p) {
      const int row = lr + RPI * p, col = c0 + row, kq = k0 + 4 * seg;
      w_[p] = hd_load4(B + (long long)(col < ncols ? col : 0) * ldb + kq, col < ncols ? kmax - kq : 0, vec);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const int r = rb * 32 + row;
        a_[rb][p] = hd_load4(A + (long long)(r < R ? r : 0) * lda + kq, r < R ? kmax - kq : 0, vec);
      }
    }
  };
  for (int kg = kbeg; kg < kend; kg += PF * CK) {
#pragma unroll
    for (int f = 0; f < PF; ++f)
      if (kg + f * CK < kend) fetch(kg + f * CK, wv[f], av[f]);
#pragma unroll
    for (int f = 0; f < PF; ++f) {
      if (kg + f * CK >= kend) break;
      __builtin_amdgcn_wave_barrier();               // (compiler-only: LDS operations of a wave complete in order)
#pragma unroll
      for (int p = 0; p < NB; ++p) {
        const int row = lr + RPI * p;
        Ws[seg * 32 + (row ^ seg)] = wv[f][p];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) As[(rb * KG + seg) * 32 + (row ^ seg)] = av[f][rb][p];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int g = 0; g < KG / 2; ++g) {
        const int kq = 2 * g + h;
        const float4 b4 = Ws[kq * 32 + (c ^ kq)];
        float4 a4[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) a4[rb] = As[(rb * KG + kq) * 32 + (c ^ kq)];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int rb = 0; rb < RB; ++rb)
            acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(hd_get(a4[rb], i), hd_get(b4, i), acc[rb], 0, 0, 0);
      }
    }
  }
}

// acc[rb] += A[rb*32 + i][n] * W[n][c0 + j] over n in [nbeg, nend): A n-contiguous, W column-contiguous (dz . W)
template <int RB>
__device__ __forceinline__ void hd_mma_nn(const float *__restrict__ A, int lda, int R, const float *__restrict__ W, int ldw,
                                          int ncols, int c0, int N, int nbeg, int nend, hd_f16 (&acc)[RB]) {
  const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  const bool vec = (((lda | N) & 3) == 0) && (((size_t)A & 15) == 0);
  const bool cok = c0 + c < ncols;
  const float *wp = W + (cok ? c0 + c : 0);
  constexpr int U = 4;
  // whole trips of aligned rows: plain loads, rows / columns beyond the end clamped (their products are never read); see hd_mma_nt_lds
  const bool fast = vec && nend <= N && ((nend - nbeg) % (8 * U)) == 0;
  const int cc = min(c0 + c, ncols - 1);
  for (int n0 = nbeg + 4 * h; n0 < nend; n0 += 8 * U) {
    float b[U][4];
    float4 a4[U][RB];
    if (fast) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int nq = n0 + 8 * u;
#pragma unroll
        for (int i = 0; i < 4; ++i) b[u][i] = W[(long long)(nq + i) * ldw + cc];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) a4[u][rb] = *reinterpret_cast<const float4 *>(A + (long long)min(rb * 32 + c, R - 1) * lda + nq);
      }
    } else {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int nq = n0 + 8 * u, avail = min(nend, N) - nq;
#pragma unroll
      for (int i = 0; i < 4; ++i) b[u][i] = (cok && i < avail) ? wp[(long long)(nq + i) * ldw] : 0.f;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const int r = rb * 32 + c;
        a4[u][rb] = hd_load4(A + (long long)(r < R ? r : 0) * lda + nq, r < R ? avail : 0, vec);
      }
    }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
          acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(hd_get(a4[u][rb], i), b[u][i], acc[rb], 0, 0, 0);
  }
}

// the 8 waves' partial tiles -> part[wave][row][col] in LDS
template <int RB>
__device__ __forceinline__ void hd_store_partials(const hd_f16 (&acc)[RB], float *part) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) part[(wave * HD_MAXR + rb * 32 + hd_drow(i, h)) * HD_COLS + c] = acc[rb][i];
}
// element (r, c) of the workgroup's block: sum of the waves' partials
__device__ __forceinline__ float hd_block_at(const float *part, int r, int c) {
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < HD_WAVES; ++w) t += part[(w * HD_MAXR + r) * HD_COLS + c];
  return t;
}
// sum over the rows of a column: thread (c = tid & 31, rg = tid >> 5) contributes `v`; all threads get the column total
__device__ __forceinline__ float hd_col_sum(float v, float *colred) {
  const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
  __syncthreads();
  colred[rg * HD_COLS + c] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int g = 0; g < HD_THREADS / 32; ++g) t += colred[g * HD_COLS + c];
  return t;
}
// split [0, n) into HD_WAVES ranges that are multiples of 8
__device__ __forceinline__ void hd_wave_range(int n, int &beg, int &end) {
  const int per = ((n + 8 * HD_WAVES - 1) / (8 * HD_WAVES)) * 8, wave = threadIdx.x >> 6;
  beg = min(n, wave * per); end = min(n, beg + per);
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

// Linear + BatchNorm1d(train) + ReLU + Dropout for 32 owned columns and all rows: the 8 waves split K, their partial
// MFMA tiles meet in LDS, then thread (c, rg) finishes the rows rg, rg + 16, ... of column c.
__global__ void __launch_bounds__(HD_THREADS)
head_layer_fwd_kernel(HeadLayer L) {
  __shared__ float part[HD_WAVES * HD_MAXR * HD_COLS];
  __shared__ float colred[(HD_THREADS / 32) * HD_COLS];
  const int tid = threadIdx.x, c = tid & 31, rg = tid >> 5, n0 = blockIdx.x * HD_COLS, n = n0 + c;
  // the waves' k ranges are multiples of 32 (whole staging chunks); a wave's staging buffer is its own slot of `part`
  const int kper = ((L.K + 32 * HD_WAVES - 1) / (32 * HD_WAVES)) * 32, wv_ = tid >> 6;
  const int kbeg = min(L.K, wv_ * kper), kend = min(L.K, kbeg + kper);
  float *stage = part + wv_ * HD_MAXR * HD_COLS;
  if (L.R <= 32) {
    hd_f16 acc[1] = {};
    hd_mma_nt_lds<1>(L.x, L.ldx, L.R, L.w, L.K, L.N, n0, L.K, kbeg, kend, acc, stage);
    __builtin_amdgcn_wave_barrier();
    hd_store_partials<1>(acc, part);
  } else {
    hd_f16 acc[2] = {};
    hd_mma_nt_lds<2>(L.x, L.ldx, L.R, L.w, L.K, L.N, n0, L.K, kbeg, kend, acc, stage);
    __builtin_amdgcn_wave_barrier();
    hd_store_partials<2>(acc, part);
  }
  __syncthreads();
  const bool nok = n < L.N;
  const float bias = nok ? L.b[n] : 0.f;
  float v[4], s = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = rg + 16 * q;
    v[q] = (r < L.R && nok) ? hd_block_at(part, r, c) + bias : 0.f;
    s += v[q];
  }
  const float mean = hd_col_sum(s, colred) / (float)L.R;
  float d[4], s2 = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    d[q] = (rg + 16 * q < L.R && nok) ? v[q] - mean : 0.f;
    s2 = fmaf(d[q], d[q], s2);
  }
  const float var = hd_col_sum(s2, colred) / (float)L.R;
  const float invstd = 1.0f / sqrtf(var + L.eps);
  if (!nok) return;
  if (rg == 0) {
    L.mean[n] = mean; L.invstd[n] = invstd;
    if (L.running_mean) {       // nn.BatchNorm1d: running = (1-m) running + m batch, variance unbiased
      L.running_mean[n] = (1.f - L.momentum) * L.running_mean[n] + L.momentum * mean;
      const float unb = L.R > 1 ? var * (float)L.R / (float)(L.R - 1) : var;
      L.running_var[n] = (1.f - L.momentum) * L.running_var[n] + L.momentum * unb;
    }
  }
  const float ga = L.gamma[n], be = L.beta[n];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = rg + 16 * q;
    if (r >= L.R) continue;
    const float z = d[q] * invstd * ga + be;
    float a = fmaxf(z, 0.f);
    if (L.drop_p > 0.f)
      a = hd_keep(L.seed, (unsigned)*L.step, (unsigned)L.layer, (unsigned)(r * L.N + n), L.drop_p) ? a / (1.f - L.drop_p) : 0.f;
    L.y[(long long)r * L.N + n] = v[q];
    L.h[(long long)r * L.N + n] = a;
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
  // both matrices are contiguous: one pass over (R + classes) * K elements, 4 loads in flight per thread (one per trip was a
  // chain of ~12 round trips for 12 K floats -- most of this launch's 12 us)
  {
    const int nh = O.R * O.K, nall = nh + O.classes * O.K, st = (int)blockDim.x;
    for (int e0 = tid; e0 < nall; e0 += 4 * st) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = min(e0 + u * st, nall - 1);
        v[u] = e < nh ? O.h[e] : O.w[e - nh];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * st;
        if (e >= nall) break;
        const int row = e / O.K, k = e - row * O.K;                 // rows of h first, then rows of w: hs and wsm are adjacent, same pitch
        hs[row * ldk + k] = v[u];
      }
    }
  }
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
  extern __shared__ float dl[];                      // dlogits | dlogp | logp, (R, classes) each
  const int tid = threadIdx.x;
  // dlogp / logp rows meet in LDS first (one element per thread and trip: the per-row loops below read LDS, not a chain of
  // 2 * classes global round trips per row)
  float *gp = dl + O.R * O.classes, *lp = gp + O.R * O.classes;
  for (int e = tid; e < O.R * O.classes; e += blockDim.x) { gp[e] = O.dlogp[e]; lp[e] = O.logp[e]; }
  __syncthreads();
  for (int r = tid; r < O.R; r += blockDim.x) {
    float s = 0.f;
    for (int j = 0; j < O.classes; ++j) s += gp[r * O.classes + j];
    for (int j = 0; j < O.classes; ++j) {
      const float g = gp[r * O.classes + j] - expf(lp[r * O.classes + j]) * s;
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
  __shared__ float part[HD_WAVES * HD_MAXR * HD_COLS];
  __shared__ float colred[(HD_THREADS / 32) * HD_COLS];
  __shared__ float dzs[HD_MAXR * (HD_COLS + 1)];
  const int tid = threadIdx.x, c = tid & 31, rg = tid >> 5, n0 = blockIdx.x * HD_COLS, n = n0 + c;
  // dh block (rows x 32 owned columns) = dz_next . w_next[:, n0 : n0 + 32]; the waves split the next layer's columns
  int jbeg, jend;
  hd_wave_range(L.n2, jbeg, jend);
  if (L.R <= 32) {
    hd_f16 acc[1] = {};
    hd_mma_nn<1>(L.dz_next, L.n2, L.R, L.w_next, L.N, L.N, n0, L.n2, jbeg, jend, acc);
    hd_store_partials<1>(acc, part);
  } else {
    hd_f16 acc[2] = {};
    hd_mma_nn<2>(L.dz_next, L.n2, L.R, L.w_next, L.N, L.N, n0, L.n2, jbeg, jend, acc);
    hd_store_partials<2>(acc, part);
  }
  __syncthreads();
  const bool nok = n < L.N;
  const float mean = nok ? L.mean[n] : 0.f, invstd = nok ? L.invstd[n] : 0.f;
  const float ga = nok ? L.gamma[n] : 0.f, be = nok ? L.beta[n] : 0.f;
  unsigned step = 0;
  if (L.drop_p > 0.f) step = (unsigned)(*L.step - L.step_back);
  float g[4], xh[4], sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = rg + 16 * q;
    g[q] = 0.f; xh[q] = 0.f;
    if (r < L.R && nok) {
      xh[q] = (L.y[(long long)r * L.N + n] - mean) * invstd;
      const float z = xh[q] * ga + be;
      float t = hd_block_at(part, r, c);
      if (L.drop_p > 0.f) t = hd_keep(L.seed, step, (unsigned)L.layer, (unsigned)(r * L.N + n), L.drop_p) ? t / (1.f - L.drop_p) : 0.f;
      g[q] = z > 0.f ? t : 0.f;                  // gradient w.r.t. the BatchNorm output
    }
    sg += g[q];
    sgx = fmaf(g[q], xh[q], sgx);
  }
  const float sum_g = hd_col_sum(sg, colred), sum_gx = hd_col_sum(sgx, colred);
  if (nok && rg == 0) { L.dgamma[n] = sum_gx; L.dbeta[n] = sum_g; }
  const float inv_r = 1.f / (float)L.R;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = rg + 16 * q;
    const float dzv = (r < L.R && nok) ? ga * invstd * (g[q] - sum_g * inv_r - xh[q] * sum_gx * inv_r) : 0.f;
    dzs[r * (HD_COLS + 1) + c] = dzv;
    if (r < L.R && nok) L.dz[(long long)r * L.N + n] = dzv;
  }
  __syncthreads();
  // weight gradient rows n0 .. n0+31: tile (32 n x 32 k) = dzs^T . x over the rows; a wave takes every 8th k tile
  const int wave = tid >> 6, lane = tid & 63, h = lane >> 5, lc = lane & 31;
  const int steps = (L.R + 1) >> 1;
  for (int k0 = wave * 32; k0 < L.K; k0 += HD_WAVES * 32) {
    hd_f16 acc = {};
    const bool kok = k0 + lc < L.K;
    for (int s0 = 0; s0 < steps; s0 += 8) {
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = 2 * (s0 + u) + h;
        a[u] = r < HD_MAXR ? dzs[r * (HD_COLS + 1) + lc] : 0.f;     // rows R .. 63 hold 0
        b[u] = (kok && r < L.R) ? L.x[(long long)r * L.ldx + k0 + lc] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int nn = n0 + hd_drow(i, h);
      if (kok && nn < L.N) L.dw[(long long)nn * L.K + k0 + lc] = acc[i];
    }
  }
}

// dx[r][k] = sum_n dz[r][n] * w[n][k]: gradient that leaves the head (into the last abstraction stage).
// Workgroup = 32 consecutive k for all rows; the 8 waves split n.
__global__ void __launch_bounds__(HD_THREADS)
head_dx_kernel(int R, int N, int K, const float *__restrict__ dz, const float *__restrict__ w, float *__restrict__ dx) {
  __shared__ float part[HD_WAVES * HD_MAXR * HD_COLS];
  const int tid = threadIdx.x, c = tid & 31, rg = tid >> 5, kb = blockIdx.x * HD_COLS;
  int nbeg, nend;
  hd_wave_range(N, nbeg, nend);
  if (R <= 32) {
    hd_f16 acc[1] = {};
    hd_mma_nn<1>(dz, N, R, w, K, K, kb, N, nbeg, nend, acc);
    hd_store_partials<1>(acc, part);
  } else {
    hd_f16 acc[2] = {};
    hd_mma_nn<2>(dz, N, R, w, K, K, kb, N, nbeg, nend, acc);
    hd_store_partials<2>(acc, part);
  }
  __syncthreads();
  if (kb + c < K)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = rg + 16 * q;
      if (r < R) dx[(long long)r * K + kb + c] = hd_block_at(part, r, c);
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

// ---- nn.CrossEntropyLoss(ignore_index) on logits (rows, classes): segmentation/tool/train.py:110,296 -------------------------
// loss = mean over the rows whose target != ignore_index of (logsumexp(row) - row[target]).  One thread per row: the row's
// softmax minus the one-hot target goes out as the UNSCALED gradient (zero row when ignored; the 1 / count factor is known
// only after the reduction and is applied by ce_scale_kernel in backward), the row losses and the valid-row count meet in a
// fixed-order tree per workgroup -> partial[block] = {sum, count}.
__global__ void __launch_bounds__(256)
ce_rows_kernel(long long rows, int classes, long long ignore_index, const float *__restrict__ logits,
               const long long *__restrict__ target, float *__restrict__ dlogits, double *__restrict__ partial, int *__restrict__ bad_labels) {
  __shared__ double ps[256], pc[256];
  const int tid = threadIdx.x;
  const long long r = (long long)blockIdx.x * 256 + tid;
  double loss = 0.0, cnt = 0.0;
  if (r < rows) {
    const float *x = logits + r * classes;
    float *d = dlogits + r * classes;
    const long long t = target[r];
    if (t == ignore_index) {
      for (int j = 0; j < classes; ++j) d[j] = 0.f;
    } else if (t < 0 || t >= classes) {
      // a label outside [0, classes) that is NOT the ignore label is a data bug (torch traps it with a device assert): it must
      // not train as a silently masked row -- the row's loss and gradient become NaN, which the mean carries to the caller
      // (a NaN inside a replayed graph reaches Adam's moments with no host-visible error: the persistent counter is what a
      //  training loop polls -- repsurf_amd.head.bad_label_count -- when it reads its loss)
      const float qnan = __builtin_nanf("");
      if (bad_labels) atomicAdd(bad_labels, 1);
      for (int j = 0; j < classes; ++j) d[j] = qnan;
      loss = (double)qnan;
      cnt = 1.0;
    } else {
      float mx = x[0];
      for (int j = 1; j < classes; ++j) mx = fmaxf(mx, x[j]);
      float se = 0.f;
      for (int j = 0; j < classes; ++j) se += expf(x[j] - mx);
      const float lse = mx + logf(se);
      for (int j = 0; j < classes; ++j) d[j] = expf(x[j] - lse) - (j == (int)t ? 1.f : 0.f);
      loss = (double)(lse - x[t]);
      cnt = 1.0;
    }
  }
  ps[tid] = loss; pc[tid] = cnt;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) { ps[tid] += ps[tid + off]; pc[tid] += pc[tid + off]; }
    __syncthreads();
  }
  if (tid == 0) { partial[2 * (long long)blockIdx.x] = ps[0]; partial[2 * (long long)blockIdx.x + 1] = pc[0]; }
}

// loss[0] = sum / count (NaN for an all-ignored batch, like torch), inv_count[0] = 1 / count (0 when count = 0)
__global__ void __launch_bounds__(256)
ce_finalize_kernel(int nblk, const double *__restrict__ partial, float *__restrict__ loss, float *__restrict__ inv_count) {
  __shared__ double ps[256], pc[256];
  const int tid = threadIdx.x;
  double s = 0.0, c = 0.0;
  for (int b = tid; b < nblk; b += 256) { s += partial[2 * b]; c += partial[2 * b + 1]; }
  ps[tid] = s; pc[tid] = c;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) { ps[tid] += ps[tid + off]; pc[tid] += pc[tid + off]; }
    __syncthreads();
  }
  if (tid == 0) { loss[0] = (float)(ps[0] / pc[0]); inv_count[0] = pc[0] > 0.0 ? (float)(1.0 / pc[0]) : 0.f; }
}

// out = x * a[0] * (b ? b[0] : 1): the cross-entropy gradient times 1 / count and the incoming scalar gradient
__global__ void __launch_bounds__(256)
ce_scale_kernel(long long n, const float *__restrict__ x, const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out) {
  const float f = a[0] * (b ? b[0] : 1.f);
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) out[e] = x[e] * f;
}

// column sums of x (rows, n) with row pitch ldx, stage 1: workgroup b sums rows [b * per, (b + 1) * per) -> partial[b][:]
// (the bias gradient of a row Linear: dout.sum(0); stage 2 is rs_reduce_partials' fixed-order sum)
__global__ void __launch_bounds__(256)
col_sum_kernel(long long rows_arg, const int *__restrict__ rows_dev, int n, const float *__restrict__ x, long long ldx, long long per, float scale, float *__restrict__ partial) {
  __shared__ float red[256];
  const int tid = threadIdx.x;
  const long long rows = rows_dev ? min(rows_arg, (long long)*rows_dev) : rows_arg;      // (rows beyond a device count are not read)
  const long long r0 = (long long)blockIdx.x * per, r1 = min(rows, r0 + per);
  if (n <= 16) {      // few columns (13 classes): a thread walks whole rows, 256 consecutive rows per trip, sums in registers
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
    for (long long r = r0 + tid; r < r1; r += 256)
#pragma unroll
      for (int c = 0; c < 16; ++c) if (c < n) acc[c] += x[r * ldx + c];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (c >= n) continue;                         // (n is uniform: the barriers below are met by all threads or none)
      red[tid] = acc[c];
      __syncthreads();
      for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) red[tid] += red[tid + off];
        __syncthreads();
      }
      if (tid == 0) partial[(long long)blockIdx.x * n + c] = red[0] * scale;
      __syncthreads();
    }
    return;
  }
  // lanes along the columns when there are many of them, along the rows otherwise
  const int tc = n >= 256 ? 256 : (n >= 64 ? 64 : (n >= 16 ? 16 : 1)), tr = 256 / tc;
  const int cx = tid % tc, rx = tid / tc;
  for (int c0 = 0; c0 < n; c0 += tc) {
    const int c = c0 + cx;
    float s = 0.f;
    if (c < n) for (long long r = r0 + rx; r < r1; r += tr) s += x[r * ldx + c];
    red[tid] = s;
    __syncthreads();
    for (int off = tr / 2; off > 0; off >>= 1) {
      if (rx < off) red[tid] += red[tid + off * tc];
      __syncthreads();
    }
    if (rx == 0 && c < n) partial[(long long)blockIdx.x * n + c] = red[cx] * scale;
    __syncthreads();
  }
}

}  // namespace

extern "C" int rs_cross_entropy_forward(long long rows, int classes, long long ignore_index, const float *logits, const long long *target,
                                        float *loss, float *inv_count, float *dlogits, double *partial, int *bad_labels, void *stream) {
  RS_REQUIRE(rows > 0 && classes > 0 && rows <= (1LL << 31) * 255, "rs_cross_entropy_forward: bad size");
  RS_REQUIRE(logits && target && loss && inv_count && dlogits && partial, "rs_cross_entropy_forward: null pointer");
  const int nblk = (int)((rows + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_rows_kernel, dim3(nblk), dim3(256), 0, st, rows, classes, ignore_index, logits, target, dlogits, partial, bad_labels);
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, st, nblk, partial, loss, inv_count);
  RS_CHECK_LAUNCH("rs_cross_entropy_forward");
  return RS_OK;
}

extern "C" int rs_scale_by_scalars(long long n, const float *x, const float *a, const float *b, float *out, void *stream) {
  RS_REQUIRE(n >= 0, "rs_scale_by_scalars: negative size");
  if (n == 0) return RS_OK;
  RS_REQUIRE(x && a && out, "rs_scale_by_scalars: null pointer");
  long long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(ce_scale_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, n, x, a, b, out);
  RS_CHECK_LAUNCH("rs_scale_by_scalars");
  return RS_OK;
}

extern "C" int rs_col_sum_partials_dev(long long rows, int n, const float *x, long long ldx, float scale, float *partial, int nblk,
                                       const int *rows_dev, void *stream) {
  RS_REQUIRE(rows > 0 && n > 0 && nblk > 0 && ldx >= n, "rs_col_sum_partials: bad size");
  RS_REQUIRE(x && partial, "rs_col_sum_partials: null pointer");
  const long long per = (rows + nblk - 1) / nblk;
  hipLaunchKernelGGL(col_sum_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, rows, rows_dev, n, x, ldx, per, scale, partial);
  RS_CHECK_LAUNCH("rs_col_sum_partials");
  return RS_OK;
}

extern "C" int rs_col_sum_partials(long long rows, int n, const float *x, long long ldx, float scale, float *partial, int nblk, void *stream) {
  return rs_col_sum_partials_dev(rows, n, x, ldx, scale, partial, nblk, nullptr, stream);
}

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
  hipLaunchKernelGGL(head_layer_fwd_kernel, dim3(rs_cdiv(l->N, HD_COLS)), dim3(HD_THREADS), 0, (hipStream_t)stream, L);
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
  const size_t lds = 3 * sizeof(float) * (size_t)rows * classes;      // dlogits, dlogp, logp staged for all rows
  RS_REQUIRE(lds <= 64 * 1024, "rs_head_output_backward: rows=%d x classes=%d needs %zu bytes of LDS staging (limit 65536): the fused head serves "
             "rows * classes <= 5461 (the shipped 15 / 40-class heads at <= 64 rows)", rows, classes, lds);
  hipLaunchKernelGGL(head_out_bwd_kernel, dim3(1), dim3(1024), lds, (hipStream_t)stream, O);
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
  hipLaunchKernelGGL(head_layer_bwd_kernel, dim3(rs_cdiv(l->N, HD_COLS)), dim3(HD_THREADS), 0, (hipStream_t)stream, L);
  RS_CHECK_LAUNCH("rs_head_layer_backward");
  return RS_OK;
}

extern "C" int rs_head_input_backward(int rows, int n, int k, const float *dz, const float *w, float *dx, void *stream) {
  RS_REQUIRE(rows > 0 && rows <= HD_MAXR && n > 0 && k > 0, "rs_head_input_backward: bad size");
  RS_REQUIRE(dz && w && dx, "rs_head_input_backward: null pointer");
  hipLaunchKernelGGL(head_dx_kernel, dim3(rs_cdiv(k, HD_COLS)), dim3(HD_THREADS), 0, (hipStream_t)stream, rows, n, k, dz, w, dx);
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
