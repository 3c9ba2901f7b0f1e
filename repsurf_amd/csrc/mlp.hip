// mlp.hip — grouped shared-MLP stack on fp32 MFMA (v_mfma_f32_32x32x2_f32) for gfx950.
//
// The reference runs every layer of SurfaceAbstractionCD / UmbrellaSurfaceConstructor as three
// framework calls (nn.Conv2d 1x1 -> nn.BatchNorm2d -> F.relu,
// classification/modules/repsurface_utils.py:236-244, 296-305), i.e. per layer ~5 full passes
// over the (B, C, nsample, npoint) activation in forward and ~8 in backward.  Here a layer is
//   forward : ONE row-GEMM  y[rows, cout] = act(x)[rows, cin] . W^T + bias      (rs_mlp_gemm_rows)
//             whose PROLOGUE applies the previous layer's BatchNorm affine + ReLU while staging
//             the operand into LDS, and whose EPILOGUE accumulates the per-channel sum / sum of
//             squares BatchNorm needs (so the normalised activation is never written);
//   backward: ONE data-gradient row-GEMM (same kernel, prologue = BatchNorm backward affine of
//             (dz, y), epilogue = ReLU mask of the producing layer + its BatchNorm-backward
//             sums) and ONE weight-gradient GEMM reducing over rows (rs_mlp_wgrad).
// Only pre-BatchNorm conv outputs y_l and tiny per-channel vectors are kept for backward.
//
// Matrix cores: fp32-input MFMA 32x32x2 (exact fp32 FMA chain, 157 TF peak = 1/16 of bf16).
// Tiling is for 64-wide wavefronts: a 256-thread workgroup owns a 128-row x BN-column output
// tile, each of its 4 waves a 32-row slab and BN/32 accumulator tiles of 16 VGPRs; operands are
// staged K-major in LDS ([k][row + 1 pad]) so that the MFMA fragment read
// (lane -> row = lane & 31, k = lane >> 5) is a conflict-free ds_read_b32.
// Workgroups are persistent over row tiles, so BatchNorm partial sums live in registers (fp64)
// and leave the workgroup once, as one deterministic partial row (no atomics).
#include "rs_common.h"
#include <math.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- on-the-fly row operands -----------------------------------------------------------------
// A logical matrix E[r][c] (r < rows, c < cols) assembled while loading:
//   OPM_ID      E = a[r][c]
//   OPM_RELU1   E = relu(s1[c]*a + t1[c])                         BN + ReLU of a stored conv output
//   OPM_RELU2   E = relu(s1[c]*a + t1[c] + s2[c]*b[r][c] + t2[c]) two-branch first layer (bn_l0 + bn_f0)
//   OPM_AFF2    E = s1[c]*a + s2[c]*b + t1[c]                     BatchNorm backward: a = dz, b = y
//   OPM_POOLED  dz = (arg[g][c] == r % ns) ? a[g][c] : 0, g = r / ns;  E = s1*dz + s2*b + t1
//               (gradient arriving through the max-pool over nsample, never materialised)
//   OPM_BCAST   E = a[r / ns][c]                                  gradient through a sum over ns
enum { OPM_ID = RS_OP_ID, OPM_RELU1 = RS_OP_RELU1, OPM_RELU2 = RS_OP_RELU2, OPM_AFF2 = RS_OP_AFF2,
       OPM_POOLED = RS_OP_POOLED, OPM_BCAST = RS_OP_BCAST };
typedef rs_row_operand RowOperand;

__device__ __forceinline__ float op_fetch(const RowOperand &o, long long r, int c) {
  switch (o.mode) {
    case OPM_ID: return o.a[r * o.lda + c];
    case OPM_RELU1: return fmaxf(fmaf(o.s1[c], o.a[r * o.lda + c], o.t1[c]), 0.f);
    case OPM_RELU2:
      return fmaxf(fmaf(o.s1[c], o.a[r * o.lda + c], o.t1[c]) + fmaf(o.s2[c], o.b[r * o.ldb + c], o.t2[c]), 0.f);
    case OPM_AFF2: return fmaf(o.s1[c], o.a[r * o.lda + c], fmaf(o.s2[c], o.b[r * o.ldb + c], o.t1[c]));
    case OPM_POOLED: {
      const long long g = r / o.ns;
      const int k = (int)(r - g * o.ns);
      const float dz = (o.arg[g * o.lda + c] == k) ? o.a[g * o.lda + c] : 0.f;
      return fmaf(o.s1[c], dz, fmaf(o.s2[c], o.b[r * o.ldb + c], o.t1[c]));
    }
    default: return o.a[(r / o.ns) * o.lda + c];
  }
}

namespace {

constexpr int GM_THREADS = 256;
constexpr int GM_BM = 128;        // rows per workgroup tile (4 waves x 32)
constexpr int GM_BK = 32;         // reduction chunk staged per barrier pair
constexpr int GM_LDA = GM_BM + 1; // K-major LDS rows, +1 pad: transposing stores stay <= 2-way conflicted

enum { EPI_STORE = RS_EPI_STORE, EPI_STATS = RS_EPI_STATS, EPI_MASK = RS_EPI_MASK };
typedef rs_mlp_epilogue Epilogue;

// y[rows, cols] = E[rows, kdim] . B   with B[k][n] = w[n*ldw + k] (TRANSW = false, weights stored [cols][kdim])
//                                       or w[k*ldw + n] (TRANSW = true,  weights stored [kdim][cols])
template <int BN, int EPI, bool TRANSW>
__global__ void __launch_bounds__(GM_THREADS)
gemm_rows_kernel(long long rows, int kdim, int cols, RowOperand E, const float *__restrict__ w, int ldw,
                 Epilogue ep) {
  constexpr int CT = BN / 32;
  __shared__ float At[GM_BK * GM_LDA];
  __shared__ float Wt[GM_BK * (BN + 1)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.y * BN;
  const long long tiles = (rows + GM_BM - 1) / GM_BM;
  const int lrow = lane & 31, lk = lane >> 5;

  double st[3][CT];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int c = 0; c < CT; ++c) st[s][c] = 0.0;

  for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long long r0 = tile * GM_BM;
    f32x16 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;

    for (int k0 = 0; k0 < kdim; k0 += GM_BK) {
      __syncthreads();   // previous chunk's fragment reads are done
      // ---- stage E[r0 .. r0+127][k0 .. k0+31] transposed into At[k][r] (prologue applied here)
      {
        const int kq = (tid & 7) * 4;          // 4 consecutive k per thread
        const int rr = tid >> 3;               // 32 rows per pass
#pragma unroll
        for (int pass = 0; pass < GM_BM / 32; ++pass) {
          const int rl = pass * 32 + rr;
          const long long r = r0 + rl;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = k0 + kq + j;
            const float v = (r < rows && k < kdim) ? op_fetch(E, r, k) : 0.f;
            At[(kq + j) * GM_LDA + rl] = v;
          }
        }
      }
      // ---- stage the weight chunk into Wt[k][n]
      if (TRANSW) {
        for (int e = tid; e < GM_BK * BN; e += GM_THREADS) {
          const int k = e / BN, n = e - k * BN;
          const float v = (k0 + k < kdim && n0 + n < cols) ? w[(long long)(k0 + k) * ldw + n0 + n] : 0.f;
          Wt[k * (BN + 1) + n] = v;
        }
      } else {
        for (int e = tid; e < GM_BK * BN; e += GM_THREADS) {
          const int n = e / GM_BK, k = e - n * GM_BK;
          const float v = (k0 + k < kdim && n0 + n < cols) ? w[(long long)(n0 + n) * ldw + k0 + k] : 0.f;
          Wt[k * (BN + 1) + n] = v;
        }
      }
      __syncthreads();
      const int ksteps = min(GM_BK, kdim - k0 + 1) >> 1;   // pairs of k that hold data
#pragma unroll 4
      for (int ks = 0; ks < ksteps; ++ks) {
        const float a = At[(2 * ks + lk) * GM_LDA + wave * 32 + lrow];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const float b = Wt[(2 * ks + lk) * (BN + 1) + c * 32 + lrow];
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        }
      }
    }

    // ---- epilogue: D[i][j], j = lane & 31, i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const int col = n0 + c * 32 + lrow;
      const bool cok = col < cols;
      const float bias = (ep.bias && cok) ? ep.bias[col] : 0.f;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
      float ms1 = 0.f, mt1 = 0.f, ms2 = 0.f, mt2 = 0.f, mu1 = 0.f, is1 = 0.f, mu2 = 0.f, is2 = 0.f;
      if (EPI == EPI_MASK && cok) {
        ms1 = ep.ms1[col]; mt1 = ep.mt1[col]; mu1 = ep.mean1[col]; is1 = ep.invstd1[col];
        if (ep.my2) { ms2 = ep.ms2[col]; mt2 = ep.mt2[col]; mu2 = ep.mean2[col]; is2 = ep.invstd2[col]; }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const long long r = r0 + wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * lk;
        if (r < rows && cok) {
          float y = acc[c][i] + bias;
          if (EPI == EPI_MASK) {
            const float y1 = ep.my1[r * ep.ldm1 + col];
            float z = fmaf(ms1, y1, mt1);
            float y2 = 0.f;
            if (ep.my2) { y2 = ep.my2[r * ep.ldm2 + col]; z += fmaf(ms2, y2, mt2); }
            y = z > 0.f ? y : 0.f;
            s0 += y;
            s1 = fmaf(y, (y1 - mu1) * is1, s1);
            if (ep.my2) s2 = fmaf(y, (y2 - mu2) * is2, s2);
          } else if (EPI == EPI_STATS) {
            s0 += y;
            s1 = fmaf(y, y, s1);
          }
          ep.out[r * ep.ldo + col] = y;
        }
      }
      if (EPI != EPI_STORE) { st[0][c] += (double)s0; st[1][c] += (double)s1; st[2][c] += (double)s2; }
    }
  }

  if (EPI != EPI_STORE) {
    // workgroup reduction of the fp64 partial sums: 8 contributions (4 waves x 2 lane halves) per column
    __syncthreads();
    double *red = reinterpret_cast<double *>(At);       // 8 x BN doubles <= 8 KB, fits in At (16.5 KB)
    const int nstat = (EPI == EPI_MASK && ep.my2) ? 3 : 2;
    for (int s = 0; s < nstat; ++s) {
#pragma unroll
      for (int c = 0; c < CT; ++c) red[(wave * 2 + lk) * BN + c * 32 + lrow] = st[s][c];
      __syncthreads();
      if (tid < BN) {
        double t = 0.0;
#pragma unroll
        for (int p = 0; p < 8; ++p) t += red[p * BN + tid];
        if (n0 + tid < cols) ep.partial[((long long)blockIdx.x * nstat + s) * cols + n0 + tid] = t;
      }
      __syncthreads();
    }
  }
}

// ---- weight gradient: dw[n][k] = sum_r P[r][n] * Q[r][k] ------------------------------------------
// Workgroup = (row chunk, 128 x (TK*64... ) output block); reduction index = rows -> MFMA k.
constexpr int WG_BR = 32;   // rows staged per barrier pair

template <int WN, int WK, int TN, int TK>   // waves arranged WN x WK, each owning TN x TK 32x32 tiles
__global__ void __launch_bounds__(GM_THREADS)
wgrad_kernel(long long rows, int ncols, int kcols, RowOperand P, RowOperand Q, long long rows_per_chunk,
             float *__restrict__ partial) {
  constexpr int BNN = WN * TN * 32, BKK = WK * TK * 32;
  __shared__ float Ps[WG_BR * BNN];
  __shared__ float Qs[WG_BR * BKK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave / WK, wk = wave % WK;
  const int n0 = blockIdx.y * BNN, k0 = blockIdx.z * BKK;
  const int lcol = lane & 31, lr = lane >> 5;
  const long long rbeg = (long long)blockIdx.x * rows_per_chunk;
  const long long rend = min(rows, rbeg + rows_per_chunk);

  f32x16 acc[TN][TK];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TK; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  for (long long r0 = rbeg; r0 < rend; r0 += WG_BR) {
    __syncthreads();
    for (int e = tid; e < WG_BR * BNN; e += GM_THREADS) {
      const int rl = e / BNN, c = e - rl * BNN;
      const long long r = r0 + rl;
      Ps[e] = (r < rend && n0 + c < ncols) ? op_fetch(P, r, n0 + c) : 0.f;
    }
    for (int e = tid; e < WG_BR * BKK; e += GM_THREADS) {
      const int rl = e / BKK, c = e - rl * BKK;
      const long long r = r0 + rl;
      Qs[e] = (r < rend && k0 + c < kcols) ? op_fetch(Q, r, k0 + c) : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int s = 0; s < WG_BR / 2; ++s) {
      float pa[TN], qb[TK];
#pragma unroll
      for (int a = 0; a < TN; ++a) pa[a] = Ps[(2 * s + lr) * BNN + (wn * TN + a) * 32 + lcol];
#pragma unroll
      for (int b = 0; b < TK; ++b) qb[b] = Qs[(2 * s + lr) * BKK + (wk * TK + b) * 32 + lcol];
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TK; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[a], qb[b], acc[a][b], 0, 0, 0);
    }
  }
  float *dst = partial + (long long)blockIdx.x * ncols * kcols;
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TK; ++b) {
      const int kk = k0 + (wk * TK + b) * 32 + lcol;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int n = n0 + (wn * TN + a) * 32 + (i & 3) + 8 * (i >> 2) + 4 * lr;
        if (n < ncols && kk < kcols) dst[(long long)n * kcols + kk] = acc[a][b][i];
      }
    }
}

// out[e] = sum_c partial[c][e]   (deterministic order)
__global__ void __launch_bounds__(GM_THREADS)
reduce_partials_kernel(int chunks, long long n, const float *__restrict__ partial, float *__restrict__ out) {
  for (long long e = (long long)blockIdx.x * GM_THREADS + threadIdx.x; e < n; e += (long long)gridDim.x * GM_THREADS) {
    float s = 0.f;
    for (int c = 0; c < chunks; ++c) s += partial[(long long)c * n + e];
    out[e] = s;
  }
}

// ---- BatchNorm statistics -> affine (forward) ---------------------------------------------------
__global__ void bn_finalize_kernel(int c, long long rows, int nblk, const double *__restrict__ partial,
                                   const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                   float momentum, float *__restrict__ scale, float *__restrict__ shift,
                                   float *__restrict__ mean_out, float *__restrict__ invstd_out,
                                   float *__restrict__ running_mean, float *__restrict__ running_var) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  double s = 0.0, q = 0.0;
  for (int b = 0; b < nblk; ++b) {
    s += partial[((long long)b * 2 + 0) * c + ch];
    q += partial[((long long)b * 2 + 1) * c + ch];
  }
  const double mean = s / (double)rows;
  double var = q / (double)rows - mean * mean;
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  const double g = gamma ? (double)gamma[ch] : 1.0, bt = beta ? (double)beta[ch] : 0.0;
  scale[ch] = (float)(g * invstd);
  shift[ch] = (float)(bt - mean * g * invstd);
  mean_out[ch] = (float)mean;
  invstd_out[ch] = (float)invstd;
  if (running_mean) {
    const double unbiased = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
    running_mean[ch] = (float)((1.0 - momentum) * (double)running_mean[ch] + momentum * mean);
    running_var[ch] = (float)((1.0 - momentum) * (double)running_var[ch] + momentum * unbiased);
  }
}

// ---- BatchNorm backward sums -> coefficients of dy = p*dz + q*y + r ------------------------------
// which: 1 -> dgamma from stat row 1, 2 -> from stat row 2 (second branch of the two-branch first layer)
__global__ void bn_bwd_finalize_kernel(int c, long long rows, int nblk, int nstat, int which,
                                       const double *__restrict__ partial, const float *__restrict__ scale,
                                       const float *__restrict__ mean, const float *__restrict__ invstd,
                                       float *__restrict__ p, float *__restrict__ q, float *__restrict__ r,
                                       float *__restrict__ dgamma, float *__restrict__ dbeta) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  double db = 0.0, dg = 0.0;
  for (int b = 0; b < nblk; ++b) {
    db += partial[((long long)b * nstat + 0) * c + ch];
    dg += partial[((long long)b * nstat + which) * c + ch];
  }
  const double s = scale[ch], is = invstd[ch], mu = mean[ch], m = (double)rows;
  // dy = s * (dz - db/m - yhat * dg/m),  yhat = (y - mu) * is
  const double qq = -s * is * dg / m;
  p[ch] = (float)s;
  q[ch] = (float)qq;
  r[ch] = (float)(-s * db / m - qq * mu);
  if (dgamma) dgamma[ch] = (float)dg;
  if (dbeta) dbeta[ch] = (float)db;
}

// ---- pooling over nsample, fused with the last BatchNorm + ReLU -----------------------------------
// out[g][c] = max_k relu(scale*y[g*ns+k][c] + shift), arg = first k attaining it
__global__ void __launch_bounds__(GM_THREADS)
pool_max_kernel(long long groups, int ns, int c, int relu, const float *__restrict__ y, const float *__restrict__ scale,
                const float *__restrict__ shift, float *__restrict__ out, int *__restrict__ arg) {
  const long long total = groups * c;
  for (long long e = (long long)blockIdx.x * GM_THREADS + threadIdx.x; e < total; e += (long long)gridDim.x * GM_THREADS) {
    const long long g = e / c;
    const int ch = (int)(e - g * c);
    const float s = scale ? scale[ch] : 1.f, t = shift ? shift[ch] : 0.f;
    float best = -INFINITY; int bi = 0;
    for (int k = 0; k < ns; ++k) {
      float z = fmaf(s, y[(g * ns + k) * c + ch], t);
      if (relu) z = fmaxf(z, 0.f);
      if (z > best) { best = z; bi = k; }
    }
    out[e] = best; arg[e] = bi;
  }
}

// v[g][c] = dout * (out > 0); partial sums {sum v, sum v * yhat[arg row]} per column (BatchNorm backward
// of the pooled layer computed from G x C data only)
__global__ void __launch_bounds__(GM_THREADS)
pool_max_bwd_kernel(long long groups, int ns, int c, const float *__restrict__ dout, const float *__restrict__ out,
                    const int *__restrict__ arg, const float *__restrict__ y, const float *__restrict__ mean,
                    const float *__restrict__ invstd, float *__restrict__ v, double *__restrict__ partial) {
  // one thread per column, grid-stride over group slabs: column sums stay in registers
  const int ch = blockIdx.y * GM_THREADS + threadIdx.x;
  if (ch >= c) return;
  const float mu = mean[ch], is = invstd[ch];
  double s0 = 0.0, s1 = 0.0;
  for (long long g = blockIdx.x; g < groups; g += gridDim.x) {
    const long long e = g * c + ch;
    const float val = out[e] > 0.f ? dout[e] : 0.f;
    v[e] = val;
    const float yy = y[(g * ns + arg[e]) * c + ch];
    s0 += (double)val;
    s1 += (double)(val * ((yy - mu) * is));
  }
  partial[((long long)blockIdx.x * 2 + 0) * c + ch] = s0;
  partial[((long long)blockIdx.x * 2 + 1) * c + ch] = s1;
}

// out[g][c] = sum_k y[g*ns+k][c]   (umbrella aggregation 'sum')
__global__ void __launch_bounds__(GM_THREADS)
pool_sum_kernel(long long groups, int ns, int c, const float *__restrict__ y, float *__restrict__ out) {
  const long long total = groups * c;
  for (long long e = (long long)blockIdx.x * GM_THREADS + threadIdx.x; e < total; e += (long long)gridDim.x * GM_THREADS) {
    const long long g = e / c;
    const int ch = (int)(e - g * c);
    float s = 0.f;
    for (int k = 0; k < ns; ++k) s += y[(g * ns + k) * c + ch];
    out[e] = s;
  }
}

int persistent_blocks(long long tiles, int tiles_n) {
  long long want = 512 / (tiles_n > 0 ? tiles_n : 1);   // ~2 workgroups per CU over the whole grid
  if (want < 64) want = 64;
  return (int)(tiles < want ? tiles : want);
}

int check_operand(const char *who, const RowOperand *o) {
  RS_REQUIRE(o, "%s: operand descriptor is NULL", who);
  RS_REQUIRE(o->mode >= OPM_ID && o->mode <= OPM_BCAST, "%s: unknown operand mode %d", who, o->mode);
  RS_REQUIRE(o->a, "%s: operand tensor is NULL", who);
  if (o->mode == OPM_RELU1) RS_REQUIRE(o->s1 && o->t1, "%s: RELU1 operand needs scale/shift", who);
  if (o->mode == OPM_RELU2) RS_REQUIRE(o->b && o->s1 && o->t1 && o->s2 && o->t2, "%s: RELU2 operand needs two tensors and two scale/shift pairs", who);
  if (o->mode == OPM_AFF2) RS_REQUIRE(o->b && o->s1 && o->t1 && o->s2, "%s: AFF2 operand needs dz, y and p/q/r", who);
  if (o->mode == OPM_POOLED) RS_REQUIRE(o->b && o->s1 && o->t1 && o->s2 && o->arg && o->ns > 0, "%s: POOLED operand needs v, arg, y, p/q/r, nsample", who);
  if (o->mode == OPM_BCAST) RS_REQUIRE(o->ns > 0, "%s: BCAST operand needs nsample", who);
  return RS_OK;
}

}  // namespace

extern "C" int rs_mlp_gemm_rows(long long rows, int kdim, int cols, const rs_row_operand *x, const float *w, int ldw,
                                int w_is_k_by_n, const rs_mlp_epilogue *epi, void *stream) {
  RS_REQUIRE(rows >= 0 && kdim >= 0 && cols >= 0, "rs_mlp_gemm_rows: negative size");
  if (rows == 0 || cols == 0) return RS_OK;
  RS_REQUIRE(kdim > 0, "rs_mlp_gemm_rows: empty reduction dimension");
  RS_REQUIRE(w && epi && epi->out, "rs_mlp_gemm_rows: null pointer");
  int rc = check_operand("rs_mlp_gemm_rows", x);
  if (rc != RS_OK) return rc;
  Epilogue ep = *epi;
  RowOperand E = *x;
  if (E.ns <= 0) E.ns = 1;
  const int epi_mode = ep.mode;
  RS_REQUIRE(epi_mode >= EPI_STORE && epi_mode <= EPI_MASK, "rs_mlp_gemm_rows: unknown epilogue %d", epi_mode);
  if (epi_mode != EPI_STORE) RS_REQUIRE(ep.partial && ep.partial_blocks > 0, "rs_mlp_gemm_rows: statistics need a partial buffer");
  if (epi_mode == EPI_MASK) RS_REQUIRE(ep.my1 && ep.ms1 && ep.mt1 && ep.mean1 && ep.invstd1, "rs_mlp_gemm_rows: mask epilogue needs the producing layer's y/scale/shift/mean/invstd");
  if (epi_mode == EPI_MASK && ep.my2) RS_REQUIRE(ep.ms2 && ep.mt2 && ep.mean2 && ep.invstd2, "rs_mlp_gemm_rows: second mask branch incomplete");
  const int nstat = (epi_mode == EPI_MASK && ep.my2) ? 3 : 2;
  const long long tiles = (rows + GM_BM - 1) / GM_BM;
  const int bn = cols <= 32 ? 32 : (cols <= 64 ? 64 : 128);
  const int tiles_n = rs_cdiv(cols, bn);
  int gx = persistent_blocks(tiles, tiles_n);
  if (epi_mode != EPI_STORE) gx = gx < ep.partial_blocks ? gx : ep.partial_blocks;
  const dim3 grid(gx, tiles_n), block(GM_THREADS);
  hipStream_t st = (hipStream_t)stream;
  if (epi_mode != EPI_STORE && gx < ep.partial_blocks)   // unused partial rows must read as zero
    hipMemsetAsync(ep.partial + (long long)gx * nstat * cols, 0, sizeof(double) * (size_t)(ep.partial_blocks - gx) * nstat * cols, st);

#define RS_GEMM(BN_, EPI_, TW_) hipLaunchKernelGGL((gemm_rows_kernel<BN_, EPI_, TW_>), grid, block, 0, st, rows, kdim, cols, E, w, ldw, ep)
#define RS_GEMM_BN(EPI_, TW_) do { if (bn == 32) RS_GEMM(32, EPI_, TW_); else if (bn == 64) RS_GEMM(64, EPI_, TW_); else RS_GEMM(128, EPI_, TW_); } while (0)
  if (w_is_k_by_n) {
    if (epi_mode == EPI_STORE) RS_GEMM_BN(EPI_STORE, true);
    else if (epi_mode == EPI_STATS) RS_GEMM_BN(EPI_STATS, true);
    else RS_GEMM_BN(EPI_MASK, true);
  } else {
    if (epi_mode == EPI_STORE) RS_GEMM_BN(EPI_STORE, false);
    else if (epi_mode == EPI_STATS) RS_GEMM_BN(EPI_STATS, false);
    else RS_GEMM_BN(EPI_MASK, false);
  }
#undef RS_GEMM_BN
#undef RS_GEMM
  RS_CHECK_LAUNCH("rs_mlp_gemm_rows");
  return RS_OK;
}

extern "C" int rs_mlp_wgrad(long long rows, int ncols, int kcols, const rs_row_operand *p, const rs_row_operand *q,
                            float *partial, int chunks, float *dw, void *stream) {
  RS_REQUIRE(rows >= 0 && ncols >= 0 && kcols >= 0 && chunks > 0, "rs_mlp_wgrad: bad size");
  if (ncols == 0 || kcols == 0) return RS_OK;
  RS_REQUIRE(partial && dw, "rs_mlp_wgrad: null pointer");
  int rc = check_operand("rs_mlp_wgrad(P)", p);
  if (rc != RS_OK) return rc;
  rc = check_operand("rs_mlp_wgrad(Q)", q);
  if (rc != RS_OK) return rc;
  RowOperand P = *p, Q = *q;
  if (P.ns <= 0) P.ns = 1;
  if (Q.ns <= 0) Q.ns = 1;
  long long rpc = (rows + chunks - 1) / chunks;
  rpc = (rpc + WG_BR - 1) / WG_BR * WG_BR;
  hipStream_t st = (hipStream_t)stream;
  const dim3 block(GM_THREADS);
  if (kcols > 64) {          // 128 x 128 output block: waves 2 x 2, 2 x 2 tiles each
    const dim3 grid(chunks, rs_cdiv(ncols, 128), rs_cdiv(kcols, 128));
    hipLaunchKernelGGL((wgrad_kernel<2, 2, 2, 2>), grid, block, 0, st, rows, ncols, kcols, P, Q, rpc, partial);
  } else if (kcols > 32) {   // 128 x 64: waves 4 x 1, 1 x 2 tiles
    const dim3 grid(chunks, rs_cdiv(ncols, 128), 1);
    hipLaunchKernelGGL((wgrad_kernel<4, 1, 1, 2>), grid, block, 0, st, rows, ncols, kcols, P, Q, rpc, partial);
  } else {                   // 128 x 32: waves 4 x 1, 1 x 1 tile
    const dim3 grid(chunks, rs_cdiv(ncols, 128), 1);
    hipLaunchKernelGGL((wgrad_kernel<4, 1, 1, 1>), grid, block, 0, st, rows, ncols, kcols, P, Q, rpc, partial);
  }
  const long long n = (long long)ncols * kcols;
  long long rb = (n + GM_THREADS - 1) / GM_THREADS;
  if (rb > 1024) rb = 1024;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((int)rb), block, 0, st, chunks, n, partial, dw);
  RS_CHECK_LAUNCH("rs_mlp_wgrad");
  return RS_OK;
}

extern "C" int rs_bn_finalize(int c, long long rows, int nblk, const double *partial, const float *gamma,
                              const float *beta, float eps, float momentum, float *scale, float *shift,
                              float *save_mean, float *save_invstd, float *running_mean, float *running_var,
                              void *stream) {
  RS_REQUIRE(c >= 0 && rows > 0 && nblk > 0, "rs_bn_finalize: bad size");
  if (c == 0) return RS_OK;
  RS_REQUIRE(partial && scale && shift && save_mean && save_invstd, "rs_bn_finalize: null pointer");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(rs_cdiv(c, 128)), dim3(128), 0, (hipStream_t)stream, c, rows, nblk,
                     partial, gamma, beta, eps, momentum, scale, shift, save_mean, save_invstd, running_mean, running_var);
  RS_CHECK_LAUNCH("rs_bn_finalize");
  return RS_OK;
}

extern "C" int rs_bn_backward_finalize(int c, long long rows, int nblk, int nstat, int which, const double *partial,
                                       const float *scale, const float *mean, const float *invstd, float *p,
                                       float *q, float *r, float *dgamma, float *dbeta, void *stream) {
  RS_REQUIRE(c >= 0 && rows > 0 && nblk > 0 && nstat >= 2 && which >= 1 && which < nstat, "rs_bn_backward_finalize: bad size");
  if (c == 0) return RS_OK;
  RS_REQUIRE(partial && scale && mean && invstd && p && q && r, "rs_bn_backward_finalize: null pointer");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(rs_cdiv(c, 128)), dim3(128), 0, (hipStream_t)stream, c, rows, nblk,
                     nstat, which, partial, scale, mean, invstd, p, q, r, dgamma, dbeta);
  RS_CHECK_LAUNCH("rs_bn_backward_finalize");
  return RS_OK;
}

extern "C" int rs_pool_max(long long groups, int nsample, int c, int relu, const float *y, const float *scale,
                           const float *shift, float *out, int *arg, void *stream) {
  RS_REQUIRE(groups >= 0 && nsample > 0 && c >= 0, "rs_pool_max: bad size");
  if (groups == 0 || c == 0) return RS_OK;
  RS_REQUIRE(y && out && arg, "rs_pool_max: null pointer");
  long long blocks = (groups * c + GM_THREADS - 1) / GM_THREADS;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pool_max_kernel, dim3((int)blocks), dim3(GM_THREADS), 0, (hipStream_t)stream, groups, nsample, c,
                     relu, y, scale, shift, out, arg);
  RS_CHECK_LAUNCH("rs_pool_max");
  return RS_OK;
}

extern "C" int rs_pool_max_backward(long long groups, int nsample, int c, const float *dout, const float *out,
                                    const int *arg, const float *y, const float *mean, const float *invstd,
                                    float *v, double *partial, int partial_blocks, void *stream) {
  RS_REQUIRE(groups >= 0 && nsample > 0 && c >= 0 && partial_blocks > 0, "rs_pool_max_backward: bad size");
  if (groups == 0 || c == 0) return RS_OK;
  RS_REQUIRE(dout && out && arg && y && mean && invstd && v && partial, "rs_pool_max_backward: null pointer");
  int gx = (int)(groups < partial_blocks ? groups : partial_blocks);
  hipStream_t st = (hipStream_t)stream;
  if (gx < partial_blocks)
    hipMemsetAsync(partial + (long long)gx * 2 * c, 0, sizeof(double) * (size_t)(partial_blocks - gx) * 2 * c, st);
  hipLaunchKernelGGL(pool_max_bwd_kernel, dim3(gx, rs_cdiv(c, GM_THREADS)), dim3(GM_THREADS), 0, st, groups, nsample, c,
                     dout, out, arg, y, mean, invstd, v, partial);
  RS_CHECK_LAUNCH("rs_pool_max_backward");
  return RS_OK;
}

extern "C" int rs_pool_sum(long long groups, int nsample, int c, const float *y, float *out, void *stream) {
  RS_REQUIRE(groups >= 0 && nsample > 0 && c >= 0, "rs_pool_sum: bad size");
  if (groups == 0 || c == 0) return RS_OK;
  RS_REQUIRE(y && out, "rs_pool_sum: null pointer");
  long long blocks = (groups * c + GM_THREADS - 1) / GM_THREADS;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pool_sum_kernel, dim3((int)blocks), dim3(GM_THREADS), 0, (hipStream_t)stream, groups, nsample, c, y, out);
  RS_CHECK_LAUNCH("rs_pool_sum");
  return RS_OK;
}
