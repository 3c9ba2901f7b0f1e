// umbrella_mfma.hip — the 10-channel shared MLP of UmbrellaSurfaceConstructor on the fp32 matrix pipe (gfx950), round 4.
//
// Reference: classification/modules/repsurface_utils.py:266-274,296-305 (Conv2d(10,10,bias=False)-BN-ReLU-Conv2d(10,10)-BN-ReLU-
// Conv2d(10,10), sum over the fan) and segmentation/modules/repsurface_utils.py:298-303,323-327 (Conv1d-BN-ReLU-Conv1d, sum).
//
// The register-resident VALU passes of umbrella_mlp.hip keep a row per THREAD: 110 weight-gradient accumulators per thread, 256
// VGPRs, one wave per SIMD, every k-step behind a broadcast LDS read of its weights -- 14-29 us per pass for 5 us of arithmetic.
// Here a WAVE owns 16 points at a time and walks their fan rows k = 0 .. group-1 as 16-row MFMA tiles (v_mfma_f32_16x16x4_f32):
//   * activations live in one of two register layouts, both 4 VGPRs per tensor and tile (l = lane & 15, g = lane >> 4):
//       L-R "lanes along rows":     reg i = T[point l][channel 4g + i]      = the D layout of  D[ch][row]  = W . T^T
//       L-C "lanes along channels": reg i = T[point 4g + i][channel l]      = the D layout of  D[row][ch]  = T . W^T
//     an L-R tensor is, register for register, the B operand of the next L-R product AND the A operand of an L-C product (the k
//     order of an MFMA chain is free as long as both operands agree: step s carries channel 4g + s), so a layer chain needs NO
//     cross-lane movement, and the SAME weight fragment (lane (l,g), step s: W[l][4g+s]) serves both orientations;
//   * sums over rows (weight gradients, the BatchNorm-backward moments) are MFMAs too, with the ROWS as the reduction index: two
//     L-C tensors are exactly the A and B operands of  S[chP][chQ] += sum_rows P[row][chP] Q[row][chQ]  (step i carries row 4g+i);
//     a 10 x 10 gradient is 4 accumulator VGPRs for the life of the wave;
//   * weights, BatchNorm vectors: registers, loaded once per wave.  No LDS in the loop, ~100 VGPRs, 4 waves per SIMD.
// Passes that only existed to produce sums that are LINEAR in the input are gone:
//   * BatchNorm 0 sees y0 = W0 x (+ b0): its batch mean / variance follow from the first and second moments of x (sum x, sum x x^T;
//     65 numbers per batch, computed once in the geometry stage by rs_umbrella_moments): no statistics pass over y0;
//   * dW0 = sum dy0 x^T with dy0 = s0 (dz0 - mean(dz0) - yhat0 mean(dz0 yhat0)) = s0 (T1 - db0/m Sx - dg0/m G), T1 = sum dz0 x^T
//     accumulated by the pass that produces dz0, G = sum yhat0 x^T from the moments: no third backward pass; dW1 likewise from
//     sum dz1 a0^T, sum yhat1 a0^T, sum a0.
//   * the BatchNorm finalize launches between the passes are prologues of the consuming pass (every workgroup reduces the few
//     partial rows itself, fixed order, identical in every workgroup; workgroup 0 publishes the vectors / running statistics).
// Three-layer (classification): F1 (statistics of y1) -> F2 (output) | B1 -> B2 -> FIN: 5 launches against 6 passes + 4 finalizes
// + reductions; two-layer (segmentation): F2 | B2 -> FIN.  (Merging the passes of a direction into ONE launch whose workgroups meet
// at a grid-wide barrier was built and measured in round 4 -- tools/probes/grid_meet.hip: 9.4 us per meeting with release /
// acquire fences, 6.1 with write-through stores and relaxed polling, 4.3 with per-XCD arrival counters -- and LOST to the kernel
// boundary it replaces: forward 30 us merged against 25 us as two launches, backward 80 against 59.  Not kept.)  Partial sums: fp32 inside a workgroup (<= a few thousand rows), fp64
// across workgroups, fixed order everywhere (deterministic).
#include "rs_common.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int C = RS_UMB_C;           // 10 channels
constexpr int CP = 16;                // padded (one MFMA tile edge)
constexpr int TH = 512;               // 8 waves per workgroup: two per SIMD when a workgroup has a CU to itself
constexpr int NW = TH / 64;
constexpr int TILE = C * CP;          // rows m < 10 of a 16 x 16 accumulator tile
constexpr int B1_ROW = RS_UMB_B1_ROW; // [dW2][S1 = sum dz1 a0^T][Sy = sum yhat1 a0^T][db1][dg1][sa0][db2]
constexpr int B2_ROW = RS_UMB_B2_ROW; // [T1 = sum dz0 x^T][dWlast (two-layer)][db0][dg0][dblast]
constexpr int MOM_ROW = RS_UMB_MOM_ROW;  // 11 x 16: S[m][n] = sum x_m x_n, index 10 = the constant 1
static_assert(B1_ROW == 3 * TILE + 4 * CP && B2_ROW == 2 * TILE + 3 * CP && MOM_ROW == 11 * CP, "row layouts");

__device__ __forceinline__ f4 mfma(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct Frag { float s[4]; };
// Weight fragments: unconditional loads from clamped addresses, zeroed by a select (a load under `if` is a branch per load, and the
// 64-bit addresses the compiler keeps live across those branches spilled the tile's rows to scratch).  W == NULL (a layer the
// variant does not have): any valid matrix stands in, the fragment is never used.
// y = W v: lane (l, g), step s holds W[out = l][in = 4g + s]
__device__ __forceinline__ Frag frag_w(const float *W, int l, int g) {
  Frag f;
  const int lc = l < C ? l : C - 1;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int ch = 4 * g + s, cc = ch < C ? ch : C - 1;
    const float v = W[lc * C + cc];
    f.s[s] = (l < C && ch < C) ? v : 0.f;
  }
  return f;
}
// da = W^T dy: step s holds W[out = 4g + s][in = l]
__device__ __forceinline__ Frag frag_wt(const float *W, int l, int g) {
  Frag f;
  const int lc = l < C ? l : C - 1;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int ch = 4 * g + s, cc = ch < C ? ch : C - 1;
    const float v = W[cc * C + lc];
    f.s[s] = (l < C && ch < C) ? v : 0.f;
  }
  return f;
}
// the identity as a weight fragment: mm_lc(v, I) re-lays an L-R tensor out as L-C (exact: one product per element, the rest zeros)
__device__ __forceinline__ Frag frag_id(int l, int g) {
  Frag f;
#pragma unroll
  for (int s = 0; s < 4; ++s) f.s[s] = (l == 4 * g + s && l < C) ? 1.f : 0.f;
  return f;
}
// rows as the N index: D[ch 4g+i][row l] = sum_k W[ch][k] v[row][k]   (v: L-R; result: L-R)
__device__ __forceinline__ f4 mm_lr(const Frag &w, f4 v, f4 c) {
#pragma unroll
  for (int s = 0; s < 4; ++s) c = mfma(w.s[s], v[s], c);
  return c;
}
// rows as the M index: D[row 4g+i][ch l]   (v: L-R; result: L-C)
__device__ __forceinline__ f4 mm_lc(f4 v, const Frag &w, f4 c) {
#pragma unroll
  for (int s = 0; s < 4; ++s) c = mfma(v[s], w.s[s], c);
  return c;
}
// S[chP 4g+r][chQ l] += sum over the tile's 16 rows of P[row][chP] Q[row][chQ]   (P, Q: L-C)
__device__ __forceinline__ f4 mm_rows(f4 p, f4 q, f4 c) {
#pragma unroll
  for (int i = 0; i < 4; ++i) c = mfma(p[i], q[i], c);
  return c;
}
__device__ __forceinline__ f4 splat(float v) { f4 r = {v, v, v, v}; return r; }

// a (points, 10) or (rows, 10) row in L-R form: lane (l, g) holds channels 4g .. 4g+3 of ITS row (g = 2: channels 8, 9; g = 3: none).
// Unconditional loads from clamped in-row addresses; what a lane holds beyond channel 9 is real data of the row (finite) and only
// ever meets zero weights.
__device__ __forceinline__ f4 ld_lr(const float *base, long long row, int g) {
  const float2 *p = reinterpret_cast<const float2 *>(base + row * C);      // rows are 40 B: 8-byte aligned
  const int gg = g < 2 ? g : 2;
  const float2 a = p[2 * gg], b = p[g < 2 ? 2 * g + 1 : 4];
  f4 v = {a.x, a.y, b.x, b.y};
  return v;
}

struct Sh {
  float sc0[CP], sh0[CP], mu0[CP], is0[CP];
  float sc1[CP], sh1[CP], mu1[CP], is1[CP];
  float b0[CP], c1[CP], c2[CP];
  float m1[CP], m2[CP];                 // db1 / rows, dg1 / rows
  double slice[16][32];
  double fin[64];
  float wred[NW][B1_ROW];
};

__device__ __forceinline__ f4 vec_r(const float *v, int g) { f4 r = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]}; return r; }

__device__ void stage_vectors(Sh &L, const rs_umbrella_mfma &m) {
  const int t = threadIdx.x;
  if (t < CP) {
    const bool ok = t < C;
    L.b0[t] = (ok && m.b0) ? m.b0[t] : 0.f;
    L.c1[t] = (ok && m.b1) ? m.b1[t] : 0.f;
    L.c2[t] = (ok && m.b2) ? m.b2[t] : 0.f;
  }
}
// (4, 10) = scale, shift, mean, invstd as published by an earlier pass
__device__ void load_bn(float *sc, float *sh, float *mu, float *is, const float *bn) {
  const int t = threadIdx.x;
  if (t < CP) {
    const bool ok = t < C;
    sc[t] = ok ? bn[t] : 0.f;
    sh[t] = ok ? bn[C + t] : 0.f;
    mu[t] = ok ? bn[2 * C + t] : 0.f;
    is[t] = ok ? bn[3 * C + t] : 0.f;
  }
}
__device__ __forceinline__ void publish_bn(int ch, double mean, double var, double rows, double gam, double bet, double invstd,
                                           float *bn, float *run_mean, float *run_var, float momentum) {
  bn[ch] = (float)(gam * invstd);
  bn[C + ch] = (float)(bet - mean * gam * invstd);
  bn[2 * C + ch] = (float)mean;
  bn[3 * C + ch] = (float)invstd;
  if (run_mean) {
    const double unbiased = rows > 1.0 ? var * rows / (rows - 1.0) : var;
    run_mean[ch] = (float)((1.0 - momentum) * (double)run_mean[ch] + momentum * mean);
    run_var[ch] = (float)((1.0 - momentum) * (double)run_var[ch] + momentum * unbiased);
  }
}

// BatchNorm 0 of y0 = W0 x (+ b0) from the moments of x: mean = W0 Sx / n + b0, E[(y - b0)^2] = w^T Sxx w / n  (fp64).
// Thread (c, k), c, k < 10, takes row k of the quadratic form (10 loads + 10 fused multiply-adds), thread c sums its ten rows.
__device__ void bn0_from_moments(Sh &L, const rs_umbrella_mfma &m, bool publish) {
  const int t = threadIdx.x;
  const double *S = m.moments;
  if (t < C * C) {
    const int c = t / C, k = t % C;
    double r = 0.0;
#pragma unroll
    for (int k2 = 0; k2 < C; ++k2) r += (double)m.w0[c * C + k2] * S[k * CP + k2];
    const double wk = (double)m.w0[c * C + k];
    L.slice[c][k] = wk * r;                       // w_k (Sxx w)_k
    L.slice[c][C + k] = wk * S[10 * CP + k];      // w_k Sx_k
  }
  __syncthreads();
  if (t < CP) {
    float sc = 0.f, sh = 0.f, mu = 0.f, is = 0.f;
    if (t < C) {
      const double n = S[10 * CP + 10];
      double lin = 0.0, quad = 0.0;
#pragma unroll
      for (int k = 0; k < C; ++k) { quad += L.slice[t][k]; lin += L.slice[t][C + k]; }
      const double ml = lin / n;
      double var = quad / n - ml * ml;
      if (var < 0.0) var = 0.0;
      const double mean = ml + (m.b0 ? (double)m.b0[t] : 0.0);
      const double invstd = 1.0 / sqrt(var + (double)m.eps0);
      const double gam = m.gamma0 ? (double)m.gamma0[t] : 1.0, bet = m.beta0 ? (double)m.beta0[t] : 0.0;
      sc = (float)(gam * invstd);
      sh = (float)(bet - mean * gam * invstd);
      mu = (float)mean;
      is = (float)invstd;
      if (publish) publish_bn(t, mean, var, n, gam, bet, invstd, m.bn0, m.run_mean0, m.run_var0, m.mom0);
    }
    L.sc0[t] = sc; L.sh0[t] = sh; L.mu0[t] = mu; L.is0[t] = is;
  }
  __syncthreads();
}

// sum of 32 values per partial row over `nblk` rows (row pitch `pitch` elements), fixed order: 16 slices of rows (each thread batches 8
// independent loads per trip: the prologue is a couple of memory latencies deep, not nblk / 16), then the slices -> L.fin[0..32)
template <typename T>
__device__ void reduce32(Sh &L, const T *part, int nblk, int pitch) {
  const int t = threadIdx.x, v = t & 31, sl = t >> 5;
  double a = 0.0;
  for (int b = sl; b < nblk; b += 16 * 8) {
    double x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = b + 16 * u < nblk ? (double)part[(long long)(b + 16 * u) * pitch + v] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) a += x[u];
  }
  L.slice[sl][v] = a;
  __syncthreads();
  if (t < 32) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += L.slice[k][t];
    L.fin[t] = s;
  }
  __syncthreads();
}

// BatchNorm 1 from the F1 partials {sum y1, sum y1^2} (nblk, 2, 16) fp64
__device__ void bn1_from_partials(Sh &L, const rs_umbrella_mfma &m, bool publish) {
  reduce32<double>(L, m.stat, m.nblk_f1, 32);
  const int t = threadIdx.x;
  if (t < CP) {
    float sc = 0.f, sh = 0.f, mu = 0.f, is = 0.f;
    if (t < C) {
      const double n = (double)m.rows;
      const double mean = L.fin[t] / n;
      double var = L.fin[CP + t] / n - mean * mean;
      if (var < 0.0) var = 0.0;
      const double invstd = 1.0 / sqrt(var + (double)m.eps1);
      const double gam = m.gamma1 ? (double)m.gamma1[t] : 1.0, bet = m.beta1 ? (double)m.beta1[t] : 0.0;
      sc = (float)(gam * invstd);
      sh = (float)(bet - mean * gam * invstd);
      mu = (float)mean;
      is = (float)invstd;
      if (publish) publish_bn(t, mean, var, n, gam, bet, invstd, m.bn1, m.run_mean1, m.run_var1, m.mom1);
    }
    L.sc1[t] = sc; L.sh1[t] = sh; L.mu1[t] = mu; L.is1[t] = is;
  }
}

// one workgroup row of partial sums out of the waves' rows (fixed order: waves ascending)
__device__ void store_partial_row(Sh &L, float *dst, int n) {
  __syncthreads();
  for (int e = threadIdx.x; e < n; e += TH) {
    float s = L.wred[0][e];
#pragma unroll
    for (int w = 1; w < NW; ++w) s += L.wred[w][e];
    dst[e] = s;
  }
}
__device__ __forceinline__ void put_tile(float *row, f4 acc, int l, int g) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (4 * g + r < C) row[(4 * g + r) * CP + l] = acc[r];
}
// per-lane sums of channel l over the lane's rows -> sum over the four lane groups (fixed order) -> row[l]
__device__ __forceinline__ void put_vec(float *row, float v, int l, int g) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  if (g == 0) row[l] = v;
}
__device__ __forceinline__ f4 relu_affine(f4 y, f4 s, f4 t) {
  f4 a;
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = fmaxf(fmaf(s[i], y[i], t[i]), 0.f);
  return a;
}
__device__ __forceinline__ f4 relu_affine(f4 y, float s, float t) {
  f4 a;
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = fmaxf(fmaf(s, y[i], t), 0.f);
  return a;
}

struct Walk {      // the tiles of one wave: 16 points each, interleaved over all waves of the grid
  long long points, tiles, tile, step;
  int l, g, wave;
  __device__ Walk(const rs_umbrella_mfma &m) {
    points = m.rows / m.group;
    tiles = (points + 15) >> 4;
    wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    l = lane & 15;
    g = lane >> 4;
    tile = (long long)blockIdx.x * NW + wave;
    step = (long long)gridDim.x * NW;
  }
};

// The G fan rows of a tile's 16 points, L-R form, ALL requested before anything waits for them: a wave has one or two tiles, so
// per-row prefetching left every sub-tile behind its own memory latency (the first version: 18 us per pass for 2-6 us of MFMAs).
template <int G>
__device__ __forceinline__ void ld_tile_lr(const float *x, long long pl, int g, f4 (&xr)[G]) {
#pragma unroll
  for (int k = 0; k < G; ++k) xr[k] = ld_lr(x, pl * G + k, g);
}

// ---------------------------------------------------------------------------------------------------- moments of x
// 4 tiles (64 rows) per trip, their 16 loads in flight together
__global__ void __launch_bounds__(TH)
umb_moments_kernel(const float *__restrict__ x, long long rows, float *__restrict__ partial) {
  __shared__ float red[NW][MOM_ROW];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l = lane & 15, g = lane >> 4;
  const int lc = l < C ? l : C - 1;
  const float one = l == C ? 1.f : 0.f;
  f4 acc = splat(0.f);
  const long long quads = (rows + 63) >> 6;
  for (long long q = (long long)blockIdx.x * NW + wave; q < quads; q += (long long)gridDim.x * NW) {
    f4 xa[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long r = q * 64 + u * 16 + 4 * g + i;
        const float v = x[(r < rows ? r : rows - 1) * C + lc];
        xa[u][i] = r < rows ? (l < C ? v : one) : 0.f;
      }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = mm_rows(xa[u], xa[u], acc);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (4 * g + r <= C) red[wave][(4 * g + r) * CP + l] = acc[r];
  __syncthreads();
  for (int e = threadIdx.x; e < MOM_ROW; e += TH) {
    float s = red[0][e];
#pragma unroll
    for (int w = 1; w < NW; ++w) s += red[w][e];
    partial[(long long)blockIdx.x * MOM_ROW + e] = s;
  }
}
// (nblk, 176) fp32 -> (176) fp64: 5 slices of rows x 176 values (880 of 1024 threads), 8 loads per trip, then the slices in order
__global__ void __launch_bounds__(1024)
umb_moments_reduce_kernel(const float *__restrict__ partial, int nblk, double *__restrict__ out) {
  __shared__ double sl[5][MOM_ROW];
  const int t = threadIdx.x, v = t % MOM_ROW, s = t / MOM_ROW;
  if (s < 5) {
    double a = 0.0;
    for (int b = s; b < nblk; b += 5 * 8) {
      double x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = b + 5 * u < nblk ? (double)partial[(long long)(b + 5 * u) * MOM_ROW + v] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) a += x[u];
    }
    sl[s][v] = a;
  }
  __syncthreads();
  if (t < MOM_ROW) out[t] = (((sl[0][t] + sl[1][t]) + sl[2][t]) + sl[3][t]) + sl[4][t];
}

// ---------------------------------------------------------------------------------------------------- the four tile loops
// Each loop walks the wave's tiles (first one: rows already in xr, requested by the kernel before its prologue); the next tile's
// rows are requested under the current tile's tail.  `pl`: the wave's L-R point of the current tile (clamped).
template <int G>
__device__ __forceinline__ void reload_first(const rs_umbrella_mfma &m, const Walk &w, f4 (&xr)[G], long long &pl) {
  const long long pn = w.tile * 16 + w.l;
  pl = pn < w.points ? pn : w.points - 1;
  if (w.tile < w.tiles) ld_tile_lr<G>(m.x, pl, w.g, xr);
}
template <int G>
__device__ __forceinline__ void next_tile(const rs_umbrella_mfma &m, const Walk &w, long long tile, f4 (&xr)[G], long long &pl) {
  if (tile + w.step < w.tiles) {
    const long long pn = (tile + w.step) * 16 + w.l;
    pl = pn < w.points ? pn : w.points - 1;
    ld_tile_lr<G>(m.x, pl, w.g, xr);
  }
}

// F1: per-lane sums of y1 and y1^2 (channel l, the lane's rows)
template <int G>
__device__ __forceinline__ void f1_loop(const rs_umbrella_mfma &m, const Sh &L, const Walk &w, f4 (&xr)[G], long long &pl,
                                        const Frag &W0, const Frag &W1, float &sum, float &sq) {
  const int l = w.l, g = w.g;
  const f4 s0r = vec_r(L.sc0, g), t0r = vec_r(L.sh0, g), b0r = vec_r(L.b0, g);
  const float c1c = L.c1[l];
  for (long long tile = w.tile; tile < w.tiles; tile += w.step) {
    const long long p0 = tile * 16;
    const bool tail = p0 + 16 > w.points;
    f4 y1[G];
#pragma unroll
    for (int k = 0; k < G; ++k) {
      const f4 a0 = relu_affine(mm_lr(W0, xr[k], b0r), s0r, t0r);
      y1[k] = mm_lc(a0, W1, splat(c1c));
    }
    next_tile<G>(m, w, tile, xr, pl);
#pragma unroll
    for (int k = 0; k < G; ++k) {
      if (tail)
#pragma unroll
        for (int i = 0; i < 4; ++i) y1[k][i] = p0 + 4 * g + i < w.points ? y1[k][i] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { sum += y1[k][i]; sq = fmaf(y1[k][i], y1[k][i], sq); }
    }
  }
}
__device__ __forceinline__ void f1_store(Sh &L, const rs_umbrella_mfma &m, const Walk &w, float sum, float sq) {
  put_vec(&L.wred[w.wave][0], sum, w.l, w.g);
  put_vec(&L.wred[w.wave][CP], sq, w.l, w.g);
  __syncthreads();
  if (threadIdx.x < 32) {
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < NW; ++k) a += (double)L.wred[k][threadIdx.x];
    m.stat[(long long)blockIdx.x * 32 + threadIdx.x] = a;
  }
}

// F2: out = out_scale * sum over the fan of the last layer
template <int LAYERS, int G>
__device__ __forceinline__ void f2_loop(const rs_umbrella_mfma &m, const Sh &L, const Walk &w, f4 (&xr)[G], long long &pl,
                                        const Frag &W0, const Frag &W1, const Frag &W2) {
  const int l = w.l, g = w.g;
  const f4 s0r = vec_r(L.sc0, g), t0r = vec_r(L.sh0, g), b0r = vec_r(L.b0, g);
  const f4 s1r = vec_r(L.sc1, g), t1r = vec_r(L.sh1, g), c1r = vec_r(L.c1, g);
  const float c1c = L.c1[l], c2c = L.c2[l];
  for (long long tile = w.tile; tile < w.tiles; tile += w.step) {
    const long long p0 = tile * 16;
    f4 acc = splat(0.f);
#pragma unroll
    for (int k = 0; k < G; ++k) {
      const f4 a0 = relu_affine(mm_lr(W0, xr[k], b0r), s0r, t0r);
      if (LAYERS == 3) {
        const f4 a1 = relu_affine(mm_lr(W1, a0, c1r), s1r, t1r);
        acc += mm_lc(a1, W2, splat(c2c));
      } else {
        acc += mm_lc(a0, W1, splat(c1c));
      }
    }
    next_tile<G>(m, w, tile, xr, pl);
    if (l < C)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long pc = p0 + 4 * g + i;
        if (pc < w.points) m.out[pc * C + l] = acc[i] * m.out_scale;
      }
  }
}

// B1 (three layers): dW2 = sum dy2^T a1, db2; BatchNorm-1 backward sums db1 = sum dz1, dg1 = sum dz1 yhat1; for dW1:
// S1 = sum dz1^T a0, Sy = sum yhat1^T a0, sa0 = sum a0.  Leaves the wave's row in L.wred[wave].
template <int G>
__device__ __forceinline__ void b1_loop(const rs_umbrella_mfma &m, Sh &L, const Walk &w, f4 (&xr)[G], long long &pl) {
  const int l = w.l, g = w.g, lc = l < C ? l : C - 1;
  const Frag W0 = frag_w(m.w0, l, g), W1 = frag_w(m.w1, l, g), W2t = frag_wt(m.w2, l, g);
  const f4 s0r = vec_r(L.sc0, g), t0r = vec_r(L.sh0, g);
  const float s0c = L.sc0[l], t0c = L.sh0[l], s1c = L.sc1[l], t1c = L.sh1[l], mu1c = L.mu1[l], is1c = L.is1[l], c1c = L.c1[l];
  f4 aW2 = splat(0.f), aS1 = splat(0.f), aSy = splat(0.f);
  float db1 = 0.f, dg1 = 0.f, sa0 = 0.f, db2 = 0.f;
  for (long long tile = w.tile; tile < w.tiles; tile += w.step) {
    const long long p0 = tile * 16;
    const f4 dy2r = ld_lr(m.dout, pl, g);
    f4 dy2c, vm;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long pc = p0 + 4 * g + i;
      vm[i] = pc < w.points ? 1.f : 0.f;
      const float v = m.dout[(pc < w.points ? pc : w.points - 1) * C + lc];
      dy2c[i] = (l < C && pc < w.points) ? v : 0.f;
    }
    const f4 z = splat(0.f);
    const f4 da1c = mm_lc(dy2r, W2t, z);                           // the same for every fan row of the point
#pragma unroll
    for (int k = 0; k < G; ++k) {
      const f4 y0r = mm_lr(W0, xr[k], z);
      f4 a0c = relu_affine(mm_lc(xr[k], W0, z), s0c, t0c);
      const f4 a0r = relu_affine(y0r, s0r, t0r);
      const f4 y1c = mm_lc(a0r, W1, splat(c1c));
      f4 a1c, yh1c, dz1c;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a1c[i] = fmaxf(fmaf(s1c, y1c[i], t1c), 0.f);
        yh1c[i] = (y1c[i] - mu1c) * is1c;
        dz1c[i] = a1c[i] > 0.f ? da1c[i] : 0.f;
      }
      a0c *= vm; yh1c *= vm; dz1c *= vm;          // (rows beyond the last point: zero; unconditional -- a branch per fan row splits the schedule)
      aW2 = mm_rows(dy2c, a1c, aW2);
      aS1 = mm_rows(dz1c, a0c, aS1);
      aSy = mm_rows(yh1c, a0c, aSy);
#pragma unroll
      for (int i = 0; i < 4; ++i) { db1 += dz1c[i]; dg1 = fmaf(dz1c[i], yh1c[i], dg1); sa0 += a0c[i]; }
    }
    db2 += (float)G * ((dy2c[0] + dy2c[1]) + (dy2c[2] + dy2c[3]));
    next_tile<G>(m, w, tile, xr, pl);
  }
  float *row = L.wred[w.wave];
  put_tile(row, aW2, l, g);
  put_tile(row + TILE, aS1, l, g);
  put_tile(row + 2 * TILE, aSy, l, g);
  put_vec(row + 3 * TILE, db1, l, g);
  put_vec(row + 3 * TILE + CP, dg1, l, g);
  put_vec(row + 3 * TILE + 2 * CP, sa0, l, g);
  put_vec(row + 3 * TILE + 3 * CP, db2, l, g);
}

// B2: T1 = sum dz0^T x, db0 = sum dz0, dg0 = sum dz0 yhat0;  two layers: also dW1 = sum dy1^T a0, db1 = sum dy1 (dy1 = dout[point]).
// Leaves the wave's row in L.wred[wave].  xr: the first tile's rows, already requested; x in L-C form (the B operand of T1) comes
// out of the matrix pipe too (mm_lc with the identity: 4 MFMAs instead of 4 strided loads per fan row and 4 more registers per row).
template <int LAYERS, int G>
__device__ __forceinline__ void b2_loop(const rs_umbrella_mfma &m, Sh &L, const Walk &w, f4 (&xr)[G], long long &pl) {
  const int l = w.l, g = w.g, lc = l < C ? l : C - 1;
  const Frag W0 = frag_w(m.w0, l, g), W1 = frag_w(m.w1, l, g), W1t = frag_wt(m.w1, l, g), W2t = frag_wt(m.w2 ? m.w2 : m.w0, l, g), Id = frag_id(l, g);
  const f4 s0r = vec_r(L.sc0, g), t0r = vec_r(L.sh0, g);
  const f4 s1r = vec_r(L.sc1, g), t1r = vec_r(L.sh1, g), c1r = vec_r(L.c1, g);
  // BatchNorm-1 backward as dy1 = p dz1 + q y1 + r:  p = s1, q = -s1 is1 dg1/m, r = -s1 db1/m - q mu1  (three vectors in registers)
  f4 q1r, r1r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    q1r[i] = -s1r[i] * L.is1[4 * g + i] * L.m2[4 * g + i];
    r1r[i] = -s1r[i] * L.m1[4 * g + i] - q1r[i] * L.mu1[4 * g + i];
  }
  const float s0c = L.sc0[l], t0c = L.sh0[l], mu0c = L.mu0[l], is0c = L.is0[l], b0c = L.b0[l];
  f4 aT1 = splat(0.f), aWl = splat(0.f);
  float db0 = 0.f, dg0 = 0.f, dbl = 0.f;
  for (long long tile = w.tile; tile < w.tiles; tile += w.step) {
    const long long p0 = tile * 16;
    const f4 dor = ld_lr(m.dout, pl, g);          // the incoming gradient of the point, L-R
    f4 doc = splat(0.f), vm;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long pc = p0 + 4 * g + i;
      vm[i] = pc < w.points ? 1.f : 0.f;
      if (LAYERS == 2) {
        const float v = m.dout[(pc < w.points ? pc : w.points - 1) * C + lc];
        doc[i] = (l < C && pc < w.points) ? v : 0.f;
      }
    }
    const f4 z = splat(0.f);
    f4 da1r = z, da0c2 = z;
    if (LAYERS == 3) da1r = mm_lr(W2t, dor, z);                     // the same for every fan row of the point
    else da0c2 = mm_lc(dor, W1t, z);
#pragma unroll
    for (int k = 0; k < G; ++k) {
      const f4 y0c = mm_lc(xr[k], W0, splat(b0c));
      f4 da0c;
      if (LAYERS == 3) {
        const f4 a0r = relu_affine(mm_lr(W0, xr[k], z), s0r, t0r);
        const f4 y1r = mm_lr(W1, a0r, c1r);
        f4 dy1r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float dz1 = fmaf(s1r[i], y1r[i], t1r[i]) > 0.f ? da1r[i] : 0.f;
          dy1r[i] = fmaf(s1r[i], dz1, fmaf(q1r[i], y1r[i], r1r[i]));
        }
        da0c = mm_lc(dy1r, W1t, z);
      } else {
        da0c = da0c2;
      }
      const f4 xq = mm_lc(xr[k], Id, z);
      f4 a0c, yh0c, dz0c;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a0c[i] = fmaxf(fmaf(s0c, y0c[i], t0c), 0.f);
        yh0c[i] = (y0c[i] - mu0c) * is0c;
        dz0c[i] = a0c[i] > 0.f ? da0c[i] : 0.f;
      }
      dz0c *= vm;
      aT1 = mm_rows(dz0c, xq, aT1);
      if (LAYERS == 2) aWl = mm_rows(doc, a0c, aWl);
#pragma unroll
      for (int i = 0; i < 4; ++i) { db0 += dz0c[i]; dg0 = fmaf(dz0c[i], yh0c[i], dg0); }
      // (no __builtin_amdgcn_sched_barrier between the fan rows: with one here the launch computed WRONG sums on gfx950 / ROCm 7.0 --
      //  dW0, dgamma0, dbeta0 off by O(1), found by tools A/B builds in round 4 -- so the schedule is left to the compiler)
    }
    if (LAYERS == 2) dbl += (float)G * ((doc[0] + doc[1]) + (doc[2] + doc[3]));
    next_tile<G>(m, w, tile, xr, pl);
  }
  float *row = L.wred[w.wave];
  put_tile(row, aT1, l, g);
  put_tile(row + TILE, aWl, l, g);
  put_vec(row + 2 * TILE, db0, l, g);
  put_vec(row + 2 * TILE + CP, dg0, l, g);
  put_vec(row + 2 * TILE + 2 * CP, dbl, l, g);
}
// db1 / rows, dg1 / rows out of the B1 partial rows (the BatchNorm-1 backward coefficients every lane needs)
__device__ void b1_coefficients(Sh &L, const rs_umbrella_mfma &m) {
  reduce32<float>(L, m.part_b1 + 3 * TILE, m.nblk_b1, B1_ROW);     // db1 (16), dg1 (16)
  if (threadIdx.x < CP) {
    const double n = (double)m.rows;
    L.m1[threadIdx.x] = threadIdx.x < C ? (float)(L.fin[threadIdx.x] / n) : 0.f;
    L.m2[threadIdx.x] = threadIdx.x < C ? (float)(L.fin[CP + threadIdx.x] / n) : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------- one launch per pass
template <int G>
__global__ void __launch_bounds__(TH, 3)
umb_f1_kernel(rs_umbrella_mfma m) {
  __shared__ Sh L;
  if (m.rows_dev) m.rows = min(m.rows, (long long)*m.rows_dev);      // (a packed batch under a captured capacity: include/repsurf_hip.h, rs_bn_item)
  Walk w(m);
  f4 xr[G];
  long long pl;
  reload_first<G>(m, w, xr, pl);                                   // in flight under the prologue
  const Frag W0 = frag_w(m.w0, w.l, w.g), W1 = frag_w(m.w1, w.l, w.g);
  stage_vectors(L, m);
  bn0_from_moments(L, m, blockIdx.x == 0);
  float sum = 0.f, sq = 0.f;
  f1_loop<G>(m, L, w, xr, pl, W0, W1, sum, sq);
  f1_store(L, m, w, sum, sq);
}

template <int LAYERS, int G>
__global__ void __launch_bounds__(TH, 3)
umb_f2_kernel(rs_umbrella_mfma m) {
  __shared__ Sh L;
  if (m.rows_dev) m.rows = min(m.rows, (long long)*m.rows_dev);      // (a packed batch under a captured capacity: include/repsurf_hip.h, rs_bn_item)
  Walk w(m);
  f4 xr[G];
  long long pl;
  reload_first<G>(m, w, xr, pl);
  const Frag W0 = frag_w(m.w0, w.l, w.g), W1 = frag_w(m.w1, w.l, w.g), W2 = frag_w(m.w2 ? m.w2 : m.w0, w.l, w.g);
  stage_vectors(L, m);
  if (LAYERS == 3) {
    load_bn(L.sc0, L.sh0, L.mu0, L.is0, m.bn0);
    bn1_from_partials(L, m, blockIdx.x == 0);
    __syncthreads();
  } else {
    bn0_from_moments(L, m, blockIdx.x == 0);
  }
  f2_loop<LAYERS, G>(m, L, w, xr, pl, W0, W1, W2);
}

template <int G>
__global__ void __launch_bounds__(TH, 3)
umb_b1_kernel(rs_umbrella_mfma m) {
  __shared__ Sh L;
  if (m.rows_dev) m.rows = min(m.rows, (long long)*m.rows_dev);      // (a packed batch under a captured capacity: include/repsurf_hip.h, rs_bn_item)
  Walk w(m);
  f4 xr[G];
  long long pl;
  reload_first<G>(m, w, xr, pl);
  stage_vectors(L, m);
  load_bn(L.sc0, L.sh0, L.mu0, L.is0, m.bn0);
  load_bn(L.sc1, L.sh1, L.mu1, L.is1, m.bn1);
  __syncthreads();
  b1_loop<G>(m, L, w, xr, pl);
  store_partial_row(L, m.part_b1 + (long long)blockIdx.x * B1_ROW, B1_ROW);
}

template <int LAYERS, int G>
__global__ void __launch_bounds__(TH, 3)
umb_b2_kernel(rs_umbrella_mfma m) {
  __shared__ Sh L;
  if (m.rows_dev) m.rows = min(m.rows, (long long)*m.rows_dev);      // (a packed batch under a captured capacity: include/repsurf_hip.h, rs_bn_item)
  Walk w(m);
  f4 xr[G];
  long long pl;
  reload_first<G>(m, w, xr, pl);
  stage_vectors(L, m);
  load_bn(L.sc0, L.sh0, L.mu0, L.is0, m.bn0);
  if (LAYERS == 3) {
    load_bn(L.sc1, L.sh1, L.mu1, L.is1, m.bn1);
    b1_coefficients(L, m);
  }
  __syncthreads();
  b2_loop<LAYERS, G>(m, L, w, xr, pl);
  store_partial_row(L, m.part_b2 + (long long)blockIdx.x * B2_ROW, B2_ROW);
}

// ---------------------------------------------------------------------------------------------------- FIN: the gradients
// workgroup j = output channel j of every layer.  grads: [dW0 100][dgamma0 10][dbeta0 10][dW1 100][dbias1 10][dgamma1 10]
// [dbeta1 10][dW2 100][dbias2 10]  (two layers: W1 is the last conv, dbias1 = sum dy1; three layers: dbias1 is 0 -- a bias in
// front of a BatchNorm -- and is left to the caller).
// `count` (<= 64) values per partial row at offsets offs[v], summed over nblk rows: 8 slices of rows, 8 independent loads per trip
__device__ void reduce64(Sh &L, const float *part, int nblk, int pitch, const int *offs, int count) {
  __shared__ double sl8[8][64];
  const int t = threadIdx.x, v = t & 63, s = t >> 6;
  double a = 0.0;
  if (v < count) {
    const float *src = part + offs[v];
    for (int b = s; b < nblk; b += 8 * 8) {
      double x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = b + 8 * u < nblk ? (double)src[(long long)(b + 8 * u) * pitch] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) a += x[u];
    }
  }
  sl8[s][v] = a;
  __syncthreads();
  if (t < 64) {
    double r = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) r += sl8[k][t];
    L.fin[t] = r;
  }
  __syncthreads();
}

template <int LAYERS>
__device__ void fin_body(const rs_umbrella_mfma &m, Sh &L, int j) {
  __shared__ int offs[64];
  __shared__ double keep[64];
  const int t = threadIdx.x;
  const double n = (double)m.rows;
  float *G = m.grads;
  if (LAYERS == 3) {
    // [0,10) dW2[j][:]  [10,20) S1[j][:]  [20,30) Sy[j][:]  [30,40) sa0[:]  40 db1[j]  41 dg1[j]  42 db2[j]
    if (t < 64) {
      int o = 0;
      if (t < 30) o = (t / 10) * TILE + j * CP + t % 10;
      else if (t < 40) o = 3 * TILE + 2 * CP + (t - 30);
      else if (t == 40) o = 3 * TILE + j;
      else if (t == 41) o = 3 * TILE + CP + j;
      else if (t == 42) o = 3 * TILE + 3 * CP + j;
      offs[t] = o;
    }
    __syncthreads();
    reduce64(L, m.part_b1, m.nblk_b1, B1_ROW, offs, 43);
    if (t < 64) keep[t] = L.fin[t];
    __syncthreads();
    if (t < C) {
      const double s1 = (double)m.bn1[j], db1 = keep[40], dg1 = keep[41];
      G[250 + j * C + t] = (float)keep[t];                                                             // dW2[j][t]
      G[120 + j * C + t] = (float)(s1 * ((keep[10 + t] - db1 / n * keep[30 + t]) - dg1 / n * keep[20 + t]));   // dW1[j][t]
      if (t == 0) { G[350 + j] = (float)keep[42]; G[230 + j] = (float)dg1; G[240 + j] = (float)db1; }
    }
    __syncthreads();
  }
  // [0,10) T1[j][:]  [10,20) dWlast[j][:]  20 db0[j]  21 dg0[j]  22 dblast[j]
  if (t < 64) {
    int o = 0;
    if (t < 20) o = (t / 10) * TILE + j * CP + t % 10;
    else if (t == 20) o = 2 * TILE + j;
    else if (t == 21) o = 2 * TILE + CP + j;
    else if (t == 22) o = 2 * TILE + 2 * CP + j;
    offs[t] = o;
  }
  __syncthreads();
  reduce64(L, m.part_b2, m.nblk_b2, B2_ROW, offs, 23);
  if (t < C) {
    const double *S = m.moments;
    const double s0 = (double)m.bn0[j], mu0 = (double)m.bn0[2 * C + j], is0 = (double)m.bn0[3 * C + j];
    const double db0 = L.fin[20], dg0 = L.fin[21];
    double wx = 0.0;                               // sum_rows (W0 x)[j] x[t] = sum_k W0[j][k] Sxx[k][t]
#pragma unroll
    for (int k = 0; k < C; ++k) wx += (double)m.w0[j * C + k] * S[k * CP + t];
    const double sx = S[10 * CP + t];
    const double gy = is0 * (wx + ((m.b0 ? (double)m.b0[j] : 0.0) - mu0) * sx);          // sum_rows yhat0[j] x[t]
    G[j * C + t] = (float)(s0 * ((L.fin[t] - db0 / n * sx) - dg0 / n * gy));            // dW0[j][t]
    if (t == 0) { G[100 + j] = (float)dg0; G[110 + j] = (float)db0; }
    if (LAYERS == 2) {
      G[120 + j * C + t] = (float)L.fin[10 + t];
      if (t == 0) G[220 + j] = (float)L.fin[22];
    }
  }
}


template <int LAYERS>
__global__ void __launch_bounds__(TH)
umb_fin_kernel(rs_umbrella_mfma m) {
  __shared__ Sh L;
  if (m.rows_dev) m.rows = min(m.rows, (long long)*m.rows_dev);
  fin_body<LAYERS>(m, L, blockIdx.x);
}

}  // namespace

extern "C" int rs_umbrella_moments(const float *x, long long rows, float *partial, int nblk, double *moments, void *stream) {
  RS_REQUIRE(x && partial && moments && rows > 0 && nblk > 0, "rs_umbrella_moments: null pointer / empty input");
  RS_REQUIRE(rows < (1LL << 40), "rs_umbrella_moments: rows out of range");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(umb_moments_kernel, dim3(nblk), dim3(TH), 0, st, x, rows, partial);
  hipLaunchKernelGGL(umb_moments_reduce_kernel, dim3(1), dim3(1024), 0, st, partial, nblk, moments);
  RS_CHECK_LAUNCH("rs_umbrella_moments");
  return RS_OK;
}

// fans of 8 (classification, group_size 8) and 9 (segmentation: k = group_size + 1 triangles): the tile's rows are register arrays
#define RS_UMB_G(KERNEL, ...) do { if (m->group == 8) hipLaunchKernelGGL((KERNEL<__VA_ARGS__ 8>), grid, block, 0, st, *m); \
                                   else hipLaunchKernelGGL((KERNEL<__VA_ARGS__ 9>), grid, block, 0, st, *m); } while (0)
#define RS_COMMA ,

extern "C" int rs_umbrella_mfma_pass(int pass, const rs_umbrella_mfma *m, int nblk, void *stream) {
  RS_REQUIRE(m && m->x && m->w0 && m->w1 && m->moments && m->bn0, "rs_umbrella_mfma_pass: null descriptor / input / weights / moments");
  RS_REQUIRE(m->layers == 2 || m->layers == 3, "rs_umbrella_mfma_pass: layers = %d (2 or 3)", m->layers);
  RS_REQUIRE(m->group == 8 || m->group == 9, "rs_umbrella_mfma_pass: fans of 8 or 9 rows (group = %d: use rs_umbrella_mlp_pass)", m->group);
  RS_REQUIRE(m->rows > 0 && m->rows % m->group == 0 && nblk > 0, "rs_umbrella_mfma_pass: bad size");
  RS_REQUIRE(((uintptr_t)m->x % 8) == 0 && (m->dout == nullptr || ((uintptr_t)m->dout % 8) == 0), "rs_umbrella_mfma_pass: rows must be 8-byte aligned");
  const bool l3 = m->layers == 3;
  if (l3) RS_REQUIRE(m->w2 && m->bn1, "rs_umbrella_mfma_pass: the three-layer MLP needs w2 and bn1");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(nblk), block(TH);
  switch (pass) {
    case RS_UMB_F1:
      RS_REQUIRE(l3 && m->stat && m->nblk_f1 == nblk, "rs_umbrella_mfma_pass: F1 is a three-layer pass with stat (nblk_f1 = nblk rows)");
      RS_UMB_G(umb_f1_kernel, );
      break;
    case RS_UMB_F2:
      RS_REQUIRE(m->out && (!l3 || (m->stat && m->nblk_f1 > 0)), "rs_umbrella_mfma_pass: F2 needs out (and the F1 partials)");
      if (l3) RS_UMB_G(umb_f2_kernel, 3 RS_COMMA);
      else RS_UMB_G(umb_f2_kernel, 2 RS_COMMA);
      break;
    case RS_UMB_B1:
      RS_REQUIRE(l3 && m->dout && m->part_b1 && m->nblk_b1 == nblk, "rs_umbrella_mfma_pass: B1 is a three-layer pass with dout and part_b1 (nblk_b1 = nblk rows)");
      RS_UMB_G(umb_b1_kernel, );
      break;
    case RS_UMB_B2:
      RS_REQUIRE(m->dout && m->part_b2 && m->nblk_b2 == nblk && (!l3 || (m->part_b1 && m->nblk_b1 > 0)),
                 "rs_umbrella_mfma_pass: B2 needs dout, part_b2 (nblk_b2 = nblk rows) (and the B1 partials)");
      if (l3) RS_UMB_G(umb_b2_kernel, 3 RS_COMMA);
      else RS_UMB_G(umb_b2_kernel, 2 RS_COMMA);
      break;
    case RS_UMB_FIN:
      RS_REQUIRE(m->grads && m->part_b2 && m->nblk_b2 > 0 && (!l3 || (m->part_b1 && m->nblk_b1 > 0)), "rs_umbrella_mfma_pass: FIN needs grads and the partials");
      if (l3) hipLaunchKernelGGL(umb_fin_kernel<3>, dim3(C), block, 0, st, *m);
      else hipLaunchKernelGGL(umb_fin_kernel<2>, dim3(C), block, 0, st, *m);
      break;
    default:
      RS_REQUIRE(false, "rs_umbrella_mfma_pass: pass %d out of range", pass);
  }
  RS_CHECK_LAUNCH("rs_umbrella_mfma_pass");
  return RS_OK;
}
