// umbrella_mlp.hip — the 10-channel shared MLP of UmbrellaSurfaceConstructor, fused (gfx950).
//
// Reference: self.mlps = Conv2d(10,10,bias=False)-BN-ReLU-Conv2d(10,10)-BN-ReLU-Conv2d(10,10) over
// (B, 10, 8, N), then a sum over the 8 fan triangles (classification/modules/repsurface_utils.py:266-274,
// 296-305): seven framework calls that each stream a (B,10,8,N) tensor.
// With 10 channels a 32x32 MFMA tile is 90 % padding and every intermediate is tiny, so this path does
// NOT go through the generic row-GEMM: one thread owns one row (one triangle's 10 features), keeps the
// whole layer chain in registers, and re-derives upstream activations from the input features instead
// of storing them (a layer is 100 FMAs; the input row is 40 bytes).  The only global synchronisation
// points are the ones BatchNorm imposes — batch statistics between layers — so forward is three passes
// over x (stats of y0, stats of y1, output) and backward three more (each produces one weight gradient
// and the BatchNorm-backward sums of the layer below).  Weights sit in LDS (k-major and natural copies,
// broadcast float4 reads); per-thread partial sums are reduced wave -> workgroup -> one fp64/fp32
// partial row per workgroup, summed in fixed order by the finalize kernels (deterministic).
// HBM traffic per pass: rows * 40 B (10.5 MB at B=32) — all six passes together move less than one
// activation tensor of the unfused path.
#include "rs_common.h"

namespace {

constexpr int UM_C = 10;        // channels (repsurf_channel)
constexpr int UM_CP = 12;       // padded row length in LDS (float4 aligned)
constexpr int UM_THREADS = 256;
// register budget of the row kernels: left to the compiler (the backward passes then take 256 VGPRs + AGPRs, one wave per SIMD);
// -DRS_UM_OCC=2 cuts it for two waves per SIMD: 20-34 spilled registers, 24-30 us per backward pass against 20-23
#ifdef RS_UM_OCC
#define RS_UM_BOUNDS __launch_bounds__(UM_THREADS, RS_UM_OCC)
#else
#define RS_UM_BOUNDS __launch_bounds__(UM_THREADS)
#endif

struct UmbWeights {             // LDS image
  float w0t[UM_C * UM_CP];      // k-major: w0t[k][j] = w0[j][k]   (forward)
  float w1t[UM_C * UM_CP];
  float w2t[UM_C * UM_CP];
  float w1n[UM_C * UM_CP];      // natural: w1n[j][k] = w1[j][k]   (data gradient)
  float w2n[UM_C * UM_CP];
  float b0[UM_CP], b1[UM_CP], b2[UM_CP];
  float bn0[4 * UM_CP], bn1[4 * UM_CP];    // scale, shift, mean, invstd
  float c0[3 * UM_CP], c1[3 * UM_CP];      // p, q, r of BatchNorm backward
};

__device__ void load_weights(UmbWeights &L, const rs_umbrella_mlp &m) {
  for (int e = threadIdx.x; e < UM_C * UM_CP; e += UM_THREADS) {
    const int a = e / UM_CP, b = e % UM_CP;
    const bool ok = b < UM_C;
    L.w0t[e] = ok ? m.w0[b * UM_C + a] : 0.f;
    L.w1t[e] = (ok && m.w1) ? m.w1[b * UM_C + a] : 0.f;
    L.w2t[e] = (ok && m.w2) ? m.w2[b * UM_C + a] : 0.f;
    L.w1n[e] = (ok && m.w1) ? m.w1[a * UM_C + b] : 0.f;
    L.w2n[e] = (ok && m.w2) ? m.w2[a * UM_C + b] : 0.f;
  }
  for (int e = threadIdx.x; e < UM_CP; e += UM_THREADS) {
    L.b0[e] = (e < UM_C && m.b0) ? m.b0[e] : 0.f;
    L.b1[e] = (e < UM_C && m.b1) ? m.b1[e] : 0.f;
    L.b2[e] = (e < UM_C && m.b2) ? m.b2[e] : 0.f;
  }
  for (int e = threadIdx.x; e < 4 * UM_CP; e += UM_THREADS) {
    const int v = e / UM_CP, c = e % UM_CP;
    L.bn0[e] = (c < UM_C && m.bn0) ? m.bn0[v * UM_C + c] : 0.f;
    L.bn1[e] = (c < UM_C && m.bn1) ? m.bn1[v * UM_C + c] : 0.f;
  }
  for (int e = threadIdx.x; e < 3 * UM_CP; e += UM_THREADS) {
    const int v = e / UM_CP, c = e % UM_CP;
    L.c0[e] = (c < UM_C && m.c0) ? m.c0[v * UM_C + c] : 0.f;
    L.c1[e] = (c < UM_C && m.c1) ? m.c1[v * UM_C + c] : 0.f;
  }
  __syncthreads();
}

// out[j] = bias[j] + sum_k in[k] * wt[k][j]     (wt k-major in LDS; all lanes read the same address)
__device__ __forceinline__ void matvec_t(const float *wt, const float *bias, const float (&in)[UM_C], float (&out)[UM_C]) {
#pragma unroll
  for (int j = 0; j < UM_C; ++j) out[j] = bias ? bias[j] : 0.f;
#pragma unroll
  for (int k = 0; k < UM_C; ++k) {
    const float4 a = *reinterpret_cast<const float4 *>(wt + k * UM_CP);
    const float4 b = *reinterpret_cast<const float4 *>(wt + k * UM_CP + 4);
    const float2 c = *reinterpret_cast<const float2 *>(wt + k * UM_CP + 8);
    const float w[UM_C] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y};
#pragma unroll
    for (int j = 0; j < UM_C; ++j) out[j] = fmaf(in[k], w[j], out[j]);
  }
}
// out[k] = sum_j in[j] * wn[j][k]               (data gradient dA = dY . W)
__device__ __forceinline__ void matvec_n(const float *wn, const float (&in)[UM_C], float (&out)[UM_C]) {
  matvec_t(wn, nullptr, in, out);                 // same access shape: rows of the natural copy
}
__device__ __forceinline__ void bn_relu(const float *bn, const float (&y)[UM_C], float (&a)[UM_C]) {
#pragma unroll
  for (int j = 0; j < UM_C; ++j) a[j] = fmaxf(fmaf(bn[j], y[j], bn[UM_CP + j]), 0.f);
}
__device__ __forceinline__ void load_row(const float *x, long long r, float (&v)[UM_C]) {
  const float2 *p = reinterpret_cast<const float2 *>(x + r * UM_C);   // rows are 40 B: 8-byte aligned
#pragma unroll
  for (int i = 0; i < UM_C / 2; ++i) { const float2 t = p[i]; v[2 * i] = t.x; v[2 * i + 1] = t.y; }
}

// workgroup reduction of NV per-thread values -> dst[blockIdx.x * NV + i] (T = float or double), through LDS.
// (Round 2 summed every value across the wave by DPP: ~29 instructions x 110 values = 3 200 instructions per wave for the weight-
// gradient tiles -- more than the 4 rows x 740 instructions of the pass itself.)  The lanes write their values to LDS as [value][lane]
// (pitch 68: conflict-free writes, 4-way = minimal on the 16-byte reads), four threads per value sum one wave's 64 lanes each
// (16 ds_read_b128), the four meet by two quad shuffles; RED_H values per round (55 of the 110 weight-gradient values: 60 KB).
// Order of the sum: lanes ascending within a wave, then the waves -- fixed, so the result is deterministic.
// (The statistics-only passes reduce 20 values and size the buffer for those: a 60 KB buffer tells the register allocator that
// two workgroups per CU is the most there can be, and it then takes 256 VGPRs where 60 do.)
constexpr int RED_PITCH = 68;
constexpr int red_floats(int h) { return (UM_THREADS / 64) * h * RED_PITCH; }

template <int NV, int RED_H, typename T>
__device__ void block_reduce_store(float (&v)[NV], float *red, T *dst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c0 = 0; c0 < NV; c0 += RED_H) {
    const int n = NV - c0 < RED_H ? NV - c0 : RED_H;
#pragma unroll
    for (int i = 0; i < RED_H; ++i)
      if (i < n) red[(wave * RED_H + i) * RED_PITCH + lane] = v[c0 + i];
    __syncthreads();
    const int i = threadIdx.x >> 2, w = threadIdx.x & 3;
    float s = 0.f;
    if (i < n) {
      const float4 *src = reinterpret_cast<const float4 *>(red + (w * RED_H + i) * RED_PITCH);
#pragma unroll
      for (int q = 0; q < 16; ++q) { const float4 x = src[q]; s += x.x; s += x.y; s += x.z; s += x.w; }
    }
    T t = (T)s;                                                    // the four waves of value i sit in one quad
    t += __shfl_xor(t, 1, 64);
    t += __shfl_xor(t, 2, 64);
    if (i < n && w == 0) dst[(long long)blockIdx.x * NV + c0 + i] = t;
    __syncthreads();
  }
}

// PASS 0: stats(y0)  1: stats(y1)  3: dW2,db2 + BN1-backward sums  4: dW1 + BN0-backward sums  5: dW0
// (The backward passes take 256 VGPRs + 20-34 AGPRs: ONE wave per SIMD, so a second workgroup on a CU simply waits for the first
//  -- 256 workgroups is the grid.  Cutting the budget for two waves spills and is slower: RS_UM_OCC.)
template <int PASS>
__global__ void RS_UM_BOUNDS
umbrella_mlp_rows_kernel(rs_umbrella_mlp m, double *__restrict__ stat_partial, float *__restrict__ dw_partial) {
  __shared__ UmbWeights L;
  constexpr int RH = PASS >= 3 ? 55 : 2 * UM_C;               // values per reduction round (block_reduce_store)
  __shared__ __attribute__((aligned(16))) float scratch[red_floats(RH)];
  load_weights(L, m);
  float st[2 * UM_C];
  float dw[UM_C * UM_C + UM_C];
#pragma unroll
  for (int i = 0; i < 2 * UM_C; ++i) st[i] = 0.f;
#pragma unroll
  for (int i = 0; i < UM_C * UM_C + UM_C; ++i) dw[i] = 0.f;

  for (long long r = (long long)blockIdx.x * UM_THREADS + threadIdx.x; r < m.rows; r += (long long)gridDim.x * UM_THREADS) {
    // The weights are loop-invariant LDS data: without this clobber the compiler hoists all ~500 of them
    // into registers and spills; re-reading them per row (broadcast ds_read_b128) is what we want.
    asm volatile("" ::: "memory");
    float x[UM_C], y0[UM_C];
    load_row(m.x, r, x);
    matvec_t(L.w0t, nullptr, x, y0);
    if (PASS == 0) {
#pragma unroll
      for (int j = 0; j < UM_C; ++j) { st[j] += y0[j]; st[UM_C + j] = fmaf(y0[j], y0[j], st[UM_C + j]); }
      continue;
    }
    float a0[UM_C], y1[UM_C];
    bn_relu(L.bn0, y0, a0);
    matvec_t(L.w1t, L.b1, a0, y1);
    if (PASS == 1) {
#pragma unroll
      for (int j = 0; j < UM_C; ++j) { st[j] += y1[j]; st[UM_C + j] = fmaf(y1[j], y1[j], st[UM_C + j]); }
      continue;
    }
    // ---- backward passes: dY2 = dout[point]  (sum over the fan; `scale` folds the 1/group of 'avg')
    float a1[UM_C], dy2[UM_C], da1[UM_C], dz1[UM_C];
    bn_relu(L.bn1, y1, a1);
    load_row(m.dout, (long long)((unsigned)r / (unsigned)m.group), dy2);
    matvec_n(L.w2n, dy2, da1);
#pragma unroll
    for (int k = 0; k < UM_C; ++k) dz1[k] = a1[k] > 0.f ? da1[k] : 0.f;
    if (PASS == 3) {
#pragma unroll
      for (int j = 0; j < UM_C; ++j) {
#pragma unroll
        for (int k = 0; k < UM_C; ++k) dw[j * UM_C + k] = fmaf(dy2[j], a1[k], dw[j * UM_C + k]);
        dw[UM_C * UM_C + j] += dy2[j];                                          // bias of the last conv
      }
#pragma unroll
      for (int k = 0; k < UM_C; ++k) {
        st[k] += dz1[k];
        st[UM_C + k] = fmaf(dz1[k], (y1[k] - L.bn1[2 * UM_CP + k]) * L.bn1[3 * UM_CP + k], st[UM_C + k]);
      }
      continue;
    }
    float dy1[UM_C], da0[UM_C], dz0[UM_C];
#pragma unroll
    for (int j = 0; j < UM_C; ++j) dy1[j] = fmaf(L.c1[j], dz1[j], fmaf(L.c1[UM_CP + j], y1[j], L.c1[2 * UM_CP + j]));
    matvec_n(L.w1n, dy1, da0);
#pragma unroll
    for (int k = 0; k < UM_C; ++k) dz0[k] = a0[k] > 0.f ? da0[k] : 0.f;
    if (PASS == 4) {
#pragma unroll
      for (int j = 0; j < UM_C; ++j)
#pragma unroll
        for (int k = 0; k < UM_C; ++k) dw[j * UM_C + k] = fmaf(dy1[j], a0[k], dw[j * UM_C + k]);
#pragma unroll
      for (int k = 0; k < UM_C; ++k) {
        st[k] += dz0[k];
        st[UM_C + k] = fmaf(dz0[k], (y0[k] - L.bn0[2 * UM_CP + k]) * L.bn0[3 * UM_CP + k], st[UM_C + k]);
      }
      continue;
    }
    // PASS 5
    float dy0[UM_C];
#pragma unroll
    for (int j = 0; j < UM_C; ++j) dy0[j] = fmaf(L.c0[j], dz0[j], fmaf(L.c0[UM_CP + j], y0[j], L.c0[2 * UM_CP + j]));
#pragma unroll
    for (int j = 0; j < UM_C; ++j)
#pragma unroll
      for (int k = 0; k < UM_C; ++k) dw[j * UM_C + k] = fmaf(dy0[j], x[k], dw[j * UM_C + k]);
  }
  if (PASS != 5) block_reduce_store<2 * UM_C, RH, double>(st, scratch, stat_partial);
  if (PASS >= 3) block_reduce_store<UM_C * UM_C + UM_C, RH, float>(dw, scratch, dw_partial);
}

// ---- two-layer variant: the segmentation constructor's mlps = Conv1d(10,10)-BN-ReLU-Conv1d(10,10), summed over the fan
// (segmentation/modules/repsurface_utils.py:298-303,323-327).  PASS 0: stats(y0), y0 = W0 x + b0;  4: {dW1, db1} + BN0-backward
// sums (dy1 = dout[point]);  5: dW0 (needs c0).  Same register-resident scheme, one stage shorter.
template <int PASS>
__global__ void RS_UM_BOUNDS
umbrella_mlp2_rows_kernel(rs_umbrella_mlp m, double *__restrict__ stat_partial, float *__restrict__ dw_partial) {
  __shared__ UmbWeights L;
  constexpr int RH = PASS >= 3 ? 55 : 2 * UM_C;               // values per reduction round (block_reduce_store)
  __shared__ __attribute__((aligned(16))) float scratch[red_floats(RH)];
  load_weights(L, m);
  float st[2 * UM_C];
  float dw[UM_C * UM_C + UM_C];
#pragma unroll
  for (int i = 0; i < 2 * UM_C; ++i) st[i] = 0.f;
#pragma unroll
  for (int i = 0; i < UM_C * UM_C + UM_C; ++i) dw[i] = 0.f;
  for (long long r = (long long)blockIdx.x * UM_THREADS + threadIdx.x; r < m.rows; r += (long long)gridDim.x * UM_THREADS) {
    asm volatile("" ::: "memory");                 // (weights stay in LDS: see umbrella_mlp_rows_kernel)
    float x[UM_C], y0[UM_C];
    load_row(m.x, r, x);
    matvec_t(L.w0t, L.b0, x, y0);
    if (PASS == 0) {
#pragma unroll
      for (int j = 0; j < UM_C; ++j) { st[j] += y0[j]; st[UM_C + j] = fmaf(y0[j], y0[j], st[UM_C + j]); }
      continue;
    }
    float a0[UM_C], dy1[UM_C], da0[UM_C], dz0[UM_C];
    bn_relu(L.bn0, y0, a0);
    load_row(m.dout, (long long)((unsigned)r / (unsigned)m.group), dy1);
    matvec_n(L.w1n, dy1, da0);
#pragma unroll
    for (int k = 0; k < UM_C; ++k) dz0[k] = a0[k] > 0.f ? da0[k] : 0.f;
    if (PASS == 4) {
#pragma unroll
      for (int j = 0; j < UM_C; ++j) {
#pragma unroll
        for (int k = 0; k < UM_C; ++k) dw[j * UM_C + k] = fmaf(dy1[j], a0[k], dw[j * UM_C + k]);
        dw[UM_C * UM_C + j] += dy1[j];                                          // bias of the last conv
      }
#pragma unroll
      for (int k = 0; k < UM_C; ++k) {
        st[k] += dz0[k];
        st[UM_C + k] = fmaf(dz0[k], (y0[k] - L.bn0[2 * UM_CP + k]) * L.bn0[3 * UM_CP + k], st[UM_C + k]);
      }
      continue;
    }
    // PASS 5
    float dy0[UM_C];
#pragma unroll
    for (int j = 0; j < UM_C; ++j) dy0[j] = fmaf(L.c0[j], dz0[j], fmaf(L.c0[UM_CP + j], y0[j], L.c0[2 * UM_CP + j]));
#pragma unroll
    for (int j = 0; j < UM_C; ++j)
#pragma unroll
      for (int k = 0; k < UM_C; ++k) dw[j * UM_C + k] = fmaf(dy0[j], x[k], dw[j * UM_C + k]);
  }
  if (PASS != 5) block_reduce_store<2 * UM_C, RH, double>(st, scratch, stat_partial);
  if (PASS >= 4) block_reduce_store<UM_C * UM_C + UM_C, RH, float>(dw, scratch, dw_partial);
}

// two-layer PASS 2: out[p] = scale * sum_g y1[p*group + g]
__global__ void __launch_bounds__(UM_THREADS)
umbrella_mlp2_out_kernel(rs_umbrella_mlp m, float scale, float *__restrict__ out) {
  __shared__ UmbWeights L;
  load_weights(L, m);
  const long long points = m.rows / m.group;
  for (long long p = (long long)blockIdx.x * UM_THREADS + threadIdx.x; p < points; p += (long long)gridDim.x * UM_THREADS) {
    float acc[UM_C];
#pragma unroll
    for (int j = 0; j < UM_C; ++j) acc[j] = 0.f;
    for (int g = 0; g < m.group; ++g) {
      asm volatile("" ::: "memory");
      float x[UM_C], y0[UM_C], a0[UM_C], y1[UM_C];
      load_row(m.x, p * m.group + g, x);
      matvec_t(L.w0t, L.b0, x, y0);
      bn_relu(L.bn0, y0, a0);
      matvec_t(L.w1t, L.b1, a0, y1);
#pragma unroll
      for (int j = 0; j < UM_C; ++j) acc[j] += y1[j];
    }
    float2 *o = reinterpret_cast<float2 *>(out + p * UM_C);
#pragma unroll
    for (int i = 0; i < UM_C / 2; ++i) o[i] = make_float2(acc[2 * i] * scale, acc[2 * i + 1] * scale);
  }
}

// PASS 2: out[p] = scale * sum_g y2[p*group + g]     (one thread per point)
__global__ void __launch_bounds__(UM_THREADS)
umbrella_mlp_out_kernel(rs_umbrella_mlp m, float scale, float *__restrict__ out) {
  __shared__ UmbWeights L;
  load_weights(L, m);
  const long long points = m.rows / m.group;
  for (long long p = (long long)blockIdx.x * UM_THREADS + threadIdx.x; p < points; p += (long long)gridDim.x * UM_THREADS) {
    float acc[UM_C];
#pragma unroll
    for (int j = 0; j < UM_C; ++j) acc[j] = 0.f;
    for (int g = 0; g < m.group; ++g) {
      asm volatile("" ::: "memory");     // keep the LDS weight reads inside the loop (see rows kernel)
      float x[UM_C], y0[UM_C], a0[UM_C], y1[UM_C], a1[UM_C], y2[UM_C];
      load_row(m.x, p * m.group + g, x);
      matvec_t(L.w0t, nullptr, x, y0);
      bn_relu(L.bn0, y0, a0);
      matvec_t(L.w1t, L.b1, a0, y1);
      bn_relu(L.bn1, y1, a1);
      matvec_t(L.w2t, L.b2, a1, y2);
#pragma unroll
      for (int j = 0; j < UM_C; ++j) acc[j] += y2[j];
    }
    float2 *o = reinterpret_cast<float2 *>(out + p * UM_C);
#pragma unroll
    for (int i = 0; i < UM_C / 2; ++i) o[i] = make_float2(acc[2 * i] * scale, acc[2 * i + 1] * scale);
  }
}

// PASS 2 for fans of 8 (the classification constructor): one thread per ROW.  With one thread per point only 32 768 threads exist
// at B = 32 (half a wave per SIMD) and each walks 8 rows serially: 20 us.  Here the 8 rows of a point are 8 adjacent lanes, their
// y2 vectors meet by three DPP steps (xor 1, xor 2, half-row mirror) and lane 0 of the octet stores the point.
__global__ void __launch_bounds__(UM_THREADS)
umbrella_mlp_out8_kernel(rs_umbrella_mlp m, float scale, float *__restrict__ out) {
  __shared__ UmbWeights L;
  load_weights(L, m);
  const long long step = (long long)gridDim.x * UM_THREADS;
  const long long rows_up = (m.rows + 63) & ~63LL;                 // whole waves: every lane takes part in the DPP steps
  for (long long r = (long long)blockIdx.x * UM_THREADS + threadIdx.x; r < rows_up; r += step) {
    asm volatile("" ::: "memory");                                // keep the LDS weight reads inside the loop (see rows kernel)
    const bool ok = r < m.rows;
    float x[UM_C], y0[UM_C], a0[UM_C], y1[UM_C], a1[UM_C], y2[UM_C];
    load_row(m.x, ok ? r : m.rows - 1, x);
    matvec_t(L.w0t, nullptr, x, y0);
    bn_relu(L.bn0, y0, a0);
    matvec_t(L.w1t, L.b1, a0, y1);
    bn_relu(L.bn1, y1, a1);
    matvec_t(L.w2t, L.b2, a1, y2);
#pragma unroll
    for (int j = 0; j < UM_C; ++j) {
      unsigned v = __float_as_uint(ok ? y2[j] : 0.f);
      v = __float_as_uint(__uint_as_float(v) + __uint_as_float(rs_dpp<RS_DPP_QUAD_XOR1>(v)));
      v = __float_as_uint(__uint_as_float(v) + __uint_as_float(rs_dpp<RS_DPP_QUAD_XOR2>(v)));
      v = __float_as_uint(__uint_as_float(v) + __uint_as_float(rs_dpp<RS_DPP_ROW_HALF_MIRROR>(v)));
      y2[j] = __uint_as_float(v) * scale;
    }
    if (ok && (threadIdx.x & 7) == 0) {
      float2 *o = reinterpret_cast<float2 *>(out + (r >> 3) * UM_C);
#pragma unroll
      for (int i = 0; i < UM_C / 2; ++i) o[i] = make_float2(y2[2 * i], y2[2 * i + 1]);
    }
  }
}

}  // namespace

extern "C" int rs_umbrella_mlp_pass(int pass, const rs_umbrella_mlp *m, float out_scale, float *out,
                                    double *stat_partial, float *dw_partial, int nblk, void *stream) {
  RS_REQUIRE(m && m->x && m->w0, "rs_umbrella_mlp_pass: null descriptor / input / weights");
  RS_REQUIRE(pass >= 0 && pass <= 5, "rs_umbrella_mlp_pass: pass %d out of range", pass);
  RS_REQUIRE(m->rows >= 0 && m->rows < 2147483647LL && m->group > 0 && nblk > 0, "rs_umbrella_mlp_pass: bad size");
  if (m->rows == 0) return RS_OK;
  RS_REQUIRE(((uintptr_t)m->x % 8) == 0 && (m->dout == nullptr || ((uintptr_t)m->dout % 8) == 0),
             "rs_umbrella_mlp_pass: rows must be 8-byte aligned");
  RS_REQUIRE(m->layers == 0 || m->layers == 2 || m->layers == 3, "rs_umbrella_mlp_pass: layers = %d (2 or 3)", m->layers);
  if (m->layers == 2) {
    RS_REQUIRE(pass == 0 || pass == 2 || pass == 4 || pass == 5, "rs_umbrella_mlp_pass: the two-layer MLP has passes 0, 2, 4, 5 (not %d)", pass);
    if (pass >= 2) RS_REQUIRE(m->bn0 && m->w1, "rs_umbrella_mlp_pass: pass %d needs bn0 and w1", pass);
    if (pass == 2) RS_REQUIRE(out, "rs_umbrella_mlp_pass: pass 2 needs an output");
    if (pass >= 4) RS_REQUIRE(m->dout && dw_partial, "rs_umbrella_mlp_pass: backward passes need dout and dw_partial");
    if (pass == 5) RS_REQUIRE(m->c0, "rs_umbrella_mlp_pass: pass 5 needs the BN0 backward coefficients");
    if (pass == 0 || pass == 4) RS_REQUIRE(stat_partial, "rs_umbrella_mlp_pass: pass %d needs stat_partial", pass);
    hipStream_t st2 = (hipStream_t)stream;
    const dim3 grid2(nblk), block2(UM_THREADS);
    switch (pass) {
      case 0: hipLaunchKernelGGL(umbrella_mlp2_rows_kernel<0>, grid2, block2, 0, st2, *m, stat_partial, dw_partial); break;
      case 2: hipLaunchKernelGGL(umbrella_mlp2_out_kernel, grid2, block2, 0, st2, *m, out_scale, out); break;
      case 4: hipLaunchKernelGGL(umbrella_mlp2_rows_kernel<4>, grid2, block2, 0, st2, *m, stat_partial, dw_partial); break;
      default: hipLaunchKernelGGL(umbrella_mlp2_rows_kernel<5>, grid2, block2, 0, st2, *m, stat_partial, dw_partial); break;
    }
    RS_CHECK_LAUNCH("rs_umbrella_mlp_pass");
    return RS_OK;
  }
  if (pass >= 1) RS_REQUIRE(m->bn0 && m->w1, "rs_umbrella_mlp_pass: pass %d needs bn0 and w1", pass);
  if (pass >= 2) RS_REQUIRE(m->bn1 && m->w2, "rs_umbrella_mlp_pass: pass %d needs bn1 and w2", pass);
  if (pass == 2) RS_REQUIRE(out, "rs_umbrella_mlp_pass: pass 2 needs an output");
  if (pass >= 3) RS_REQUIRE(m->dout && dw_partial, "rs_umbrella_mlp_pass: backward passes need dout and dw_partial");
  if (pass == 4 || pass == 5) RS_REQUIRE(m->c1, "rs_umbrella_mlp_pass: pass %d needs the BN1 backward coefficients", pass);
  if (pass == 5) RS_REQUIRE(m->c0, "rs_umbrella_mlp_pass: pass 5 needs the BN0 backward coefficients");
  if (pass != 2 && pass != 5) RS_REQUIRE(stat_partial, "rs_umbrella_mlp_pass: pass %d needs stat_partial", pass);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(nblk), block(UM_THREADS);
  switch (pass) {
    case 0: hipLaunchKernelGGL(umbrella_mlp_rows_kernel<0>, grid, block, 0, st, *m, stat_partial, dw_partial); break;
    case 1: hipLaunchKernelGGL(umbrella_mlp_rows_kernel<1>, grid, block, 0, st, *m, stat_partial, dw_partial); break;
    case 2:
      if (m->group == 8 && m->rows % 8 == 0) hipLaunchKernelGGL(umbrella_mlp_out8_kernel, grid, block, 0, st, *m, out_scale, out);
      else hipLaunchKernelGGL(umbrella_mlp_out_kernel, grid, block, 0, st, *m, out_scale, out);
      break;
    case 3: hipLaunchKernelGGL(umbrella_mlp_rows_kernel<3>, grid, block, 0, st, *m, stat_partial, dw_partial); break;
    case 4: hipLaunchKernelGGL(umbrella_mlp_rows_kernel<4>, grid, block, 0, st, *m, stat_partial, dw_partial); break;
    default: hipLaunchKernelGGL(umbrella_mlp_rows_kernel<5>, grid, block, 0, st, *m, stat_partial, dw_partial); break;
  }
  RS_CHECK_LAUNCH("rs_umbrella_mlp_pass");
  return RS_OK;
}
