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
#include <stdlib.h>
#include <math.h>
#include <type_traits>

// The library builds this file as three translation units that compile in parallel (Makefile):
//   RS_MLP_TU = 0  every entry point except the two *_bf16 ones (fp32 kernels);
//   RS_MLP_TU = 1  only rs_mlp_gemm_rows_bf16 and rs_mlp_wgrad_bf16 (bf16 MFMA operands, fp32 tensors in HBM), plus the fp32
//                  instances they fall back to;
//   RS_MLP_TU = 3  the same two with bf16 ACTIVATION STORAGE (rs_sb_*: reached through the entry points of unit 1 when a
//                  call marks tensors as bf16).  Which tensor of a launch is bf16 is a function of the operand mode, fixed at
//                  compile time (sb_a / sb_b below): run-time flags in the load path cost 10 % of the bf16 step time, the
//                  halved bytes win back 15 %;
//   RS_MLP_TU = 4  (round 4; RS_GEMM_SPLIT3=0 switches it off) the fp32 product on the bf16 matrix pipe: every
//                  operand value is committed to LDS as THREE bf16 parts h + m + l (each the nearest-even bf16 of what the parts
//                  before it left: 24 significant bits together) and a product is six v_mfma_f32_32x32x16_bf16 -- hh, hm, mh, hl,
//                  lh, mm -- with fp32 accumulation: error below the fp32 MFMA's own (tools/probes/bf16_split_accuracy.py: 5.7e-7
//                  max against 1.3e-6 on a K = 512 layer; the usual two-part / three-product form is at 1.7e-5, outside the 1e-5
//                  bound), 192 matrix-pipe cycles per 16 k against 512.  rs_sp_* : reached through the fp32 entry points of unit 0
//                  for the launches that would take the tiled MFMA kernels;
//   RS_MLP_TU = 2  (default, experiment builds of tools/build_exp.sh) units 0 + 1 in one.
#ifndef RS_MLP_TU
#define RS_MLP_TU 2
#endif
#define RS_TU_HAS_BF16 (RS_MLP_TU != 0)                         /* BF = true instances */
#define RS_TU_BF16_ONLY (RS_MLP_TU == 1 || RS_MLP_TU == 3 || RS_MLP_TU == 4)      /* no fp32 entry points, no pooling / packing / BatchNorm kernels */

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// two fp32 -> one dword of two bf16 (round to nearest even: v_cvt_pk_bf16_f32), lo in bits [15:0]
__device__ __forceinline__ float pack_bf16(float lo, float hi) {
  const f32x2 f = {lo, hi};
  return __builtin_bit_cast(float, __builtin_convertvector(f, bf16x2));
}

// unit 4: x0, x1 -> the three dwords (bf16 pairs) of their parts.  The subtractions are exact (a bf16 of x shares x's exponent range),
// so h + m + l differs from x by at most the last rounding: 2^-8 of 2^-16 |x|.
constexpr bool RS_SPLIT = RS_MLP_TU == 4;
__device__ __forceinline__ void split_bf16(float x0, float x1, float (&d)[3]) {
  // (the two residual subtractions of a pair as ONE v_pk_add_f32 each -- exact either way: 9 instead of 11 VALU per pair)
  d[0] = pack_bf16(x0, x1);
  unsigned u = __float_as_uint(d[0]);
  const f32x2 x = {x0, x1};
  const f32x2 h = {__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
  f32x2 r = x - h;
  d[1] = pack_bf16(r[0], r[1]);
  u = __float_as_uint(d[1]);
  const f32x2 m = {__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
  r = r - m;
  d[2] = pack_bf16(r[0], r[1]);
}

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
// Internal (template arguments of the tiled kernels only, round 6): RS_OP_POOLED with the group addressing fixed at compile time.  In the
// instance that reads it from the descriptor, the branch `o.grp ? load (grp, slot) : r / ns` joins a path with pending index loads and a
// path with none in front of the operand loads; the compiler's s_waitcnt pass must assume the loads at the join, so the DENSE path too
// waited for every outstanding load -- the chunk's weight transfers and the first operand vector -- in the middle of its prefetch, twice per
// chunk (seen in the ISA; the pooled data gradient of the group_all stage is the step's dominant launch).
constexpr int OPM_POOLED_DENSE = 6, OPM_POOLED_RAGGED = 7;
__host__ __device__ constexpr int opm_base(int m) { return (m == OPM_POOLED_DENSE || m == OPM_POOLED_RAGGED) ? (int)OPM_POOLED : m; }

// ---- vectorised operand access ------------------------------------------------------------------
// A thread always handles V consecutive columns (V = 4, 2 or 1, chosen by the launcher from the
// alignment of every pointer / leading dimension involved).  Loading is split in two so that
// the global loads of the NEXT tile are in flight while the MFMAs of the current one run:
//   op_load   issues the loads and keeps raw values in registers,
//   op_finish applies the operand formula when the values are written to LDS.
template <int V> struct VecT;
template <> struct VecT<4> { typedef float4 F; typedef int4 I; };
template <> struct VecT<2> { typedef float2 F; typedef int2 I; };
template <> struct VecT<1> { typedef float F; typedef int I; };

template <int V> __device__ __forceinline__ void ldv(const float *p, float (&d)[V]) {
  typename VecT<V>::F v = *reinterpret_cast<const typename VecT<V>::F *>(p);
  const float *f = reinterpret_cast<const float *>(&v);
#pragma unroll
  for (int i = 0; i < V; ++i) d[i] = f[i];
}
// bf16 activation storage: only translation unit 3 carries the code; in the others every test below is a compile-time false
// and the kernels are what they were.
constexpr bool RS_STORE_BF16 = RS_MLP_TU == 3;
// Storage roles (unit 3).  The bf16 tensors are the pre-BatchNorm conv outputs y, and a launch meets them in fixed places:
//   operand a: y under RELU1 / RELU2 (the activation rebuilt from y); fp32 under ID (stack input), AFF2 (masked gradient dz),
//              POOLED (pooled gradient) and BCAST;
//   operand b: always a y (RELU2's second branch, the y of AFF2 / POOLED's BatchNorm-backward term);
//   mask tensors my1 / my2: always y;   output: y (bf16) behind a forward operand (ID / RELU1 / RELU2), a gradient (fp32)
//   behind a backward one.  The host checks a call's flags against this table (sb_check_*).
__host__ __device__ constexpr bool sb_a(int mode) { return RS_STORE_BF16 && (mode == RS_OP_RELU1 || mode == RS_OP_RELU2); }
__host__ __device__ constexpr bool sb_b(int mode) { return RS_STORE_BF16 && (mode == RS_OP_RELU2 || mode == RS_OP_AFF2 || mode == RS_OP_POOLED); }
__host__ __device__ constexpr bool sb_out(int mode) { return RS_STORE_BF16 && mode <= RS_OP_RELU2; }
constexpr bool SB_MASK = RS_STORE_BF16;

// V consecutive elements at index e of an fp32 or (bf) bf16 tensor behind a float* base.  A bf16 tensor is loaded as RAW
// BITS into the first V/2 registers (V = 1: the zero-extended half word) and widened by bf16_expand where the values are
// used: a conversion right behind the load would make every prefetch wait for its own data.
template <int V> __device__ __forceinline__ void ldx(const float *base, long long e, bool bf, float (&d)[V]) {
  if constexpr (RS_STORE_BF16) {
    // ONE address for both element sizes (a second per-lane 64-bit address per load cost 10-30 VGPRs in the pipelined kernels)
    const char *p = reinterpret_cast<const char *>(base) + (e << (bf ? 1 : 2));
    if (bf) {
      if constexpr (V == 4) { const uint2 u = *reinterpret_cast<const uint2 *>(p); d[0] = __uint_as_float(u.x); d[1] = __uint_as_float(u.y); }
      else if constexpr (V == 2) d[0] = __uint_as_float(*reinterpret_cast<const unsigned *>(p));
      else d[0] = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short *>(p));
    } else {
      ldv<V>(reinterpret_cast<const float *>(p), d);
    }
  } else {
    ldv<V>(base + e, d);
  }
}
// The same from a WAVE-UNIFORM base pointer and an unsigned 32-bit element offset: the address is an SGPR pair + a 32-bit VGPR byte
// offset (the `saddr` form of global_load), where `base + (long long) e` costs a sign extension and a 64-bit add per load.
// Offsets are tile-relative (< 2^30 elements).
template <int V> __device__ __forceinline__ void ldxu(const float *ubase, unsigned off, bool bf, float (&d)[V]) {
  if constexpr (RS_STORE_BF16) {
    const char *p = reinterpret_cast<const char *>(ubase) + (bf ? off << 1 : off << 2);
    if (bf) {
      if constexpr (V == 4) { const uint2 u = *reinterpret_cast<const uint2 *>(p); d[0] = __uint_as_float(u.x); d[1] = __uint_as_float(u.y); }
      else if constexpr (V == 2) d[0] = __uint_as_float(*reinterpret_cast<const unsigned *>(p));
      else d[0] = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short *>(p));
    } else {
      ldv<V>(reinterpret_cast<const float *>(p), d);
    }
  } else {
    ldv<V>(reinterpret_cast<const float *>(reinterpret_cast<const char *>(ubase) + (off << 2)), d);
  }
}
// raw bits of ldx -> fp32 values, in place (exact: a bf16 is the upper half of an fp32)
template <int V> __device__ __forceinline__ void bf16_expand(float (&d)[V], bool bf) {
  if (RS_STORE_BF16 && bf) {
    if constexpr (V == 4) {
      const unsigned u0 = __float_as_uint(d[0]), u1 = __float_as_uint(d[1]);
      d[0] = __uint_as_float(u0 << 16); d[1] = __uint_as_float(u0 & 0xffff0000u);
      d[2] = __uint_as_float(u1 << 16); d[3] = __uint_as_float(u1 & 0xffff0000u);
    } else if constexpr (V == 2) {
      const unsigned u0 = __float_as_uint(d[0]);
      d[0] = __uint_as_float(u0 << 16); d[1] = __uint_as_float(u0 & 0xffff0000u);
    } else {
      d[0] = __uint_as_float(__float_as_uint(d[0]) << 16);
    }
  }
}
__device__ __forceinline__ float ld1x(const float *base, long long e, int bf) {
  if (RS_STORE_BF16 && bf) return __uint_as_float((unsigned)reinterpret_cast<const unsigned short *>(base)[e] << 16);
  return base[e];
}
// base + e elements of an fp32 or bf16 tensor, still typed float* (the accessors above take element offsets from it)
template <typename T> __device__ __forceinline__ T *tile_base(T *base, long long e, int bf) {
  using B = typename std::conditional<std::is_const<T>::value, const char, char>::type;
  return reinterpret_cast<T *>(reinterpret_cast<B *>(base) + e * ((RS_STORE_BF16 && bf) ? 2 : 4));
}
// fp32 -> bf16 bits, round to nearest even (v_cvt_pk_bf16_f32 on a pair with itself)
__device__ __forceinline__ unsigned short bf16_bits(float v) {
  return (unsigned short)(__float_as_uint(pack_bf16(v, v)) & 0xffffu);
}
__device__ __forceinline__ float bf16_round(float v) { return __uint_as_float((unsigned)bf16_bits(v) << 16); }

template <int V> __device__ __forceinline__ void ldvi(const int *p, int (&d)[V]) {
  typename VecT<V>::I v = *reinterpret_cast<const typename VecT<V>::I *>(p);
  const int *f = reinterpret_cast<const int *>(&v);
#pragma unroll
  for (int i = 0; i < V; ++i) d[i] = f[i];
}

template <int V> struct RawVec { float a[V]; float b[V]; int g[V]; float m; int k; };
template <int V> struct ColCoef { float s1[V], t1[V], s2[V], t2[V]; };

// MODE >= 0 fixes the operand mode at compile time (dead paths and their registers vanish);
// MODE < 0 reads it from the descriptor (generic fallback).
template <int V, int MODE>
__device__ __forceinline__ void op_coef(const RowOperand &o, int c, bool ok, ColCoef<V> &k) {
  const int mode = MODE >= 0 ? opm_base(MODE) : o.mode;
#pragma unroll
  for (int i = 0; i < V; ++i) { k.s1[i] = 0.f; k.t1[i] = 0.f; k.s2[i] = 0.f; k.t2[i] = 0.f; }
  if (!ok || mode == OPM_ID || mode == OPM_BCAST) return;
  ldv<V>(o.s1 + c, k.s1);
  ldv<V>(o.t1 + c, k.t1);
  if (mode != OPM_RELU1) ldv<V>(o.s2 + c, k.s2);
  if (mode == OPM_RELU2) ldv<V>(o.t2 + c, k.t2);
}

// Rows are addressed as (wave-uniform base row r0, small local row rl): the 64-bit part of every
// address stays in SGPRs and the per-lane offset is 32-bit.
// What a row of the operand carries besides its values -- group, slot inside the group, multiplicity: functions of the ROW only.  The
// row GEMM keeps a tile's rows per thread over the whole K loop, so it fetches them once per tile (op_row_meta, round 6) instead of with
// every chunk: for ragged groups the (group, slot) pair is an index the chunk's gathers have to wait for -- a dependent global round trip
// per operand vector and chunk in front of the prefetch.
struct RowMeta { unsigned g; int k; float m; };
template <int MODE>
__device__ __forceinline__ void op_row_meta(const RowOperand &o, long long r0, int rl, RowMeta &rm) {
  const int mode = MODE >= 0 ? opm_base(MODE) : o.mode;
  rm.g = 0u; rm.k = 0; rm.m = 1.f;
  const unsigned url = (unsigned)rl;
  if (mode == OPM_POOLED) {
    if (MODE == OPM_POOLED_RAGGED || (MODE != OPM_POOLED_DENSE && o.grp)) { rm.g = (unsigned)(o.grp + r0)[url]; rm.k = (o.slot + r0)[url]; }
    else {
      const unsigned ns = (unsigned)o.ns, r = (unsigned)(r0 + rl);
      rm.g = (ns & (ns - 1)) == 0 ? r >> (31 - __clz((int)ns)) : r / ns;
      rm.k = (int)(r - rm.g * ns);
    }
  }
  if ((mode == OPM_POOLED || mode == OPM_AFF2) && o.mult) rm.m = (o.mult + r0)[url];
}

template <int V, int MODE>
__device__ __forceinline__ void op_load(const RowOperand &o, long long r0, int rl, int c, bool ok, RawVec<V> &raw, const RowMeta *rm = nullptr) {
  const int mode = MODE >= 0 ? opm_base(MODE) : o.mode;
#pragma unroll
  for (int i = 0; i < V; ++i) { raw.a[i] = 0.f; raw.b[i] = 0.f; raw.g[i] = -1; }
  raw.m = 1.f; raw.k = 0;
  if (!ok) return;
  // (r0 is wave-uniform in every caller: the row part of an address stays in SGPRs, the lane part is an unsigned 32-bit offset)
  const unsigned offa = (unsigned)(rl * (int)o.lda + c), offb = (unsigned)(rl * (int)o.ldb + c), url = (unsigned)rl;
  const float *ua = tile_base(o.a, r0 * o.lda, sb_a(mode)), *ub = o.b ? tile_base(o.b, r0 * o.ldb, sb_b(mode)) : nullptr;
  switch (mode) {
    case OPM_ID: ldxu<V>(ua, offa, sb_a(OPM_ID), raw.a); break;
    case OPM_RELU1: ldxu<V>(ua, offa, sb_a(OPM_RELU1), raw.a); break;
    case OPM_RELU2: ldxu<V>(ua, offa, sb_a(OPM_RELU2), raw.a); ldxu<V>(ub, offb, sb_b(OPM_RELU2), raw.b); break;
    case OPM_AFF2:
      ldxu<V>(ua, offa, sb_a(OPM_AFF2), raw.a); ldxu<V>(ub, offb, sb_b(OPM_AFF2), raw.b);
      if (rm) raw.m = rm->m;
      else if (o.mult) raw.m = (o.mult + r0)[url];
      break;
    case OPM_POOLED: {
      unsigned g;
      if (rm) { g = rm->g; raw.k = rm->k; }                     // (the caller fetched the row's group / slot / multiplicity once per tile)
      else if (MODE == OPM_POOLED_RAGGED || (MODE != OPM_POOLED_DENSE && o.grp)) { g = (unsigned)(o.grp + r0)[url]; raw.k = (o.slot + r0)[url]; }       // compacted (ragged) groups
      else {
        // dense groups: row / nsample.  nsample is a power of two in every shipped stack but the 2x classifier's (24): a
        // shift instead of the ~20-instruction 32-bit division sequence, per operand vector and chunk -- on fp32 MFMAs every
        // VALU instruction of the staging code comes straight out of the matrix pipe's issue slots
        const unsigned ns = (unsigned)o.ns, r = (unsigned)(r0 + rl);
        g = (ns & (ns - 1)) == 0 ? r >> (31 - __clz((int)ns)) : r / ns;
        raw.k = (int)(r - g * ns);
      }
      ldx<V>(o.a, (long long)g * o.lda + c, sb_a(OPM_POOLED), raw.a);
      ldvi<V>(o.arg + (long long)g * o.lda + c, raw.g);
      ldxu<V>(ub, offb, sb_b(OPM_POOLED), raw.b);
      if (rm) raw.m = rm->m;
      else if (o.mult) raw.m = (o.mult + r0)[url];
      break;
    }
    default: {
      const unsigned ns = (unsigned)o.ns, r = (unsigned)(r0 + rl);
      const unsigned g = (ns & (ns - 1)) == 0 ? r >> (31 - __clz((int)ns)) : r / ns;
      ldx<V>(o.a, (long long)g * o.lda + c, sb_a(OPM_BCAST), raw.a);
      break;
    }
  }
}

// The raw values of a prefetch, made opaque at the point of the call: whatever is computed from them is computed BEHIND this point.  The row
// GEMM calls it between the MFMAs of the running chunk and the commit of the next -- without it nothing keeps the commit's arithmetic
// (a pure function of the loaded values) from being placed right behind the loads, in front of the MFMAs, where it waits for the
// whole prefetch (round 6: seen in the ISA of the AFF2 / pooled instances; __builtin_amdgcn_sched_barrier orders the machine
// scheduler only, and only instructions that are already on its far side).
template <int V, int MODE>
__device__ __forceinline__ void op_pin(RawVec<V> &raw) {
  if constexpr (MODE >= 0) {
    constexpr int mode = opm_base(MODE);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      asm volatile("" : "+v"(raw.a[i]));
      if constexpr (mode == OPM_RELU2 || mode == OPM_AFF2 || mode == OPM_POOLED) asm volatile("" : "+v"(raw.b[i]));
      if constexpr (mode == OPM_POOLED) asm volatile("" : "+v"(raw.g[i]));
    }
  }
}

// The affine part (s2*b + t1) of the BatchNorm-backward operands is multiplied by the row's multiplicity:
// a compacted row stands for `m` identical copies whose pooled/masked gradients were already summed.
template <int V, int MODE>
__device__ __forceinline__ void op_finish(const RowOperand &o, const ColCoef<V> &k, const RawVec<V> &raw,
                                          long long r, bool ok, float (&out)[V]) {
  const int mode = MODE >= 0 ? opm_base(MODE) : o.mode;
  float ra[V], rb[V];
#pragma unroll
  for (int i = 0; i < V; ++i) { ra[i] = raw.a[i]; rb[i] = raw.b[i]; }
  bf16_expand<V>(ra, sb_a(mode));
  bf16_expand<V>(rb, sb_b(mode));
#pragma unroll
  for (int i = 0; i < V; ++i) {
    float v;
    switch (mode) {
      case OPM_ID: case OPM_BCAST: v = ra[i]; break;
      case OPM_RELU1: v = fmaxf(fmaf(k.s1[i], ra[i], k.t1[i]), 0.f); break;
      case OPM_RELU2: v = fmaxf(fmaf(k.s1[i], ra[i], k.t1[i]) + fmaf(k.s2[i], rb[i], k.t2[i]), 0.f); break;
      case OPM_AFF2: v = fmaf(k.s1[i], ra[i], raw.m * fmaf(k.s2[i], rb[i], k.t1[i])); break;
      default: v = fmaf(k.s1[i], (raw.g[i] == raw.k) ? ra[i] : 0.f, raw.m * fmaf(k.s2[i], rb[i], k.t1[i])); break;
    }
    out[i] = ok ? v : 0.f;
  }
}

// 16 bytes per lane global -> LDS without a register in between (LDS-DMA, global_load_lds_dwordx4): lane l's bytes land at
// lds + 16 l (wave-uniform destination), read from base + voff (wave-uniform base in SGPRs, per-lane unsigned byte offset).
// Written as an assembly statement ON PURPOSE: behind the builtin (__builtin_amdgcn_global_load_lds) the compiler's memory model
// cannot tell which LDS bytes the transfer writes and puts s_waitcnt vmcnt(0) in front of the next LDS read of ANY address -- the
// fragment reads of the running chunk then wait for the whole prefetch of the next one (measured, round 6: the step 3 % slower than
// the register-staged weights, profiles/r06/gemm_w_lds_dma.txt).  The statement is invisible to that pass, so the caller orders it
// itself: `glds_wait()` (s_waitcnt vmcnt(0)) in front of the barrier that publishes the stage.  M0 (the transfer's LDS base) is
// compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void glds16(const void *base, unsigned voff, const float *lds) {
  const unsigned dst = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)lds;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(__builtin_amdgcn_readfirstlane(dst)) : "memory");
}
__device__ __forceinline__ void glds_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

namespace {

constexpr int GM_THREADS = 256;
constexpr int GM_BM = 128;        // rows per workgroup tile (4 waves x 32)
constexpr int GM_BK = 32;         // reduction chunk per pipeline stage

enum { EPI_STORE = RS_EPI_STORE, EPI_STATS = RS_EPI_STATS, EPI_MASK = RS_EPI_MASK };
// the ABI's epilogue + what the launcher derives from it: w3_rows = rows (n) of the weights' three-part image
struct Epilogue : rs_mlp_epilogue { int w3_rows; };

// out[rows, cols] = E[rows, kdim] . W^T,  W[n][k] = w[n*ldw + k]: weights n-major (the conv weight's own
// (cout, cin) layout), ldw % 4 == 0, entries k in [kdim, ldw) zero.
//
// Pipeline per 32-deep K chunk: registers(next chunk) <- global  ||  MFMA(current chunk from LDS);
// two LDS stages, one barrier per chunk.
//
// LDS layout ("fragment-major"): v_mfma_f32_32x32x2_f32 wants, per k-step ks, lane (r = lane & 31, lk = lane >> 5)
// to hold A[r][2*ks + lk].  A chunk's 16 k-steps are stored as 8 planes p = 2*(ks >> 2) + lk of [row][ks & 3]
// float4s, so ONE ds_read_b128 per lane fetches its operand for FOUR k-steps (conflict-free: consecutive lanes,
// consecutive 16 B), and the fragments of the next four k-steps are read while the 4 x CT MFMAs of the current
// ones run.  (Reading one float per k-step right before its MFMAs exposed the ~100-cycle LDS latency twice per
// k-step and capped the loop near 50 % of the matrix pipe.)  The weights use the same layout with n in the row role.
template <int BM> struct AStage { static constexpr int PLANE = BM * 4 + 8; static constexpr int SIZE = 8 * PLANE; };   // +8 floats: the 4 planes one commit instruction hits land in different banks
template <int BN> struct WStage { static constexpr int PLANE = BN * 4 + 8; static constexpr int SIZE = 8 * PLANE; };

// LDS offset of element k (0..31, chunk-relative) of row r inside a stage with `plane` floats per plane
// Which k meets which MFMA step is free as long as both operands agree: of 4 consecutive k (one thread's vector), the first two
// go to the lk = 0 plane and the last two to the lk = 1 plane, adjacent slots -- the two 8-byte LDS stores of a vector then take
// their values from adjacent registers.  (Round 2 interleaved them, (k0, k0 + 2) | (k0 + 1, k0 + 3): 24 register moves per chunk
// to pair the values up, a tenth of the K loop's instructions.)
__device__ __forceinline__ int frag_off(int k, int r, int plane) {
  return ((k >> 3) * 2 + ((k >> 1) & 1)) * plane + r * 4 + (2 * ((k >> 2) & 1) + (k & 1));
}
// store V consecutive k of one row (k0 % V == 0)
template <int V>
__device__ __forceinline__ void frag_store(float *stage, int plane, int k0, int r, const float (&v)[V]) {
  if constexpr (V == 4) {          // (k0, k0+1) -> plane lk=0, slots i0, i0+1;  (k0+2, k0+3) -> plane lk=1, same slots
    float *b = stage + frag_off(k0, r, plane);
    *reinterpret_cast<float2 *>(b) = make_float2(v[0], v[1]);
    *reinterpret_cast<float2 *>(b + plane) = make_float2(v[2], v[3]);
  } else {
#pragma unroll
    for (int i = 0; i < V; ++i) stage[frag_off(k0 + i, r, plane)] = v[i];
  }
}

#ifdef RS_EXP_TIMING     // experiment build only (tools/gemm_variants.sh): per-phase shader-clock sums of wave 0
#define RS_T(i) do { const long long _n = clock64(); tacc[i] += _n - tlast; tlast = _n; } while (0)
#else
#define RS_T(i) do { } while (0)
#endif

// Tile = BM rows x BN columns, BM = 128 (4 waves stacked, each 32 rows x BN) or 64 (2 x 2 waves, each 32 rows x BN/2):
// the short tile doubles the number of workgroups -- with ~400-1000 tiles of 128 rows on 256 CUs the last round of a
// launch left a third of the chip idle -- and its 50 KB of LDS lets three workgroups share a CU.
//
// BF = true (rs_mlp_gemm_rows_bf16, BASELINE configs[4]): the operands are rounded to bf16 when they are committed to
// LDS -- AFTER the fp32 prologue -- and multiplied by v_mfma_f32_32x32x16_bf16 with fp32 accumulation; HBM tensors,
// prologue, epilogue and BatchNorm sums stay fp32.  A lane (row = lane & 31, g = lane >> 5) feeds k = 16 s + 8 g .. + 7
// of its row to MFMA step s, so a 32-deep chunk is FOUR planes p = k >> 3 of [row][4 dwords = 8 bf16]: one
// ds_read_b128 per operand per step, two steps per chunk (16 fp32 steps otherwise); plane pad 16 dwords keeps the
// 8-byte commit stores of a half-wave (4 rows x 4 planes x 2 halves) on distinct banks.
//
// WS = true (round 3, 64-row tiles; an EXPERIMENT kept behind RS_GEMM_WS=1, off by default): WAVE SPECIALISATION.  The workgroup has 8 waves: waves 0-3 only read fragments and issue
// MFMAs (and write the tile out), waves 4-7 only move data -- global loads two chunks ahead (two register sets), the operand
// prologue, the LDS commit.  One barrier per chunk joins them: the loaders arrive when chunk g is in LDS, the MFMA waves when they
// are done with chunk g - 1, so commit(g) runs under the MFMAs of g - 1 and the matrix pipe never waits for a global load or a
// prologue of its own wave.  (Without it a wave alternates between the two roles; two workgroups per CU were meant to cover each
// other's staging phases, but workgroups launched together run their phases in lockstep -- the loop issued MFMAs 27-40 % of a
// tile's cycles)  The stage a chunk uses is the parity of a RUNNING chunk count across tiles, so the loaders start
// the next tile while the MFMA waves are still in the epilogue of this one.
// What the per-role cycle stamps showed (tools/gemm_bench.py timing, profiles/r03/gemm_ws_timing.txt): the loaders are the
// bottleneck -- issuing one chunk's ~50 address / load instructions takes 3 600 cycles beside running MFMAs and 800 with the
// MFMAs compiled out, whatever the loaders' priority (s_setprio 3) or age (waves 0-3 or 4-7).  v_mfma_f32_32x32x2_f32 holds its
// SIMD's issue for 64 cycles (MI355X_MICROARCH.md) and leaves about one slot per MFMA to everything else on that SIMD, so
// on fp32 MFMAs "a wave that only loads" does not run BESIDE the matrix stream, it runs in its gaps: the kernel's
// efficiency is 64 / (64 + issue cycles of the non-MFMA instructions per MFMA) whichever wave executes them.
template <int BM, int BN, int V, int MODE, bool BF, bool WS = false>
// (the 64 x 64 forward instances of the split products -- 50 KB of LDS: THREE workgroups fit a CU -- are held to the 168 VGPRs that takes;
//  they used 183-195 after the weights' units got registers of their own, and fit without a spill)
// (round 6: the occupancy the instruction scheduler aims for is pinned to that number -- amdgpu_waves_per_eu(n, n).  With the weights'
//  registers gone the forward instances fell to 138 VGPRs, the scheduler took that as an invitation to reach FOUR waves per SIMD (128) and
//  serialised the fragment reads -- ds_read, s_waitcnt lgkmcnt(0), MFMA, one pair at a time -- to get there; LDS allows three workgroups.)
#if defined(RS_EXP_WG2)
#define RS_GEMM_WG_PER_CU (WS ? 4 : 2)
#elif defined(RS_EXP_WG3_ALL)      /* every 64 x 64 split instance at three workgroups per CU (168 VGPRs) */
#define RS_GEMM_WG_PER_CU (WS ? 4 : ((BF && RS_SPLIT && BM == 64 && BN == 64 && MODE >= 0) ? 3 : 2))
#else
#define RS_GEMM_WG_PER_CU (WS ? 4 : ((BF && RS_SPLIT && BM == 64 && BN == 64 && MODE >= 0 && MODE <= OPM_RELU2) ? 3 : 2))
#endif
// (the attribute takes effect only when __launch_bounds__ carries no second argument: that one writes "amdgpu-waves-per-eu"="n" -- a minimum -- over it)
__global__ void __launch_bounds__(WS ? 2 * GM_THREADS : GM_THREADS)     // >= 2 workgroups per CU: one computes while another stages
__attribute__((amdgpu_waves_per_eu(WS ? 8 : RS_GEMM_WG_PER_CU, WS ? 8 : RS_GEMM_WG_PER_CU)))
gemm_rows_kernel(long long rows_arg, const int *__restrict__ rows_dev, int kdim, int cols, RowOperand E,
                 const float *__restrict__ w, int ldw, Epilogue ep) {
  // compacted inputs carry their row count on the device (no host sync); rows_arg is then the capacity
  const long long rows = rows_dev ? min(rows_arg, (long long)*rows_dev) : rows_arg;
  constexpr int WR = BM / 32, WC = 4 / WR;                  // waves along rows / columns of the tile
  constexpr int CT = BN / 32 / WC;                          // 32-column MFMA tiles per wave
#ifdef RS_EXP_LDS_EPILOGUE
  constexpr bool DIRECT = false;
#else
  constexpr bool DIRECT = BM == 64;                         // short tiles: epilogue straight from the accumulators
#endif
#if defined(RS_EXP_EARLY_PREFETCH_ALL)
  constexpr bool EARLY_PREFETCH = DIRECT;
#elif defined(RS_EXP_EARLY_PREFETCH_FWD)
  constexpr bool EARLY_PREFETCH = DIRECT && MODE >= 0 && MODE <= OPM_RELU2;
#else
  // Requesting the next tile's first chunk in front of the write-out was measured on one box (60-step replays, two runs
  // each): off 1.7201 / 1.7225 ms per step, forward instances only 1.7338 / 1.7279, all instances 1.7459 / 1.7512 -- the
  // ~20 extra live VGPRs cost more than the hidden latency brings.  Kept for experiments only.
  constexpr bool EARLY_PREFETCH = false;
#endif
  // 128-row tiles (four stacked waves, the 32 / 64-column layers of the segmentation step's 524 288-row stages): the same epilogue
  // straight from the accumulators unless a max-pool over groups of other than 32 rows is folded into this launch (that one walks whole
  // groups through the C tile in LDS; groups of 32 rows are pooled in the accumulators, see the epilogue).  Round 4: the LDS epilogue of these tiles -- C tile through LDS, three barriers, one exposed mask-load round trip per
  // 32 rows -- held the 524 288 x 32 launches at 0.30 of the HBM roof.  A run-time, wave-uniform choice: both forms live in the
  // instance.  -DRS_NO_DIRECT128 restores the LDS epilogue for A/B runs.
#ifdef RS_NO_DIRECT128
  const bool direct = DIRECT;
#else
  constexpr bool D128 = BM == 128 && BN <= 64 && !WS;       // (128-column tiles: four accumulator tiles per wave, the two forms in one
  const bool direct = DIRECT || (D128 && (ep.pool_ns <= 0 || ep.pool_ns == 32));  //  instance spill 0.5-2 KB per lane: they keep the LDS epilogue)
#endif
  constexpr int NSLOT = 2 * WR;                             // (row wave, lane half) pairs that hold sums of one column
  static_assert(CT >= 1, "tile too narrow for the wave layout");
  static_assert(!BF || V >= 2, "bf16 staging packs pairs of k");
  constexpr int PLANE_A = BF ? BM * 4 + 16 : AStage<BM>::PLANE;
  constexpr int A_ELEMS = BM * GM_BK / GM_THREADS;          // floats of the operand tile per thread (16 / 8)
  constexpr int A_VECS = A_ELEMS / V;
  constexpr int A_TPR = GM_BK / V;                          // threads per tile row
  constexpr int A_RPP = GM_THREADS / A_TPR;                 // rows per pass
  constexpr int W_VECS = BN / 32;                           // float4 (4 k of one output column) per thread and chunk
  constexpr int PLANE_W = BF ? BN * 4 + 16 : WStage<BN>::PLANE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int PARTS = (BF && RS_SPLIT) ? 3 : 1;         // unit 4: three bf16 parts per value, each a stage of 4 planes
  constexpr int PART_A = 4 * PLANE_A, PART_W = 4 * PLANE_W;
  constexpr int A_STAGE = PARTS == 3 ? 3 * PART_A : AStage<BM>::SIZE, W_STAGE = PARTS == 3 ? 3 * PART_W : WStage<BN>::SIZE;
  float *As0 = smem, *As1 = smem + A_STAGE;
  float *Ws0 = smem + 2 * A_STAGE, *Ws1 = Ws0 + W_STAGE;
  static_assert(!WS || BM == 64, "wave specialisation is built for the 64-row tiles (direct epilogue)");
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3;
#ifdef RS_EXP_WS_LOADERS_LAST
  const bool loader = WS && tid >= GM_THREADS;              // waves 4-7 of a specialised workgroup
#else
  // the loaders are waves 0-3: VALU / VMEM issue on a SIMD is arbitrated by priority, then AGE (MI355X_MICROARCH.md), and the
  // few instructions that keep the matrix pipe fed must not queue behind the MFMA waves' fragment reads
  const bool loader = WS && tid < GM_THREADS;
#endif
  const int ltid = tid & (GM_THREADS - 1);                  // position among the 256 threads of a role
  const int n0 = blockIdx.y * BN;
  const long long tiles = (rows + BM - 1) / BM;
  const int lrow = lane & 31, lk = lane >> 5;
  const int wave_r = wave % WR, wave_c = wave / WR;
  const int a_kq = (ltid % A_TPR) * V, a_r = ltid / A_TPR;
  const int w_kq = (ltid & 7) * 4, w_n = ltid >> 3;         // weights: 8 threads x float4 cover the 32 k of a column
  const int nchunks = (kdim + GM_BK - 1) / GM_BK;

  // epilogue geometry: the finished tile goes through LDS so that global traffic is row-major float4
  constexpr int E_TPR = BN / 4;                 // threads per tile row (4 columns each)
  constexpr int E_RPP = GM_THREADS / E_TPR;     // rows per pass
  const int e_col = (tid % E_TPR) * 4, e_row = tid / E_TPR;
  const bool ep_vec = (((uintptr_t)ep.out | (uintptr_t)ep.my1 | (uintptr_t)ep.my2) % 16 == 0) && (ep.ldo % 4 == 0) &&
                      (ep.ldm1 % 4 == 0) && (ep.ldm2 % 4 == 0) && (cols % 4 == 0);

  // (A software-pipelined form of this loop -- the MFMAs of chunk g and the commit of chunk g + 1 in one scheduling region, two register
  //  sets -- was written at the end of round 4 and measured in round 5: slower in every class and in both steps, 1.283 -> 1.391 ms
  //  classification, 3.378 -> 3.628 ms segmentation, profiles/r05/sp_pipe_ab.txt.  Removed.)
  constexpr bool PIPE = false;
  constexpr int NSET = (WS || PIPE) ? 2 : 1;                // WS: the loaders keep two chunks of raw operands in flight
  // Pre-split weights (round 5, ep.w3; round 6: tile-ordered image, LDS-DMA): a chunk's weight tile is 3 parts x 4 planes (8 k each) x BN
  // columns of 16 bytes (8 bf16 of one column).  The image in HBM is w3[q][k / 8][n][8]: the BN units of one (part, plane) are
  // CONTIGUOUS and in the order they lie in LDS, so a plane travels global -> LDS as one global_load_lds_dwordx4 per 64 columns
  // (lane = column; destination = wave-uniform base + 16 lane) -- no VGPRs, no ds_write, no weight work in the commit.  Wave w
  // moves plane w of every part: 3 ceil(BN / 64) instructions per wave and chunk.
  constexpr int W_HALVES = (BN + 63) / 64;
  constexpr int WR_VECS = PARTS == 3 ? 0 : W_VECS;          // (the split-product instances run ONLY on pre-split weights: the entry point sends launches without an image to the fp32 MFMA instances)
  RawVec<V> araw[NSET][A_VECS];
  // group / slot / multiplicity of this thread's rows: once per tile for the operand modes that have them (the specialised instances;
  // the wave-specialised loaders run ahead of the tile loop and keep the per-chunk form)
  constexpr bool ROW_META = !WS && !EARLY_PREFETCH && MODE >= 0 && (opm_base(MODE) == OPM_POOLED || MODE == OPM_AFF2);
  RowMeta rmeta[A_VECS];
  float4 wraw[NSET][WR_VECS > 0 ? WR_VECS : 1];
  ColCoef<V> coef[NSET];
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, NSET - 1>;

#ifdef RS_EXP_TIMING
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = clock64();
  const long long tstart = tlast;
#endif
  // Loads are never predicated: out-of-range rows / k are CLAMPED to the last valid element (always in bounds,
  // finite) and zeroed when the values are committed to LDS -- straight-line code, no exec-mask branches.
  // `part` < 0 issues the whole chunk.  (Spreading the parts 0..3 over the four MFMA groups of the running chunk was
  // measured and is worse: a load that cannot issue stalls the wave in front of its next MFMA -- in-order issue --
  // 126 us against 116 us at 262144 x 128 x 128, weight gradient 410 us against 320 us.)
  auto prefetch = [&](auto set_, long long r0, int k0, int part, int wstage = 0) {
    constexpr int S = decltype(set_)::value;
    const int k = min(k0 + a_kq, kdim - V);                 // kdim % V == 0
    const int rlast = (int)min((long long)BM - 1, rows - 1 - r0);
    if constexpr (PARTS == 3) {
      // the chunk's weight planes straight into stage `wstage` (free: the caller is past the barrier behind that stage's last readers).
      // Issued IN FRONT of the operand loads: the compiler's s_waitcnt pass does not see these transfers, and its vmcnt(n) for an operand
      // load counts only the loads it knows BEHIND that one -- exact as long as nothing invisible is younger (return is in order)
      float *wdst = smem + (2 * A_STAGE + wstage * W_STAGE) + wave * PLANE_W;
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int h = 0; h < W_HALVES; ++h) {
          const int nl = h * 64 + lane;
          if (BN >= 64 || lane < BN) {
            const int n = min(n0 + nl, cols - 1);               // a column beyond `cols` (clamped: finite) feeds only its own column of D
            const unsigned off = (unsigned)q * (unsigned)ep.w3_part + (((unsigned)(k0 >> 3) + (unsigned)wave) * (unsigned)ep.w3_rows + (unsigned)n) * 8u;   // elements; an image is < 2^31 bytes
            glds16(ep.w3, 2u * off, wdst + q * PART_W + h * 256);
          }
        }
    }
    if (part <= 0) op_coef<V, MODE>(E, k, true, coef[S]);
#pragma unroll
    for (int p = 0; p < A_VECS; ++p)
      if (part < 0 || (p * 4) / A_VECS == part) op_load<V, MODE>(E, r0, min(p * A_RPP + a_r, rlast), k, true, araw[S][p], ROW_META ? &rmeta[p] : nullptr);
    if constexpr (PARTS != 3) {
      const int kw = min(k0 + w_kq, ldw - 4);
#pragma unroll
      for (int p = 0; p < W_VECS; ++p)
        if (part < 0 || (p * 4) / W_VECS == part) {
          const int n = min(n0 + p * 32 + w_n, cols - 1);
          wraw[S][p] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(w) + 4u * ((unsigned)n * (unsigned)ldw + (unsigned)kw));   // SGPR base + 32-bit offset
        }
    }
  };
  auto commit = [&](auto set_, float *As, float *Ws, long long r0, int k0) {
    constexpr int S = decltype(set_)::value;
#ifdef RS_EXP_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (timing build: the wait for the chunk's global loads on its own stamp)
    RS_T(7);
#endif
    const bool kok = (k0 + a_kq) < kdim;
#pragma unroll
    for (int p = 0; p < A_VECS; ++p) {
      const int rl = p * A_RPP + a_r;
      const long long r = r0 + rl;
      float v[V];
      // only k >= kdim has to read as zero: a row beyond the end was loaded from the tile's last valid row (finite) and feeds
      // nothing but its own row of D, which no epilogue stores or sums
      op_finish<V, MODE>(E, coef[S], araw[S][p], r, kok, v);
      if constexpr (BF) {                                     // k = a_kq .. a_kq + V - 1 -> plane k >> 3, dword (k & 7) >> 1
        float *b = As + (a_kq >> 3) * PLANE_A + rl * 4 + ((a_kq & 7) >> 1);
        if constexpr (PARTS == 3) {
          float d0[3], d1[3];
          split_bf16(v[0], v[1], d0);
          if constexpr (V == 4) split_bf16(v[2], v[3], d1);
#ifdef RS_EXP_SP_ONE_STORE     // what-if build (timing only, wrong results): a third of the LDS writes, the split still computed
          d0[0] += d0[1] + d0[2]; if constexpr (V == 4) d1[0] += d1[1] + d1[2];
#endif
#pragma unroll
          for (int q = 0; q < 3; ++q) {
#ifdef RS_EXP_SP_ONE_STORE
            if (q > 0) continue;
#endif
            if constexpr (V == 4) *reinterpret_cast<float2 *>(b + q * PART_A) = make_float2(d0[q], d1[q]);
            else b[q * PART_A] = d0[q];
          }
        } else
        if constexpr (V == 4) *reinterpret_cast<float2 *>(b) = make_float2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
        else *b = pack_bf16(v[0], v[1]);
      } else {
        frag_store<V>(As, PLANE_A, a_kq, rl, v);
      }
    }
    // The weights go to LDS as loaded: k in [kdim, ldw) is zero in memory, k beyond ldw (clamped to the row's last vector) meets
    // operand values that were committed as zero, and a column beyond `cols` (clamped to the last one: finite) feeds only its own
    // column of D, which no epilogue stores or sums.  (16 selects per chunk, 7 % of the loop, until round 3.)
    if constexpr (PARTS == 3) {
      // the weight planes are written by the LDS-DMA the prefetch issued: nothing to commit, but the transfers must have landed
      // before the barrier behind this commit lets any wave read the stage (a pending LDS-DMA counts on vmcnt; the operand loads
      // above were waited for already, so this costs nothing extra)
      glds_wait();
    } else {
#pragma unroll
    for (int p = 0; p < W_VECS; ++p) {
      const int nl = p * 32 + w_n;
      const float v[4] = {wraw[S][p].x, wraw[S][p].y, wraw[S][p].z, wraw[S][p].w};
      if constexpr (PARTS == 3) {
        float d0[3], d1[3];
        split_bf16(v[0], v[1], d0);
        split_bf16(v[2], v[3], d1);
#ifdef RS_EXP_SP_ONE_STORE
        d0[0] += d0[1] + d0[2]; d1[0] += d1[1] + d1[2];
#endif
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#ifdef RS_EXP_SP_ONE_STORE
          if (q > 0) continue;
#endif
          *reinterpret_cast<float2 *>(Ws + q * PART_W + (w_kq >> 3) * PLANE_W + nl * 4 + ((w_kq & 7) >> 1)) = make_float2(d0[q], d1[q]);
        }
      } else if constexpr (BF)
        *reinterpret_cast<float2 *>(Ws + (w_kq >> 3) * PLANE_W + nl * 4 + ((w_kq & 7) >> 1)) =
            make_float2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
      else
        frag_store<4>(Ws, PLANE_W, w_kq, nl, v);
    }
    }
  };

  float st0[CT], st1[CT], st2[CT];                          // DIRECT: this lane's column sums over all its tiles
#pragma unroll
  for (int c = 0; c < CT; ++c) { st0[c] = 0.f; st1[c] = 0.f; st2[c] = 0.f; }

  // MFMAs of one 32-deep chunk staged at (As, Ws) into acc[]
  auto mma = [&](f32x16 (&acc)[CT], const float *As, const float *Ws, int ch) {
      if constexpr (PARTS == 3) {
        // per step: the three parts of both operands (3 + 3 CT ds_read_b128), then 6 x CT MFMAs, smallest terms first, the column
        // tiles alternating (independent accumulators back to back).  (A wave with ONE column tile runs a chunk's 12 MFMAs as one
        // dependent chain; sending the three small terms to a second accumulator, folded in once per tile, was measured: the
        // kernels alone unchanged, the classification step 1.350 -> 1.364 ms -- 16 more live VGPRs, and the other waves of the SIMD
        // already fill the chain's gaps.)
        const float *ap = As + lk * PLANE_A + (wave_r * 32 + lrow) * 4;
        const float *bp = Ws + lk * PLANE_W + (wave_c * CT * 32 + lrow) * 4;
#if defined(RS_EXP_HOIST_CT1)      // (round 6 experiment, not in the product: equal or slower in both steps -- 48 fragment VGPRs; profiles/r06/gemm_w_lds_dma.txt)
        if constexpr (CT == 1) {
          // one column tile per wave (64 x 64 tiles): BOTH steps' fragments first -- 12 ds_read_b128, 48 VGPRs -- then the 12 MFMAs,
          // waiting for the fragments in the order they were requested.  (Round 6: left to itself the scheduler issued the reads in
          // pairs, each followed by s_waitcnt lgkmcnt(0) and its MFMA: six exposed LDS round trips per chunk.)
          float4 a6[2][3], b6[2][3];
#pragma unroll
          for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int q = 2; q >= 0; --q) {                      // (l, m, h: the order the products below consume them)
              a6[st][q] = *reinterpret_cast<const float4 *>(ap + q * PART_A + 2 * st * PLANE_A);
              b6[st][q] = *reinterpret_cast<const float4 *>(bp + q * PART_W + 2 * st * PLANE_W);
            }
          __builtin_amdgcn_sched_barrier(0);
          constexpr int TA1[6] = {2, 0, 1, 1, 0, 0}, TB1[6] = {0, 2, 1, 0, 1, 0};      // lh, hl, mm, mh, hm, hh
#pragma unroll
          for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int t = 0; t < 6; ++t)
              acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a6[st][TA1[t]]),
                                                               __builtin_bit_cast(bf16x8, b6[st][TB1[t]]), acc[0], 0, 0, 0);
          return;
        }
#endif
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          float4 a3[3], b3[3][CT];
#pragma unroll
          for (int q = 0; q < 3; ++q) {
#ifdef RS_EXP_SP_ONE_FRAG      // what-if build (timing only, wrong results): a third of the fragment reads
            if (q > 0) { a3[q] = a3[0]; for (int c = 0; c < CT; ++c) b3[q][c] = b3[0][c]; continue; }
#endif
            a3[q] = *reinterpret_cast<const float4 *>(ap + q * PART_A + 2 * st * PLANE_A);
#pragma unroll
            for (int c = 0; c < CT; ++c) b3[q][c] = *reinterpret_cast<const float4 *>(bp + q * PART_W + 2 * st * PLANE_W + c * 128);
          }
          constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};      // lh, hl, mm, mh, hm, hh
#pragma unroll
#ifdef RS_EXP_SP_ONE_MFMA      // what-if build (timing only, wrong results): a sixth of the MFMAs, every fragment still read and consumed
          for (int t = 5; t < 6; ++t)
#pragma unroll
            for (int c = 0; c < CT; ++c) {
              float4 aa = a3[0], bb = b3[0][c];
              aa.x += a3[1].x + a3[2].x; bb.x += b3[1][c].x + b3[2][c].x;
              acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aa), __builtin_bit_cast(bf16x8, bb), acc[c], 0, 0, 0);
            }
#else
          for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int c = 0; c < CT; ++c)
              acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a3[TA[t]]),
                                                               __builtin_bit_cast(bf16x8, b3[TB[t]][c]), acc[c], 0, 0, 0);
#endif
        }
      } else if constexpr (BF) {
        // both steps' fragments first (2 + 2 CT ds_read_b128), then 2 x CT MFMAs; k beyond kdim was committed as zero
        const float *ap = As + lk * PLANE_A + (wave_r * 32 + lrow) * 4;
        const float *bp = Ws + lk * PLANE_W + (wave_c * CT * 32 + lrow) * 4;
        float4 af[2], bf[2][CT];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          af[st] = *reinterpret_cast<const float4 *>(ap + 2 * st * PLANE_A);
#pragma unroll
          for (int c = 0; c < CT; ++c) bf[st][c] = *reinterpret_cast<const float4 *>(bp + 2 * st * PLANE_W + c * 128);
        }
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int c = 0; c < CT; ++c)
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[st]),
                                                             __builtin_bit_cast(bf16x8, bf[st][c]), acc[c], 0, 0, 0);
      } else {
      // groups of 4 k-steps; the last chunk of a ragged K runs only the groups that hold data
      const int ngroups = (min(GM_BK, kdim - ch * GM_BK) + 7) >> 3;
      const float *ap = As + lk * PLANE_A + (wave_r * 32 + lrow) * 4;
      const float *bp = Ws + lk * PLANE_W + (wave_c * CT * 32 + lrow) * 4;
      float4 af[2], bf[2][CT];
      af[0] = *reinterpret_cast<const float4 *>(ap);
#pragma unroll
      for (int c = 0; c < CT; ++c) bf[0][c] = *reinterpret_cast<const float4 *>(bp + c * 128);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < ngroups) {                                    // wave-uniform
          if (j + 1 < 4) {                                    // fragments of the NEXT group: in flight under this group's MFMAs
            af[(j + 1) & 1] = *reinterpret_cast<const float4 *>(ap + (j + 1) * 2 * PLANE_A);
#pragma unroll
            for (int c = 0; c < CT; ++c)
              bf[(j + 1) & 1][c] = *reinterpret_cast<const float4 *>(bp + (j + 1) * 2 * PLANE_W + c * 128);
          }
          const float a4[4] = {af[j & 1].x, af[j & 1].y, af[j & 1].z, af[j & 1].w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int c = 0; c < CT; ++c) {
              const float4 bq = bf[j & 1][c];
              const float b = i == 0 ? bq.x : (i == 1 ? bq.y : (i == 2 ? bq.z : bq.w));
              acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i], b, acc[c], 0, 0, 0);
            }
          }
        }
      }
      }
  };

  long long gchunk = 0;                                     // WS: chunks this workgroup has consumed so far (stage parity)
  const long long ws_total = ((long long)blockIdx.x < tiles ? (tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0) * nchunks;   // chunks of this workgroup
  if (WS && loader) {
    // ---- loader waves: (tile, chunk) pairs in the order the MFMA waves consume them, raw operands two chunks ahead.
    // Every prefetch is UNCONDITIONAL (past the last chunk it re-reads the last tile: wasted, harmless): with a prefetch
    // under `if (more work)` the compiler's s_waitcnt pass merges the two paths and makes commit(set 0) wait for the loads
    // of set 1 as well (seen in the ISA: vmcnt(5..0) where vmcnt(13..8) was meant) -- the two-chunk distance collapsed to one.
    // The loaders run at raised priority: they issue ~50 instructions per chunk against ~100 of an MFMA wave, and an
    // arbiter that serves the MFMA waves first lets each of those instructions wait for a gap in the matrix stream.
    // The loop body is branch-free: two chunks per trip (set 0 -> stage 0, set 1 -> stage 1), an odd total is padded with
    // one dummy chunk (its barrier is matched by the MFMA waves below) -- a `break` between the halves rejoined the latch
    // in the structurised CFG and had the same effect on the wait counts as the conditional prefetch.
    __builtin_amdgcn_s_setprio(3);
    auto adv = [&](long long &t, int &c) { if (++c == nchunks) { c = 0; t += gridDim.x; } };
    auto rbase = [&](long long t) { return (t < tiles ? t : tiles - 1) * BM; };
    long long t0 = blockIdx.x, t1 = blockIdx.x;               // the chunk set 0 / set 1 holds (or is loading)
    int c0 = 0, c1 = 0;
    adv(t1, c1);
    const long long pairs = (ws_total + 1) >> 1;
    if (pairs > 0) {
      prefetch(S0{}, rbase(t0), c0 * GM_BK, -1);
      prefetch(S1{}, rbase(t1), c1 * GM_BK, -1);
    }
    for (long long pr = 0; pr < pairs; ++pr) {
      commit(S0{}, As0, Ws0, rbase(t0), c0 * GM_BK);
      RS_T(1);
      __syncthreads();
      RS_T(2);
      adv(t0, c0); adv(t0, c0);
      prefetch(S0{}, rbase(t0), c0 * GM_BK, -1);
      RS_T(3);
      commit(S1{}, As1, Ws1, rbase(t1), c1 * GM_BK);
      RS_T(1);
      __syncthreads();
      RS_T(2);
      adv(t1, c1); adv(t1, c1);
      prefetch(S1{}, rbase(t1), c1 * GM_BK, -1);
      RS_T(3);
    }
  } else {
  for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long long r0 = tile * BM;
    f32x16 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;

    if constexpr (WS) {
      for (int ch = 0; ch < nchunks; ++ch) {
        __syncthreads();                                        // chunk gchunk is in LDS (the loaders passed the same barrier)
        RS_T(2);
#ifndef RS_EXP_WS_NOMFMA
        mma(acc, (gchunk & 1) ? As1 : As0, (gchunk & 1) ? Ws1 : Ws0, ch);
#endif
        RS_T(4);
        ++gchunk;
      }
    } else {
    if constexpr (ROW_META) {
      const int rlast0 = (int)min((long long)BM - 1, rows - 1 - r0);
#pragma unroll
      for (int p = 0; p < A_VECS; ++p) op_row_meta<MODE>(E, r0, min(p * A_RPP + a_r, rlast0), rmeta[p]);
    }
    // DIRECT: the first chunk of this tile was requested in front of the previous tile's epilogue (below)
    if (!EARLY_PREFETCH || tile == (long long)blockIdx.x) prefetch(S0{}, r0, 0, -1);
    RS_T(0);
#if defined(RS_EXP_OLD_LOOP)
    for (int ch = 0; ch < nchunks; ++ch) {
      float *As = (ch & 1) ? As1 : As0;
      float *Ws = (ch & 1) ? Ws1 : Ws0;
      commit(S0{}, As, Ws, r0, ch * GM_BK);
      RS_T(1);
      __syncthreads();                                        // tile chunk visible; stage ch-1 free again
      RS_T(2);
      if (ch + 1 < nchunks) prefetch(S0{}, r0, (ch + 1) * GM_BK, -1, (ch + 1) & 1);   // loads fly under the MFMAs below (unit 4: the weights' planes go straight into the other stage)
      RS_T(3);
      mma(acc, As, Ws, ch);
      RS_T(4);
    }
#else
    // Round 6: the loop is rotated -- an iteration requests chunk ch + 1, multiplies chunk ch and commits chunk ch + 1 -- so that the
    // registers a prefetch fills are consumed in the SAME iteration.  (As `commit; barrier; prefetch; mma` the raw operand values were
    // loop-carried, and the copies the register allocator placed on the back edge for two of them -- it reuses their registers as LDS
    // address temporaries of the MFMA block -- sat behind an s_waitcnt vmcnt right after the prefetch: the whole load latency in
    // front of the MFMAs it was meant to hide under.  Seen in the ISA once the weights' own registers were gone.)
    commit(S0{}, As0, Ws0, r0, 0);
    RS_T(1);
    __syncthreads();
    RS_T(2);
    for (int ch = 0; ch + 1 < nchunks; ++ch) {
      prefetch(S0{}, r0, (ch + 1) * GM_BK, -1, (ch + 1) & 1);   // loads fly under the MFMAs below (unit 4: the weights' planes go straight into the other stage)
      RS_T(3);
      __builtin_amdgcn_sched_barrier(0);                        // (one scheduling region otherwise: the loads sink to their uses in the commit, the commit's arithmetic rises into the MFMAs and waits for the loads there)
      mma(acc, (ch & 1) ? As1 : As0, (ch & 1) ? Ws1 : Ws0, ch);
      RS_T(4);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < A_VECS; ++p) op_pin<V, MODE>(araw[0][p]);
      commit(S0{}, (ch & 1) ? As0 : As1, (ch & 1) ? Ws0 : Ws1, r0, (ch + 1) * GM_BK);
      RS_T(1);
      __syncthreads();                                        // chunk ch + 1 visible; every wave is done with chunk ch's stage
      RS_T(2);
    }
    mma(acc, ((nchunks - 1) & 1) ? As1 : As0, ((nchunks - 1) & 1) ? Ws1 : Ws0, nchunks - 1);
    RS_T(4);
#endif
    }
    if (direct) {
      // ---- epilogue straight from the accumulators (64-row tiles; 128-row tiles without fused pooling).  D[i][j]: j = lane & 31 is the output column,
      // i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) the row: one store instruction covers two 128-byte row
      // segments (full cache lines), a lane keeps ONE column per accumulator tile, so the BatchNorm column sums are
      // lane-local fp32 sums that stay in registers over all tiles of this workgroup (<= 3 x CT VGPRs) and meet in
      // LDS once, after the tile loop.  No C tile in LDS, no barrier between the K loop and the stores: the tile's
      // write-out overlaps the other workgroup's main loop instead of a chip-wide store burst behind three barriers.
      RS_T(5);
      // the next tile's first operand chunk travels while this tile's results leave: the epilogue needs no LDS and few
      // registers, and the loads have the whole write-out to land (they used to be issued behind it and waited for in full)
      if (EARLY_PREFETCH && tile + gridDim.x < tiles) prefetch(S0{}, (tile + gridDim.x) * BM, 0, -1);
      const int EPI = ep.mode;
      int rbase = wave_r * 32 + 4 * lk;
      asm volatile("" : "+v"(rbase));                         // opaque per tile: the 16 x 3 row offsets below are recomputed here,
                                                              // not hoisted out of the tile loop into ~100 long-lived VGPRs
      const int ldo = (int)ep.ldo, ldm1 = (int)ep.ldm1, ldm2 = (int)ep.ldm2;      // tile-relative offsets fit 32 bits
      const bool full = r0 + BM <= rows;
      const bool obf = sb_out(MODE >= 0 ? opm_base(MODE) : E.mode);     // bf16 tensors: same element offsets, half the bytes
      float *out_t = tile_base(ep.out, r0 * ep.ldo, obf);
      const float *my1_t = ep.my1 ? tile_base(ep.my1, r0 * ep.ldm1, SB_MASK) : nullptr;
      const float *my2_t = ep.my2 ? tile_base(ep.my2, r0 * ep.ldm2, SB_MASK) : nullptr;
      // forward operand modes never come with the mask epilogue, backward ones never with the statistics one:
      // the dead branch (and its registers) vanishes from the specialised instances
      constexpr bool CAN_STATS = MODE < 0 || MODE <= OPM_RELU2, CAN_MASK = MODE < 0 || MODE >= OPM_AFF2;
      // Row validity is ONE 32-bit compare against the tile's last valid row, and every address of the epilogue is a wave-uniform
      // base (SGPRs) + an unsigned 32-bit byte offset per lane -- the form global loads / stores take without a 64-bit VGPR pair.
      // (With `r0 + rl < rows` in 64 bits and int element indices each of the 16-32 stores and 16-64 mask loads of a tile carried
      // ~17 instructions: 300-1 000 per wave and tile, as much as the K loop of the 64 ... 128-channel layers.)
      const int last = (int)min((long long)(BM - 1), rows - 1 - r0);
      float mw[16];
      if (CAN_STATS && EPI == EPI_STATS) {
#pragma unroll
        for (int i = 0; i < 16; ++i) mw[i] = 1.f;
        if (ep.row_mult) {
          const char *mwp = reinterpret_cast<const char *>(ep.row_mult + r0);
#pragma unroll
          for (int i = 0; i < 16; ++i)                            // unconditional: a row beyond the end reads the last valid one (unused)
            mw[i] = *reinterpret_cast<const float *>(mwp + 4u * (unsigned)min(rbase + (i & 3) + 8 * (i >> 2), last));
        }
      }
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const int col = n0 + (wave_c * CT + c) * 32 + lrow;
        const bool cok = col < cols;
        const int cc = cok ? col : cols - 1;
        float y1v[16], y2v[16];
        if (CAN_MASK && EPI == EPI_MASK) {                    // the 16 (32) mask loads of this column tile in flight at once
          unsigned mrl[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) mrl[i] = (unsigned)(full ? rbase + (i & 3) + 8 * (i >> 2) : min(rbase + (i & 3) + 8 * (i >> 2), last));
          const char *m1c = reinterpret_cast<const char *>(my1_t), *m2c = reinterpret_cast<const char *>(my2_t);
          // bf16 tensors: raw half words first (one branch around all loads), widened once they are all requested
          if (SB_MASK) {
#pragma unroll
            for (int i = 0; i < 16; ++i) y1v[i] = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short *>(m1c + 2u * (mrl[i] * (unsigned)ldm1 + (unsigned)cc)));
          } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) y1v[i] = *reinterpret_cast<const float *>(m1c + 4u * (mrl[i] * (unsigned)ldm1 + (unsigned)cc));
          }
          if (!my2_t) {
#pragma unroll
            for (int i = 0; i < 16; ++i) y2v[i] = 0.f;
          } else if (SB_MASK) {
#pragma unroll
            for (int i = 0; i < 16; ++i) y2v[i] = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short *>(m2c + 2u * (mrl[i] * (unsigned)ldm2 + (unsigned)cc)));
          } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) y2v[i] = *reinterpret_cast<const float *>(m2c + 4u * (mrl[i] * (unsigned)ldm2 + (unsigned)cc));
          }
          if (SB_MASK) {
#pragma unroll
            for (int i = 0; i < 16; ++i) y1v[i] = __uint_as_float(__float_as_uint(y1v[i]) << 16);
          }
          if (SB_MASK && my2_t) {
#pragma unroll
            for (int i = 0; i < 16; ++i) y2v[i] = __uint_as_float(__float_as_uint(y2v[i]) << 16);
          }
        }
        const float bias = ep.bias ? ep.bias[cc] : 0.f;
        float ms1 = 0.f, mt1 = 0.f, mu1 = 0.f, is1 = 0.f, ms2 = 0.f, mt2 = 0.f, mu2 = 0.f, is2 = 0.f;
        if (CAN_MASK && EPI == EPI_MASK) {
          ms1 = ep.ms1[cc]; mt1 = ep.mt1[cc]; mu1 = ep.mean1[cc]; is1 = ep.invstd1[cc];
          if (my2_t) { ms2 = ep.ms2[cc]; mt2 = ep.mt2[cc]; mu2 = ep.mean2[cc]; is2 = ep.invstd2[cc]; }
        }
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
        float ykeep[16];                                        // the tile's values as stored (bf16 output: rounded, stored in pairs below)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int rl = rbase + (i & 3) + 8 * (i >> 2);
          const bool ok = cok && (full || rl <= last);
          float y = acc[c][i] + bias;
          if (obf) y = bf16_round(y);                           // the sums below see what the stored tensor holds
          if (CAN_MASK && EPI == EPI_MASK) {
            float z = fmaf(ms1, y1v[i], mt1);
            if (my2_t) z += fmaf(ms2, y2v[i], mt2);
            y = (z > 0.f && ok) ? y : 0.f;
            t0 += y;
            t1 = fmaf(y, (y1v[i] - mu1) * is1, t1);
            if (my2_t) t2 = fmaf(y, (y2v[i] - mu2) * is2, t2);
          } else if (CAN_STATS && EPI == EPI_STATS) {
            const float yy = ok ? y : 0.f;
            t0 = fmaf(mw[i], yy, t0);
            t1 = fmaf(mw[i] * yy, yy, t1);
          }
          ykeep[i] = y;
        }
        if (!obf) {
          // the tile's 16 stores of this column tile behind ONE predicate when every row of the tile is valid (all tiles of a launch
          // but its last); per-row predicates (an exec save / restore and a branch each) only there
          char *ob = reinterpret_cast<char *>(out_t);
          const unsigned o_col = 4u * ((unsigned)rbase * (unsigned)ldo + (unsigned)col);
          if (full) {
            if (cok) {
#pragma unroll
              for (int i = 0; i < 16; ++i) *reinterpret_cast<float *>(ob + (o_col + 4u * (unsigned)((i & 3) + 8 * (i >> 2)) * (unsigned)ldo)) = ykeep[i];
            }
          } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int ri = (i & 3) + 8 * (i >> 2);
              if (cok && rbase + ri <= last) *reinterpret_cast<float *>(ob + (o_col + 4u * (unsigned)ri * (unsigned)ldo)) = ykeep[i];
            }
          }
        }
        if (obf) {
          // two bf16 per store: lanes j, j + 1 (adjacent columns) trade one value per row pair (i, i + 1) -- the even lane
          // writes columns (j, j + 1) of row i, the odd lane columns (j - 1, j) of row i + 1: 8 dword stores instead of
          // 16 half-word ones, each instruction covering 64-byte row segments
          const bool odd = lrow & 1;
          const bool pair_ok = (cols % 2 == 0) && (ldo % 2 == 0);          // dword-aligned pairs, both columns in range
          unsigned short *o16 = reinterpret_cast<unsigned short *>(out_t);
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            const int rl0 = rbase + (i & 3) + 8 * (i >> 2), rl1 = rl0 + 1;
            const float send = odd ? ykeep[i] : ykeep[i + 1];
            const float recv = __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(send), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
            if (pair_ok) {
              const int rl = odd ? rl1 : rl0;
              const unsigned lo = __float_as_uint(odd ? recv : ykeep[i]), hi = __float_as_uint(odd ? ykeep[i + 1] : recv);
              if (cok && (full || rl <= last))
                *reinterpret_cast<unsigned *>(o16 + rl * ldo + (col & ~1)) = (lo >> 16) | (hi & 0xffff0000u);
            } else {
              if (cok && (full || rl0 <= last)) o16[rl0 * ldo + col] = (unsigned short)(__float_as_uint(ykeep[i]) >> 16);
              if (cok && (full || rl1 <= last)) o16[rl1 * ldo + col] = (unsigned short)(__float_as_uint(ykeep[i + 1]) >> 16);
            }
          }
        }
        st0[c] += t0; st1[c] += t1; st2[c] += t2;
        if (CAN_STATS && ep.pool_ns == 32) {
          // Max-pool over groups of 32 rows folded into THIS epilogue (round 4): a wave's 32 x 32 accumulator tile IS one group
          // per column (tiles start on multiples of 32 rows), a lane holds 16 of its rows in ascending order, the other 16 sit in
          // lane ^ 32: raw extremes of the stored values and their first positions, resolved by rs_pool_select once BatchNorm's
          // scale is known.  (Through the LDS epilogue -- C tile in LDS, a thread walks a group's 32 rows -- the 524 288 x 32 -> 64
          // layer of the segmentation step ran at 2.2 TB/s: 90 us for 201 MB.)
          float mx = -INFINITY, mn = INFINITY;
          int ax = 0, an = 0;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int k = 4 * lk + (i & 3) + 8 * (i >> 2);
            const float v = ykeep[i];
            if (v > mx) { mx = v; ax = k; }
            if (v < mn) { mn = v; an = k; }
          }
          const float pmx = __shfl_xor(mx, 32, 64), pmn = __shfl_xor(mn, 32, 64);
          const int pax = __shfl_xor(ax, 32, 64), pan = __shfl_xor(an, 32, 64);
          if (pmx > mx || (pmx == mx && pax < ax)) { mx = pmx; ax = pax; }
          if (pmn < mn || (pmn == mn && pan < an)) { mn = pmn; an = pan; }
          if (lk == 0 && cok && r0 + wave_r * 32 < rows) {
            const long long o = ((r0 + wave_r * 32) >> 5) * cols + col;
            ep.pool_max[o] = mx; ep.pool_min[o] = mn; ep.pool_amax[o] = ax; ep.pool_amin[o] = an;
          }
        }
      }
      if (!WS) __syncthreads();   // every wave is past its fragment reads: the next tile may overwrite the staging buffers
                                  // (WS: the stage parity runs across tiles and the chunk barrier orders the overwrite)
    } else {
      __syncthreads();   // all fragment reads of this tile done: the staging buffers become the C tile
      RS_T(5);

      // ---- epilogue.  D[i][j]: j = lane & 31, i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)  ->  Cs[row][col]
      float *Cs = smem;                                         // BM x BN floats (<= 64 KB)
  #pragma unroll
      for (int c = 0; c < CT; ++c)
  #pragma unroll
        for (int i = 0; i < 16; ++i)
          Cs[(wave_r * 32 + (i & 3) + 8 * (i >> 2) + 4 * lk) * BN + (wave_c * CT + c) * 32 + lrow] = acc[c][i];
      __syncthreads();
      float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};   // this tile's column sums
      {
        const int EPI = ep.mode;
        const int col = n0 + e_col;
        float bias[4], ms1[4], mt1[4], mu1[4], is1[4], ms2[4], mt2[4], mu2[4], is2[4];
  #pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool cok = col + e < cols;
          bias[e] = (ep.bias && cok) ? ep.bias[col + e] : 0.f;
          ms1[e] = mt1[e] = mu1[e] = is1[e] = ms2[e] = mt2[e] = mu2[e] = is2[e] = 0.f;
          if (EPI == EPI_MASK && cok) {
            ms1[e] = ep.ms1[col + e]; mt1[e] = ep.mt1[col + e]; mu1[e] = ep.mean1[col + e]; is1[e] = ep.invstd1[col + e];
            if (ep.my2) { ms2[e] = ep.ms2[col + e]; mt2[e] = ep.mt2[col + e]; mu2[e] = ep.mean2[col + e]; is2[e] = ep.invstd2[col + e]; }
          }
        }
        const bool obf = sb_out(MODE >= 0 ? opm_base(MODE) : E.mode);
        float *out_t = tile_base(ep.out, r0 * ep.ldo, obf);     // wave-uniform tile bases
        const float *my1_t = ep.my1 ? tile_base(ep.my1, r0 * ep.ldm1, SB_MASK) : nullptr;
        const float *my2_t = ep.my2 ? tile_base(ep.my2, r0 * ep.ldm2, SB_MASK) : nullptr;
        for (int rl = e_row; rl < BM; rl += E_RPP) {
          if (r0 + rl >= rows || col >= cols) continue;
          const float4 cv = *reinterpret_cast<const float4 *>(Cs + rl * BN + e_col);
          float y[4] = {cv.x + bias[0], cv.y + bias[1], cv.z + bias[2], cv.w + bias[3]};
          if (obf) {
  #pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = bf16_round(y[e]);
          }
          if (EPI == EPI_MASK) {
            float y1[4], y2[4] = {0.f, 0.f, 0.f, 0.f};
            if (ep_vec) {
              ldx<4>(my1_t, (long long)rl * ep.ldm1 + col, SB_MASK, y1);
              if (my2_t) ldx<4>(my2_t, (long long)rl * ep.ldm2 + col, SB_MASK, y2);
              bf16_expand<4>(y1, SB_MASK);
              if (my2_t) bf16_expand<4>(y2, SB_MASK);
            } else {
  #pragma unroll
              for (int e = 0; e < 4; ++e) {
                const bool cok = col + e < cols;
                y1[e] = cok ? ld1x(my1_t, (long long)rl * ep.ldm1 + col + e, SB_MASK) : 0.f;
                if (my2_t) y2[e] = cok ? ld1x(my2_t, (long long)rl * ep.ldm2 + col + e, SB_MASK) : 0.f;
              }
            }
  #pragma unroll
            for (int e = 0; e < 4; ++e) {
              float z = fmaf(ms1[e], y1[e], mt1[e]);
              if (my2_t) z += fmaf(ms2[e], y2[e], mt2[e]);
              y[e] = z > 0.f ? y[e] : 0.f;
              s0[e] += y[e];
              s1[e] = fmaf(y[e], (y1[e] - mu1[e]) * is1[e], s1[e]);
              if (my2_t) s2[e] = fmaf(y[e], (y2[e] - mu2[e]) * is2[e], s2[e]);
            }
          } else if (EPI == EPI_STATS) {
            const float mw = ep.row_mult ? ep.row_mult[r0 + rl] : 1.f;      // copies this compacted row stands for
  #pragma unroll
            for (int e = 0; e < 4; ++e) { s0[e] = fmaf(mw, y[e], s0[e]); s1[e] = fmaf(mw * y[e], y[e], s1[e]); }
          }
          if (obf) {
            unsigned short *o16 = reinterpret_cast<unsigned short *>(out_t) + (long long)rl * ep.ldo + col;
            if (ep_vec) {
              *reinterpret_cast<uint2 *>(o16) = make_uint2((__float_as_uint(y[0]) >> 16) | (__float_as_uint(y[1]) & 0xffff0000u),
                                                           (__float_as_uint(y[2]) >> 16) | (__float_as_uint(y[3]) & 0xffff0000u));
            } else {
  #pragma unroll
              for (int e = 0; e < 4; ++e)
                if (col + e < cols) o16[e] = (unsigned short)(__float_as_uint(y[e]) >> 16);
            }
          } else if (ep_vec) {
            *reinterpret_cast<float4 *>(out_t + (long long)rl * ep.ldo + col) = make_float4(y[0], y[1], y[2], y[3]);
          } else {
  #pragma unroll
            for (int e = 0; e < 4; ++e)
              if (col + e < cols) out_t[(long long)rl * ep.ldo + col + e] = y[e];
          }
        }
      }
      if (ep.pool_ns > 0) {
        // Max-pool over nsample folded into the producing GEMM: BatchNorm's scale is not known yet (its
        // statistics are still being summed), so keep the raw extremes of y per (group, column) — the
        // pooled activation is relu(scale * (scale >= 0 ? max : min) + shift), resolved by rs_pool_select.
        constexpr int TPC = GM_THREADS / BN, RPT = BM / TPC;        // threads per column, rows per thread
        const int c = tid % BN, part = tid / BN, col = n0 + c;
        if (col < cols) {
          const float bb = ep.bias ? ep.bias[col] : 0.f;
          for (int g0 = part * RPT; g0 < (part + 1) * RPT && r0 + g0 < rows; g0 += ep.pool_ns) {
            float mx = -INFINITY, mn = INFINITY;
            int ax = 0, an = 0;
            for (int k = 0; k < ep.pool_ns; ++k) {
              float v = Cs[(g0 + k) * BN + c] + bb;
              if (sb_out(MODE >= 0 ? opm_base(MODE) : E.mode)) v = bf16_round(v);       // pool what the stored tensor holds
              if (v > mx) { mx = v; ax = k; }
              if (v < mn) { mn = v; an = k; }
            }
            const long long o = ((r0 + g0) / ep.pool_ns) * cols + col;
            ep.pool_max[o] = mx; ep.pool_min[o] = mn; ep.pool_amax[o] = ax; ep.pool_amin[o] = an;
          }
        }
      }
      __syncthreads();   // C tile consumed before the statistics / the next tile's staging overwrite it
      if (ep.mode != EPI_STORE) {
        // Column sums of this tile -> fp64 partial row of this workgroup.  The sums leave the registers here, per tile,
        // instead of riding along in 24 VGPRs of fp64 accumulators: with those live across the K loop the dual-operand
        // instances spilled operand pointers to scratch, and a scratch reload in front of a prefetch waits for every older
        // global load (in-order vmcnt) -- the prefetch latency was exposed once per chunk.
        double *red = reinterpret_cast<double *>(smem);         // 3 x E_RPP x BN doubles = 24 KB
        const int nstat = (ep.mode == EPI_MASK && ep.my2) ? 3 : 2;
  #pragma unroll
        for (int e = 0; e < 4; ++e) {                           // columns beyond `cols` accumulated zeros only
          red[(0 * E_RPP + e_row) * BN + e_col + e] = (double)s0[e];
          red[(1 * E_RPP + e_row) * BN + e_col + e] = (double)s1[e];
          if (nstat == 3) red[(2 * E_RPP + e_row) * BN + e_col + e] = (double)s2[e];
        }
        __syncthreads();
        if (tid < BN && n0 + tid < cols) {
          for (int sidx = 0; sidx < nstat; ++sidx) {
            double t = 0.0;
            for (int p = 0; p < E_RPP; ++p) t += red[(sidx * E_RPP + p) * BN + tid];
            double *dst = ep.partial + ((long long)blockIdx.x * nstat + sidx) * cols + n0 + tid;
            *dst = (tile == (long long)blockIdx.x) ? t : *dst + t;    // first tile of this workgroup stores, later ones add
          }
        }
        __syncthreads();
      }
    }
    RS_T(6);
  }
  }
  if (WS && !loader && (ws_total & 1)) __syncthreads();     // the loaders' padding chunk (two chunks per trip)
#ifdef RS_EXP_TIMING
  if ((tid == 0 || (WS && tid == GM_THREADS)) && ep.pool_amax) {      // WS: the first loader wave reports 4096 rows further down
    long long *o = reinterpret_cast<long long *>(ep.pool_amax) + ((long long)blockIdx.y * gridDim.x + blockIdx.x + (loader ? 4096 : 0)) * 10;
    for (int i = 0; i < 7; ++i) o[i] = tacc[i];
    o[7] = clock64() - tstart;
    o[8] = tacc[7];
    o[9] = tiles;
  }
#endif

  if (direct) {
    if (ep.mode != EPI_STORE && (long long)blockIdx.x < tiles) {
      // lane-local column sums -> this workgroup's fp64 partial row: NSLOT contributions per column (row waves x 2 lane halves)
      double *red = reinterpret_cast<double *>(smem);         // [stat][NSLOT][BN] doubles <= 24 KB (staging is idle: barrier below)
      const int nstat = (ep.mode == EPI_MASK && ep.my2) ? 3 : 2;
      __syncthreads();
      if (!loader) {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const int cl = (wave_c * CT + c) * 32 + lrow, slot = wave_r * 2 + lk;
          red[(0 * NSLOT + slot) * BN + cl] = (double)st0[c];
          red[(1 * NSLOT + slot) * BN + cl] = (double)st1[c];
          if (nstat == 3) red[(2 * NSLOT + slot) * BN + cl] = (double)st2[c];
        }
      }
      __syncthreads();
      if (tid < BN && n0 + tid < cols)
        for (int sidx = 0; sidx < nstat; ++sidx) {
          double t = (red[(sidx * NSLOT + 0) * BN + tid] + red[(sidx * NSLOT + 1) * BN + tid]) +
                     (red[(sidx * NSLOT + 2) * BN + tid] + red[(sidx * NSLOT + 3) * BN + tid]);
          if (NSLOT == 8)
            t += (red[(sidx * NSLOT + 4) * BN + tid] + red[(sidx * NSLOT + 5) * BN + tid]) +
                 (red[(sidx * NSLOT + 6) * BN + tid] + red[(sidx * NSLOT + 7) * BN + tid]);
          ep.partial[((long long)blockIdx.x * nstat + sidx) * cols + n0 + tid] = t;
        }
    }
  }
  if (ep.mode != EPI_STORE && tid < BN && n0 + tid < cols) {
    // the finalize kernel sums `partial_blocks` rows: rows no workgroup owns read as zero (no memset launch);
    // a workgroup without a tile zeroes its own row too
    const int nstat = (ep.mode == EPI_MASK && ep.my2) ? 3 : 2;
    for (int sidx = 0; sidx < nstat; ++sidx) {
      if ((long long)blockIdx.x >= tiles) ep.partial[((long long)blockIdx.x * nstat + sidx) * cols + n0 + tid] = 0.0;
      for (int pb = blockIdx.x + gridDim.x; pb < ep.partial_blocks; pb += gridDim.x)
        ep.partial[((long long)pb * nstat + sidx) * cols + n0 + tid] = 0.0;
    }
  }
}

// ---- weight gradient: dw[n][k] = sum_r P[r][n] * Q[r][k] ------------------------------------------
// Workgroup = (row slab, output block of (WN*TN*32) x (WK*TK*32)); the reduction index (rows) is the
// MFMA k dimension, so both operand tiles sit row-major in LDS and fragment reads are consecutive.
// Same register-prefetch / double-buffer pipeline as the row GEMM.
constexpr int WG_BR = 32;   // rows per pipeline stage

//
// BF = true (rs_mlp_wgrad_bf16): v_mfma_f32_32x32x16_bf16.  A lane needs 8 values of ONE column along the reduction
// index (rows), so a stage is stored as 16 row-PAIRS t of [column] dwords, dword = bf16(row 2t) | bf16(row 2t+1) << 16:
// a thread loads rows 2t and 2t+1 of its columns (t = u * RPP + its row slot), rounds after the fp32 prologue and
// writes V consecutive dwords; the fragment of MFMA step s for lane (col, g) is the dwords t = 4 (2 s + g) .. + 3 of its
// column (consecutive lanes, consecutive addresses).  Which rows meet in which k slot is free as long as P and Q agree.
template <int WN, int WK, int TN, int TK, int VP, int VQ, int PM, int QM, bool BF>
__global__ void __launch_bounds__(GM_THREADS, 2)
wgrad_kernel(long long rows_arg, const int *__restrict__ rows_dev, int ncols, int kcols, RowOperand P, RowOperand Q,
             float *__restrict__ partial) {
  // Row stages (32 rows) are dealt round-robin to the gridDim.x row workgroups: at any moment the whole grid reads
  // one contiguous window of the operands (gridDim.x * 32 rows), which keeps DRAM pages and TLB entries hot.
  // (Contiguous per-workgroup slabs had 512 streams 0.5 MB apart: 1.2 TB/s; the partial sums do not care which
  // rows they hold, and the order inside a partial stays fixed, so results remain deterministic.)
  const long long rows = rows_dev ? min(rows_arg, (long long)*rows_dev) : rows_arg;
  constexpr int BNN = WN * TN * 32, BKK = WK * TK * 32;
  constexpr int P_VECS = WG_BR * BNN / VP / GM_THREADS, Q_VECS = (WG_BR * BKK / VQ + GM_THREADS - 1) / GM_THREADS;
  constexpr int P_TPR = BNN / VP, Q_TPR = BKK / VQ;           // threads per tile row
  constexpr int P_RPP = GM_THREADS / P_TPR;                   // rows per pass (P_TPR <= 256 always)
  constexpr int Q_RPP = (GM_THREADS / Q_TPR) > 0 ? (GM_THREADS / Q_TPR) : 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int PARTS = (BF && RS_SPLIT) ? 3 : 1;         // unit 4: a part = 16 row pairs x columns dwords
  constexpr int PART_P = (WG_BR / 2) * BNN, PART_Q = (WG_BR / 2) * BKK;
  constexpr int P_STAGE = PARTS == 3 ? 3 * PART_P : WG_BR * BNN, Q_STAGE = PARTS == 3 ? 3 * PART_Q : WG_BR * BKK;
  // unit 4, 128 x 128 output block: two stages of three parts would be 96 KB -- one workgroup per CU -- so it runs on ONE stage and
  // a second barrier per 32 rows
  constexpr bool ONE_STAGE = PARTS == 3 && BNN + BKK > 192;
  float *Ps0 = smem, *Ps1 = ONE_STAGE ? smem : smem + P_STAGE;
  float *Qs0 = smem + (ONE_STAGE ? 1 : 2) * P_STAGE, *Qs1 = ONE_STAGE ? Qs0 : Qs0 + Q_STAGE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave / WK, wk = wave % WK;
  const int n0 = blockIdx.y * BNN, k0 = blockIdx.z * BKK;
  const int lcol = lane & 31, lr = lane >> 5;
  const long long rbeg = (long long)blockIdx.x * WG_BR, rstep = (long long)gridDim.x * WG_BR;
  const long long rend = rows;
  const int p_c = (tid % P_TPR) * VP, p_r = tid / P_TPR;
  const int q_c = (tid % Q_TPR) * VQ, q_r = tid / Q_TPR;
  const bool q_active = tid < Q_TPR * Q_RPP;                  // Q tiles narrower than 256 vectors per pass
  static_assert(!BF || (P_VECS % 2 == 0 && Q_VECS % 2 == 0 && VP >= 2 && VQ >= 2), "bf16 staging pairs the rows a thread owns");
  // local row of a thread's p-th vector: fp32 p * RPP + slot;  bf16 2 * ((p >> 1) * RPP + slot) + (p & 1)
  auto p_row = [&](int p) { return BF ? 2 * ((p >> 1) * P_RPP + p_r) + (p & 1) : p * P_RPP + p_r; };
  auto q_row = [&](int p) { return BF ? 2 * ((p >> 1) * Q_RPP + q_r) + (p & 1) : p * Q_RPP + q_r; };

  f32x16 acc[TN][TK];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TK; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  RawVec<VP> praw[P_VECS];
  RawVec<VQ> qraw[Q_VECS];
  ColCoef<VP> pcoef;
  ColCoef<VQ> qcoef;
  const bool pc_ok = n0 + p_c < ncols, qc_ok = q_active && (k0 + q_c < kcols);
  op_coef<VP, PM>(P, n0 + p_c, pc_ok, pcoef);                     // this thread's columns never change
  op_coef<VQ, QM>(Q, k0 + q_c, qc_ok, qcoef);

  // part 0..3: a quarter of the next stage's loads (issued between the MFMA steps, see gemm_rows_kernel); < 0: all
  // Loads are never predicated (as in the row GEMM): rows beyond the slab's end and columns beyond the matrix are CLAMPED to valid
  // ones and the values zeroed when they are committed -- a predicated load is an exec-mask branch per vector, and the compiler
  // can neither cluster such loads nor count them in flight.
  auto prefetch = [&](long long r0, int part) {
    const int rlast = (int)min((long long)WG_BR - 1, rend - 1 - r0);
    const int pc = pc_ok ? n0 + p_c : n0, qc = qc_ok ? k0 + q_c : k0;
    // rows with an index of their own (ragged pooled groups: group / slot; multiplicities): ALL of the stage's row metadata first, then the
    // gathers -- one dependent round trip per stage instead of one per operand vector (round 6: the ISA had `load grp, load slot,
    // s_waitcnt, three gathers` four times in a row)
    constexpr bool P_META = PM >= 0 && (opm_base(PM) == OPM_POOLED || PM == OPM_AFF2);
    RowMeta pmeta[P_VECS];
    if constexpr (P_META) {
#pragma unroll
      for (int p = 0; p < P_VECS; ++p) op_row_meta<PM>(P, r0, min(p_row(p), rlast), pmeta[p]);
    }
#pragma unroll
    for (int p = 0; p < P_VECS; ++p) {
      if (part >= 0 && (p * 4) / P_VECS != part) continue;
      op_load<VP, PM>(P, r0, min(p_row(p), rlast), pc, true, praw[p], P_META ? &pmeta[p] : nullptr);
    }
#pragma unroll
    for (int p = 0; p < Q_VECS; ++p) {
      if (part >= 0 && (p * 4) / Q_VECS != part) continue;
      op_load<VQ, QM>(Q, r0, min(q_row(p), rlast), qc, true, qraw[p]);
    }
  };
  auto commit = [&](float *Ps, float *Qs, long long r0) {
    if constexpr (BF) {
#pragma unroll
      for (int u = 0; u < P_VECS / 2; ++u) {
        const int t = u * P_RPP + p_r;
        __attribute__((aligned(16))) float v0[VP], v1[VP], d[VP];
        op_finish<VP, PM>(P, pcoef, praw[2 * u], r0 + 2 * t, pc_ok && r0 + 2 * t < rend, v0);
        op_finish<VP, PM>(P, pcoef, praw[2 * u + 1], r0 + 2 * t + 1, pc_ok && r0 + 2 * t + 1 < rend, v1);
        if constexpr (PARTS == 3) {
          __attribute__((aligned(16))) float d3[3][VP];
#pragma unroll
          for (int i = 0; i < VP; ++i) {
            float e[3];
            split_bf16(v0[i], v1[i], e);
            d3[0][i] = e[0]; d3[1][i] = e[1]; d3[2][i] = e[2];
          }
#pragma unroll
          for (int q = 0; q < 3; ++q)
            *reinterpret_cast<typename VecT<VP>::F *>(Ps + q * PART_P + t * BNN + p_c) = *reinterpret_cast<typename VecT<VP>::F *>(d3[q]);
          continue;
        }
#pragma unroll
        for (int i = 0; i < VP; ++i) d[i] = pack_bf16(v0[i], v1[i]);
        *reinterpret_cast<typename VecT<VP>::F *>(Ps + t * BNN + p_c) = *reinterpret_cast<typename VecT<VP>::F *>(d);
      }
#pragma unroll
      for (int u = 0; u < Q_VECS / 2; ++u) {
        const int t = u * Q_RPP + q_r;
        if (q_active && t < WG_BR / 2) {
          __attribute__((aligned(16))) float v0[VQ], v1[VQ], d[VQ];
          op_finish<VQ, QM>(Q, qcoef, qraw[2 * u], r0 + 2 * t, qc_ok && r0 + 2 * t < rend, v0);
          op_finish<VQ, QM>(Q, qcoef, qraw[2 * u + 1], r0 + 2 * t + 1, qc_ok && r0 + 2 * t + 1 < rend, v1);
          if constexpr (PARTS == 3) {
            __attribute__((aligned(16))) float d3[3][VQ];
#pragma unroll
            for (int i = 0; i < VQ; ++i) {
              float e[3];
              split_bf16(v0[i], v1[i], e);
              d3[0][i] = e[0]; d3[1][i] = e[1]; d3[2][i] = e[2];
            }
#pragma unroll
            for (int q = 0; q < 3; ++q)
              *reinterpret_cast<typename VecT<VQ>::F *>(Qs + q * PART_Q + t * BKK + q_c) = *reinterpret_cast<typename VecT<VQ>::F *>(d3[q]);
            continue;
          }
#pragma unroll
          for (int i = 0; i < VQ; ++i) d[i] = pack_bf16(v0[i], v1[i]);
          *reinterpret_cast<typename VecT<VQ>::F *>(Qs + t * BKK + q_c) = *reinterpret_cast<typename VecT<VQ>::F *>(d);
        }
      }
      return;
    }
#pragma unroll
    for (int p = 0; p < P_VECS; ++p) {
      const int rl = p * P_RPP + p_r;
      const long long r = r0 + rl;
      __attribute__((aligned(16))) float v[VP];
      op_finish<VP, PM>(P, pcoef, praw[p], r, pc_ok && r < rend, v);
      *reinterpret_cast<typename VecT<VP>::F *>(Ps + rl * BNN + p_c) = *reinterpret_cast<typename VecT<VP>::F *>(v);
    }
#pragma unroll
    for (int p = 0; p < Q_VECS; ++p) {
      const int rl = p * Q_RPP + q_r;
      const long long r = r0 + rl;
      if (q_active && rl < WG_BR) {
        __attribute__((aligned(16))) float v[VQ];
        op_finish<VQ, QM>(Q, qcoef, qraw[p], r, qc_ok && r < rend, v);
        *reinterpret_cast<typename VecT<VQ>::F *>(Qs + rl * BKK + q_c) = *reinterpret_cast<typename VecT<VQ>::F *>(v);
      }
    }
  };

#ifdef RS_EXP_TIMING
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = clock64();
  const long long tstart = tlast;
#endif
  if (rbeg < rend) prefetch(rbeg, -1);
  RS_T(0);
  int it = 0;
  for (long long r0 = rbeg; r0 < rend; r0 += rstep, ++it) {
    float *Ps = (it & 1) ? Ps1 : Ps0;
    float *Qs = (it & 1) ? Qs1 : Qs0;
    if (ONE_STAGE && it > 0) __syncthreads();                 // every wave is done with the fragments of the previous 32 rows
    commit(Ps, Qs, r0);
    RS_T(1);
    __syncthreads();
    RS_T(2);
    if (r0 + rstep < rend) prefetch(r0 + rstep, -1);
    RS_T(3);
    if constexpr (PARTS == 3) {
      // per step: the three parts of every P / Q tile column, then 6 x TN x TK MFMAs (smallest terms first)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const float *pp = Ps + (4 * (2 * st + lr)) * BNN + wn * TN * 32 + lcol;
        const float *qp = Qs + (4 * (2 * st + lr)) * BKK + wk * TK * 32 + lcol;
        float4 pa[3][TN], qb[3][TK];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
          for (int a = 0; a < TN; ++a) {
            const float *x = pp + q * PART_P + a * 32;
            pa[q][a] = make_float4(x[0], x[BNN], x[2 * BNN], x[3 * BNN]);
          }
#pragma unroll
          for (int b = 0; b < TK; ++b) {
            const float *x = qp + q * PART_Q + b * 32;
            qb[q][b] = make_float4(x[0], x[BKK], x[2 * BKK], x[3 * BKK]);
          }
        }
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};      // lh, hl, mm, mh, hm, hh
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TK; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa[TA[t]][a]),
                                                                 __builtin_bit_cast(bf16x8, qb[TB[t]][b]), acc[a][b], 0, 0, 0);
      }
    } else if constexpr (BF) {
      // lane (col, g = lr), step st: dwords t = 4 (2 st + g) .. + 3 of its column; all reads first, then 2 x TN x TK MFMAs
      float4 pa[2][TN], qb[2][TK];
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const float *pp = Ps + (4 * (2 * st + lr)) * BNN + wn * TN * 32 + lcol;
        const float *qp = Qs + (4 * (2 * st + lr)) * BKK + wk * TK * 32 + lcol;
#pragma unroll
        for (int a = 0; a < TN; ++a)
          pa[st][a] = make_float4(pp[a * 32], pp[BNN + a * 32], pp[2 * BNN + a * 32], pp[3 * BNN + a * 32]);
#pragma unroll
        for (int b = 0; b < TK; ++b)
          qb[st][b] = make_float4(qp[b * 32], qp[BKK + b * 32], qp[2 * BKK + b * 32], qp[3 * BKK + b * 32]);
      }
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TK; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa[st][a]),
                                                               __builtin_bit_cast(bf16x8, qb[st][b]), acc[a][b], 0, 0, 0);
    } else {
    // fragment reads of step s+1 are issued before the MFMAs of step s (two register sets, fully unrolled):
    // the ~100-cycle LDS latency stays under the matrix pipe instead of in front of every 4 MFMAs
    const float *pp = Ps + lr * BNN + wn * TN * 32 + lcol;
    const float *qp = Qs + lr * BKK + wk * TK * 32 + lcol;
    float pa[2][TN], qb[2][TK];
#pragma unroll
    for (int a = 0; a < TN; ++a) pa[0][a] = pp[a * 32];
#pragma unroll
    for (int b = 0; b < TK; ++b) qb[0][b] = qp[b * 32];
#pragma unroll
    for (int s = 0; s < WG_BR / 2; ++s) {
      if (s + 1 < WG_BR / 2) {
#pragma unroll
        for (int a = 0; a < TN; ++a) pa[(s + 1) & 1][a] = pp[(2 * s + 2) * BNN + a * 32];
#pragma unroll
        for (int b = 0; b < TK; ++b) qb[(s + 1) & 1][b] = qp[(2 * s + 2) * BKK + b * 32];
      }
      __builtin_amdgcn_sched_barrier(0);      // keep the reads above this step's MFMAs (the scheduler sinks them otherwise)
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TK; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[s & 1][a], qb[s & 1][b], acc[a][b], 0, 0, 0);
    }
    }
    RS_T(4);
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
#ifdef RS_EXP_TIMING
  RS_T(6);
  if (tid == 0 && P.t2) {      // experiment build: the (unused) t2 slot of the P operand carries the debug buffer
    long long *o = reinterpret_cast<long long *>(const_cast<float *>(P.t2)) +
                   (((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 10;
    for (int i = 0; i < 7; ++i) o[i] = tacc[i];
    o[7] = clock64() - tstart;
    o[8] = it;
    o[9] = gridDim.x * gridDim.y * gridDim.z;
  }
#endif
}

constexpr int WS_KP = 16;          // reduction width of the narrow (streaming) kernels

// ---- narrow row GEMM: kdim <= 16, plain operand (the position / first feature branches of every stack) -------------
// out[r][n] = bias[n] + sum_k x[r][k] * w[n][k] reads 4 * kdim bytes and writes 4 * cols bytes per row: pure streaming.
// Thread (tn, tr) keeps the weights of its 4 output columns in registers (4 x 16), walks the rows tr, tr + RG, ... of the
// workgroup's window, reads the <= 16 operand values of a row (same address for the lanes sharing the row: broadcast),
// writes one float4 and keeps the BatchNorm sums of its columns; the row groups are combined in fp64 at the end, in the
// same partial layout as gemm_rows_kernel.  The 32-deep MFMA chunk of the general kernel would be >= half padding here.
template <int NB>
__global__ void __launch_bounds__(GM_THREADS)
gemm_small_kernel(long long rows_arg, const int *__restrict__ rows_dev, int kdim, int cols, RowOperand E,
                  const float *__restrict__ w, int ldw, Epilogue ep) {
  constexpr int TPN = NB / 4, RG = GM_THREADS / TPN;
  const long long rows = rows_dev ? min(rows_arg, (long long)*rows_dev) : rows_arg;
  __shared__ double red[GM_THREADS / (NB / 4)][NB];
  const int tid = threadIdx.x, tn = tid % TPN, tr = tid / TPN;
  const int n0 = blockIdx.y * NB + tn * 4;
  float wr[4][WS_KP], bias[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool nok = n0 + i < cols;
    bias[i] = (ep.bias && nok) ? ep.bias[n0 + i] : 0.f;
#pragma unroll
    for (int k = 0; k < WS_KP; ++k) wr[i][k] = (nok && k < kdim) ? w[(long long)(n0 + i) * ldw + k] : 0.f;
  }
  const bool stats = ep.mode == EPI_STATS;
  const bool vec_ok = (((uintptr_t)ep.out) % 16 == 0) && (ep.ldo % 4 == 0) && (n0 + 3 < cols);
  float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
  const long long step = (long long)gridDim.x * RG;
  auto load_row = [&](long long r, float (&x)[WS_KP]) {
    const long long rc = min(r, rows - 1);                        // clamped: never out of bounds, discarded below
    const float *src = E.a + rc * E.lda;
#pragma unroll
    for (int k = 0; k < WS_KP; ++k) x[k] = k < kdim ? src[k] : 0.f;
  };
  auto do_row = [&](long long r, const float (&x)[WS_KP]) {
    if (r >= rows) return;
    float y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < WS_KP; ++k) t = fmaf(x[k], wr[i][k], t);
      y[i] = t + bias[i];
    }
    if (sb_out(OPM_ID)) {                                         // bf16 storage (always an ID operand here): round first, sum what was stored
      unsigned short *o16 = reinterpret_cast<unsigned short *>(ep.out) + r * ep.ldo + n0;
#pragma unroll
      for (int i = 0; i < 4; ++i) y[i] = bf16_round(y[i]);
      if (vec_ok) *reinterpret_cast<uint2 *>(o16) = make_uint2((__float_as_uint(y[0]) >> 16) | (__float_as_uint(y[1]) & 0xffff0000u),
                                                               (__float_as_uint(y[2]) >> 16) | (__float_as_uint(y[3]) & 0xffff0000u));
      else {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (n0 + i < cols) o16[i] = (unsigned short)(__float_as_uint(y[i]) >> 16);
      }
    } else {
    float *o = ep.out + r * ep.ldo + n0;
    if (vec_ok) *reinterpret_cast<float4 *>(o) = make_float4(y[0], y[1], y[2], y[3]);
    else {
#pragma unroll
      for (int i = 0; i < 4; ++i) if (n0 + i < cols) o[i] = y[i];
    }
    }
    if (stats) {
      const float mw = ep.row_mult ? ep.row_mult[r] : 1.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { s0[i] = fmaf(mw, y[i], s0[i]); s1[i] = fmaf(mw * y[i], y[i], s1[i]); }
    }
  };
  if (rows > 0) {
    float xa[WS_KP], xb[WS_KP];
    long long r = (long long)blockIdx.x * RG + tr;
    load_row(r, xa);
    for (; r < rows; r += 2 * step) {
      load_row(r + step, xb);
      do_row(r, xa);
      load_row(r + 2 * step, xa);
      do_row(r + step, xb);
    }
  }
  if (stats) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < 4; ++i) red[tr][tn * 4 + i] = (double)(s == 0 ? s0[i] : s1[i]);
      __syncthreads();
      if (tid < NB) {
        double t = 0.0;
        for (int g = 0; g < RG; ++g) t += red[g][tid];
        const int col = blockIdx.y * NB + tid;
        if (col < cols) {
          ep.partial[((long long)blockIdx.x * 2 + s) * cols + col] = t;
          for (int pb = blockIdx.x + gridDim.x; pb < ep.partial_blocks; pb += gridDim.x)
            ep.partial[((long long)pb * 2 + s) * cols + col] = 0.0;
        }
      }
      __syncthreads();
    }
  }
}

// ---- narrow row GEMM on the matrix pipe, no LDS (round 3): kdim <= 16, plain operand, rows on 16- or 8-byte boundaries ---------
// The first-layer branches of every stack (3 / 6 / 10 input channels -> 32 ... 128 columns) move one wide tensor OUT and almost
// nothing in: 24.6 MB at 48 k x 128 is 3 us of HBM time.  Through the tiled kernel they took 13-23 us -- 1.5 tiles per workgroup,
// each a chain of exposed round trips (operand prefetch -> LDS commit -> barrier -> 8 MFMAs -> row-multiplicity loads -> a
// chip-wide store burst) that 2 workgroups per CU cannot overlap.  Here a WAVE owns 32 rows at a time: ONE 16-byte load per lane is
// the whole operand of the block (lane (r, h) reads x[r][4 h .. 4 h + 3] (+ 8 .. 11 for kdim > 8): v_mfma_f32_32x32x2_f32 is free in
// the order of the k's as long as both operands agree, so half-wave h feeds k = 4 h + i to step i), the weights of the wave's
// column tiles live in registers for the whole launch, the next block's operand is requested before this block's MFMAs, and the
// 32 x cols block leaves straight from the accumulators (128-byte row segments) with the BatchNorm sums lane-local in fp32 across
// all blocks of the wave -- no LDS, no barrier until the one fixed-order reduction at the end.  4 waves per SIMD (<= 128 VGPRs).
template <int CT, int KL>      // CT column tiles of 32 (cols <= 32 CT), KL float4 per lane and row (kdim <= 8 KL)
__global__ void __launch_bounds__(GM_THREADS, CT > 2 ? 3 : 4)     // 128 columns: ~170 VGPRs (3 waves per SIMD), else <= 128 (4)
gemm_narrow_kernel(long long rows_arg, const int *__restrict__ rows_dev, int kdim, int cols, RowOperand E,
                   const float *__restrict__ w, int ldw, Epilogue ep) {
  const long long rows = rows_dev ? min(rows_arg, (long long)*rows_dev) : rows_arg;
  __shared__ double red[2][GM_THREADS / 32][CT * 32];          // [stat][wave x lane half][column]
  const int tid = threadIdx.x, lane = tid & 63, lrow = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform for the compiler too: block bases and row predicates in SGPRs
                                                                   // (as a VGPR value it made every address of the block a 64-bit per-lane computation)
  // weights + bias of this lane's columns (w is n-major, zero in [kdim, ldw), ldw >= 8 KL: checked by the launcher)
  float4 wq[CT][KL];
  float bias[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const int col = c * 32 + lrow;
    const bool cok = col < cols;
    bias[c] = (ep.bias && cok) ? ep.bias[col] : 0.f;
#pragma unroll
    for (int j = 0; j < KL; ++j)      // ldw % 4 == 0: a lane's four k are inside the row or all beyond it
      wq[c][j] = (cok && 8 * j + 4 * h + 4 <= ldw) ? *reinterpret_cast<const float4 *>(w + (long long)col * ldw + 8 * j + 4 * h)
                                                   : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const bool stats = ep.mode == EPI_STATS;
  float st0[CT], st1[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) { st0[c] = 0.f; st1[c] = 0.f; }
  const long long nblk = (rows + 31) >> 5, bstep = (long long)gridDim.x * (GM_THREADS / 64);
  const bool a16 = (((uintptr_t)E.a) % 16 == 0) && (E.lda % 4 == 0);     // else 8-byte aligned rows (checked by the launcher)
  auto load_a = [&](long long blk, float4 (&a)[KL]) {
    const long long r = min(blk * 32 + lrow, rows - 1);              // clamped: always in bounds, the row is discarded below
    const float *src = E.a + r * E.lda + 4 * h;
#pragma unroll
    for (int j = 0; j < KL; ++j) {
      const int k0 = 8 * j + 4 * h;      // channels [kdim, ...) of the row belong to other tensors / padding: never multiplied
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k0 < kdim) {
        if (r == rows - 1) {                 // the last row: a vector load may run past the end of the tensor
          v.x = src[8 * j];
          if (k0 + 1 < kdim) v.y = src[8 * j + 1];
          if (k0 + 2 < kdim) v.z = src[8 * j + 2];
          if (k0 + 3 < kdim) v.w = src[8 * j + 3];
        } else if (a16) {
          v = *reinterpret_cast<const float4 *>(src + 8 * j);
        } else {                             // rows on 8-byte boundaries only (a column slice starting at channel 6 of the
          const float2 lo = *reinterpret_cast<const float2 *>(src + 8 * j);      // compacted classification rows; the
          float2 hi = make_float2(0.f, 0.f);                                     // constructor's 10-float rows)
          if (k0 + 2 < kdim) hi = *reinterpret_cast<const float2 *>(src + 8 * j + 2);
          v = make_float4(lo.x, lo.y, hi.x, hi.y);
        }
      }
      v.x = k0 + 0 < kdim ? v.x : 0.f; v.y = k0 + 1 < kdim ? v.y : 0.f; v.z = k0 + 2 < kdim ? v.z : 0.f; v.w = k0 + 3 < kdim ? v.w : 0.f;
      a[j] = v;
    }
  };
  long long blk = (long long)blockIdx.x * (GM_THREADS / 64) + wave;
  float4 a[KL], an[KL];
  if (blk < nblk) load_a(blk, a);
  for (; blk < nblk; blk += bstep) {
    if (blk + bstep < nblk) load_a(blk + bstep, an);
    const long long r0 = blk * 32;
    // D[i][j]: column j = lane & 31, row (reg & 3) + 8 (reg >> 2) + 4 h
    const bool full = r0 + 32 <= rows;
    int rbase = 4 * h;
    asm volatile("" : "+v"(rbase));                             // opaque per block: the 16 row offsets are recomputed here, not hoisted
                                                                // out of the block loop into ~100 long-lived VGPRs (as in the tiled kernel)
    // Addresses: a wave-uniform base (SGPRs) + an unsigned 32-bit byte offset per lane -- the form global loads / stores take
    // without a 64-bit VGPR pair; row validity is one 32-bit compare against the block's last valid row.  (With 64-bit per-lane
    // row arithmetic every one of the 32-64 stores of a block carried ~17 instructions, 8 MFMAs carried ~1 100.)
    const int last = (int)min(31LL, rows - 1 - r0);
    const char *mw_t = reinterpret_cast<const char *>(ep.row_mult ? ep.row_mult + r0 : nullptr);
    float mw[16];
    if (stats) {
#pragma unroll
      for (int i = 0; i < 16; ++i) mw[i] = 1.f;
      if (mw_t) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {                              // unconditional: a row beyond the end reads the last valid one (unused)
          const int rl = min(rbase + (i & 3) + 8 * (i >> 2), last);
          mw[i] = *reinterpret_cast<const float *>(mw_t + 4u * (unsigned)rl);
        }
      }
    }
    char *out_t = reinterpret_cast<char *>(ep.out + r0 * ep.ldo);
    const unsigned ldo = 4u * (unsigned)ep.ldo;
    unsigned o_out = (unsigned)rbase * ldo;
    asm volatile("" : "+v"(o_out));
    // column tiles two at a time (32 accumulator registers live): with all four the 128-column instances spilled at 128 VGPRs.
    // (Tried: a 4 x 4 transpose inside the quads by DPP and 16-byte stores, 4 per column tile instead of 16 dword stores --
    //  48 234 x 6 -> 128: 18.7 against 12.1 us; the rows are 128-byte segments either way and the shuffles cost more than the stores.)
    constexpr int CP = CT > 2 ? 2 : CT;
#pragma unroll
    for (int cb = 0; cb < CT; cb += CP) {
      f32x16 acc[CP];
#pragma unroll
      for (int c = 0; c < CP; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
#pragma unroll
      for (int j = 0; j < KL; ++j) {
        const float a4[4] = {a[j].x, a[j].y, a[j].z, a[j].w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < CP; ++c) {
            const float4 wv = wq[cb + c][j];
            const float b = i == 0 ? wv.x : (i == 1 ? wv.y : (i == 2 ? wv.z : wv.w));
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i], b, acc[c], 0, 0, 0);
          }
      }
#pragma unroll
      for (int c = 0; c < CP; ++c) {
        const int col = (cb + c) * 32 + lrow;
        const bool cok = col < cols;
        float t0 = 0.f, t1 = 0.f;
        float y[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) y[i] = acc[c][i] + bias[cb + c];
        const unsigned o_col = o_out + 4u * (unsigned)col;
        if (full) {                                                 // all 32 rows valid (every block of a launch but its last): one
          if (cok) {                                                // predicate around the 16 stores instead of one each
#pragma unroll
            for (int i = 0; i < 16; ++i) *reinterpret_cast<float *>(out_t + (o_col + (unsigned)((i & 3) + 8 * (i >> 2)) * ldo)) = y[i];
          }
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int ri = (i & 3) + 8 * (i >> 2);
            if (cok && rbase + ri <= last) *reinterpret_cast<float *>(out_t + (o_col + (unsigned)ri * ldo)) = y[i];
          }
        }
        if (stats) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const bool ok = cok && (full || rbase + (i & 3) + 8 * (i >> 2) <= last);
            const float yy = ok ? y[i] : 0.f;
            t0 = fmaf(mw[i], yy, t0);
            t1 = fmaf(mw[i] * yy, yy, t1);
          }
        }
        st0[cb + c] += t0; st1[cb + c] += t1;
      }
      __builtin_amdgcn_sched_barrier(0);                       // the next pair's MFMAs stay behind this pair's stores (register budget)
    }
#pragma unroll
    for (int j = 0; j < KL; ++j) a[j] = an[j];
  }
  if (stats) {
    const int slot = wave * 2 + h;
#pragma unroll
    for (int c = 0; c < CT; ++c) { red[0][slot][c * 32 + lrow] = (double)st0[c]; red[1][slot][c * 32 + lrow] = (double)st1[c]; }
    __syncthreads();
    for (int e = tid; e < 2 * CT * 32; e += GM_THREADS) {
      const int sidx = e / (CT * 32), col = e - sidx * (CT * 32);
      if (col < cols) {
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < GM_THREADS / 32; ++g) t += red[sidx][g][col];
        ep.partial[((long long)blockIdx.x * 2 + sidx) * cols + col] = t;
        // the finalize kernel sums `partial_blocks` rows: the ones no workgroup owns read as zero (no memset launch)
        for (int pb = blockIdx.x + gridDim.x; pb < ep.partial_blocks; pb += gridDim.x)
          ep.partial[((long long)pb * 2 + sidx) * cols + col] = 0.0;
      }
    }
  }
}

// ---- narrow weight gradient on the matrix pipe, no LDS in the loop (round 3): kcols <= 32 ------------------------------------
// dw[n][k] = sum_r P[r][n] Q[r][k] with the ROWS as the MFMA reduction index: lane (l = lane & 31, h = lane >> 5) holds
// P[r][n0 + l] as the A operand and Q[r][l] (l < kcols) as the B operand of v_mfma_f32_32x32x2_f32 for the rows r = r0 + 4 h + i
// of step i -- consecutive lanes read consecutive columns of one row (128-byte segments), a wave takes 8 rows per trip and keeps
// the (32 CT) x 32 block of dw in its accumulators over all its rows; the 4 waves of a workgroup meet in LDS once, in a fixed order,
// and the workgroup leaves one partial like the other weight-gradient kernels.  (The streaming form above spends ~110 VALU
// instructions per thread and row -- 16 predicated scalar loads + 64 FMAs whatever kcols is -- and measured 14-21 us where its
// 34 MB are 5 us of HBM time.)
template <int CT, int PM, int QM>
__global__ void __launch_bounds__(GM_THREADS, CT > 1 ? 3 : 4)      // two column tiles + the next trip's rows: 168 VGPRs
wgrad_narrow_kernel(long long rows_arg, const int *__restrict__ rows_dev, int ncols, int kcols, RowOperand P, RowOperand Q,
                    float *__restrict__ partial) {
  const long long rows = rows_dev ? min(rows_arg, (long long)*rows_dev) : rows_arg;
  __shared__ float red[GM_THREADS / 64][32][33];
  const int tid = threadIdx.x, lane = tid & 63, l = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // (wave-uniform: see gemm_narrow_kernel)
  const int n0 = blockIdx.y * (32 * CT);
  ColCoef<1> pc[CT], qc;
#pragma unroll
  for (int c = 0; c < CT; ++c) op_coef<1, PM>(P, n0 + c * 32 + l, n0 + c * 32 + l < ncols, pc[c]);
  op_coef<1, QM>(Q, l, l < kcols, qc);
  f32x16 acc[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  const long long nblk = (rows + 7) >> 3, bstep = (long long)gridDim.x * (GM_THREADS / 64);
  // The next trip's rows are requested before this trip's MFMAs, and every load is UNCONDITIONAL (row and column clamped to a
  // valid address, the value discarded by op_finish).  A predicated op_load is a branch around its loads: the compiler then
  // cannot count the loads in flight and waits for all of them (s_waitcnt vmcnt(0)) -- including the ones just requested for the
  // next trip -- and the zero-initialisation of a predicated destination waits for the previous load into that register: the
  // 12 loads of a trip went out as 4 dependent groups.  Invisible with the compacted classification rows (a wave makes ~4
  // trips: 11.7 us), 137 us per launch on the segmentation step's dense 524 288-row first layers (32 trips per wave).
  auto request = [&](long long blk_, RawVec<1> (&pr)[CT][4], RawVec<1> (&qr)[4]) {
    const long long r0_ = blk_ * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rl = 4 * h + i;
      const int rc = r0_ + rl < rows ? rl : 0;                      // (a row beyond the end: the block's first row, discarded)
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const int n = n0 + c * 32 + l;
        op_load<1, PM>(P, r0_, rc, n < ncols ? n : 0, true, pr[c][i]);
      }
      op_load<1, QM>(Q, r0_, rc, l < kcols ? l : 0, true, qr[i]);
    }
  };
  RawVec<1> praw[CT][4], qraw[4], pnext[CT][4], qnext[4];
  long long blk = (long long)blockIdx.x * (GM_THREADS / 64) + wave;
  if (blk < nblk) request(blk, praw, qraw);
  for (; blk < nblk; blk += bstep) {
    const long long r0 = blk * 8;
    const bool more = blk + bstep < nblk;
    if (more) request(blk + bstep, pnext, qnext);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rl = 4 * h + i;
      const bool rok = r0 + rl < rows;
      float qv[1];
      op_finish<1, QM>(Q, qc, qraw[i], r0 + rl, rok && l < kcols, qv);
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        float pv[1];
        op_finish<1, PM>(P, pc[c], praw[c][i], r0 + rl, rok && n0 + c * 32 + l < ncols, pv);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(pv[0], qv[0], acc[c], 0, 0, 0);
      }
    }
    if (more) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        qraw[i] = qnext[i];
#pragma unroll
        for (int c = 0; c < CT; ++c) praw[c][i] = pnext[c][i];
      }
    }
  }
  // D[i][j]: j = lane & 31 is the k index, row (reg & 3) + 8 (reg >> 2) + 4 h the column n of P
  float *dst = partial + (long long)blockIdx.x * ncols * kcols;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave][(i & 3) + 8 * (i >> 2) + 4 * h][l] = acc[c][i];
    __syncthreads();
    for (int e = tid; e < 32 * 32; e += GM_THREADS) {
      const int nl = e >> 5, k = e & 31, n = n0 + c * 32 + nl;
      if (n < ncols && k < kcols) {
        float t = red[0][nl][k];
#pragma unroll
        for (int w2 = 1; w2 < GM_THREADS / 64; ++w2) t += red[w2][nl][k];
        dst[(long long)n * kcols + k] = t;
      }
    }
  }
}

// (Round 4 built this product with 16-byte loads and a wave-private LDS transpose -- 8 lanes per 32-column row, ds_write_b128 by
//  (row, 4 columns), ds_read_b32 by (column, row), 10 load instructions per 32 rows instead of 48 -- for the segmentation step's dense
//  524 288-row first layers, as the round-3 review proposed.  tools/wgrad_narrow_bench.py, kernel + partial reduction, 167 MB: 44.8 us
//  warm / 70.6 us cold against 45.0 / 68.9 us for the kernel above: the load width is not what bounds it (the workgroup count is at
//  its optimum, 512: 256 -> 59 us, 1 024 -> 43 us, 2 048 -> 48 us).  Not kept.)

// ---- narrow weight gradient: kcols <= 16 (first-layer branches: 3 / 6 / 10 / 16 input channels) ---------------------
// dw[n][k] = sum_r P[r][n] * Q[r][k] is a pure streaming reduction here: 2 * 16 flop per byte of P.  No LDS, no
// barriers, no matrix pipe in the loop: thread (tn, tr) owns 4 columns of P and every k for the rows tr, tr + RG, ...
// of its workgroup's row window; its P values are used by nobody else (one float4 global load each), the <= 16 Q
// values of a row are the same address for the 32..8 lanes that share the row (one cache line, broadcast by L1).
// Two rows are in flight per thread.  The RG row groups are summed through LDS at the end (fixed order) and leave as
// one partial per workgroup, reduced over workgroups by reduce_partials_kernel like the MFMA variant.

template <int NB, int VP, int PM, int QM>       // NB = 32 | 64 | 128 columns of P per workgroup
__global__ void __launch_bounds__(GM_THREADS)
wgrad_small_kernel(long long rows_arg, const int *__restrict__ rows_dev, int ncols, int kcols, RowOperand P, RowOperand Q,
                   float *__restrict__ partial) {
  constexpr int TPN = NB / 4, RG = GM_THREADS / TPN;          // threads across columns, row groups
  const long long rows = rows_dev ? min(rows_arg, (long long)*rows_dev) : rows_arg;
  extern __shared__ __attribute__((aligned(16))) float smem[];   // RG x NB x 16 floats for the final reduction
  const int tid = threadIdx.x, tn = tid % TPN, tr = tid / TPN;
  const int n0 = blockIdx.y * NB + tn * 4;
  const bool n_ok = n0 < ncols;                                  // ncols % 4 == 0 is required by the launcher

  ColCoef<4> pcoef;
  op_coef<4, PM>(P, n0, n_ok, pcoef);
  float qs1[WS_KP], qt1[WS_KP], qs2[WS_KP], qt2[WS_KP];
#pragma unroll
  for (int k = 0; k < WS_KP; ++k) {
    ColCoef<1> c;
    op_coef<1, QM>(Q, k, k < kcols, c);
    qs1[k] = c.s1[0]; qt1[k] = c.t1[0]; qs2[k] = c.s2[0]; qt2[k] = c.t2[0];
  }
  float acc[4][WS_KP];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < WS_KP; ++k) acc[i][k] = 0.f;

  const long long step = (long long)gridDim.x * RG;
  auto load_row = [&](long long r, RawVec<4> &pr, RawVec<1> (&qr)[WS_KP]) {
    const bool ok = r < rows;
    op_load<4, PM>(P, r, 0, n0, ok && n_ok, pr);
#pragma unroll
    for (int k = 0; k < WS_KP; ++k) op_load<1, QM>(Q, r, 0, k, ok && k < kcols, qr[k]);
  };
  auto fma_row = [&](long long r, const RawVec<4> &pr, const RawVec<1> (&qr)[WS_KP]) {
    const bool ok = r < rows;
    float pv[4];
    op_finish<4, PM>(P, pcoef, pr, r, ok && n_ok, pv);
#pragma unroll
    for (int k = 0; k < WS_KP; ++k) {
      ColCoef<1> c;
      c.s1[0] = qs1[k]; c.t1[0] = qt1[k]; c.s2[0] = qs2[k]; c.t2[0] = qt2[k];
      float qv[1];
      op_finish<1, QM>(Q, c, qr[k], r, ok && k < kcols, qv);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][k] = fmaf(pv[i], qv[0], acc[i][k]);
    }
  };
  RawVec<4> pa, pb;
  RawVec<1> qa[WS_KP], qb[WS_KP];
  long long r = (long long)blockIdx.x * RG + tr;
  load_row(r, pa, qa);
  for (; r < rows; r += 2 * step) {
    load_row(r + step, pb, qb);
    fma_row(r, pa, qa);
    load_row(r + 2 * step, pa, qa);
    fma_row(r + step, pb, qb);
  }

  // sum the RG row groups in a fixed order, one partial per workgroup
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < WS_KP; ++k) smem[(tr * NB + tn * 4 + i) * WS_KP + k] = acc[i][k];
  __syncthreads();
  float *dst = partial + (long long)blockIdx.x * ncols * kcols;
  for (int e = tid; e < NB * WS_KP; e += GM_THREADS) {
    const int nl = e / WS_KP, k = e - nl * WS_KP;
    float t = 0.f;
    for (int g = 0; g < RG; ++g) t += smem[(g * NB + nl) * WS_KP + k];
    const int n = blockIdx.y * NB + nl;
    if (n < ncols && k < kcols) dst[(long long)n * kcols + k] = t;
  }
}

// out[e] = sum_c partial[c][e]   (deterministic order).  32 outputs x 8 chunk slices per workgroup:
// consecutive lanes read consecutive outputs (coalesced), each thread sums every 8th chunk with
// independent loads in flight, slices are combined in a fixed order through LDS.
// (bid, nb): the block's index and the number of blocks of the reduction -- the kernel's own grid, or a sub-range of
// bn_bwd_finalize_reduce_kernel's
__device__ __forceinline__ void reduce_partials_body(int bid, int nb, int chunks, long long n, const float *__restrict__ partial,
                                                     float *__restrict__ out) {
  __shared__ float red[8][32];
  const int ex = threadIdx.x & 31, sl = threadIdx.x >> 5;
  for (long long e0 = (long long)bid * 32; e0 < n; e0 += (long long)nb * 32) {
    const long long e = e0 + ex;
    float s = 0.f;
    if (e < n) {
      int c = sl;
      for (; c + 120 < chunks; c += 128) {        // 16 loads in flight: a 512-chunk reduction is 4 latencies deep, not 16
        float a[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) a[u] = partial[(long long)(c + 8 * u) * n + e];
        s += (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) +
             (((a[8] + a[9]) + (a[10] + a[11])) + ((a[12] + a[13]) + (a[14] + a[15])));
      }
      for (; c + 24 < chunks; c += 32) {
        const float a0 = partial[(long long)c * n + e], a1 = partial[(long long)(c + 8) * n + e];
        const float a2 = partial[(long long)(c + 16) * n + e], a3 = partial[(long long)(c + 24) * n + e];
        s += (a0 + a1) + (a2 + a3);
      }
      for (; c < chunks; c += 8) s += partial[(long long)c * n + e];
    }
    red[sl][ex] = s;
    __syncthreads();
    if (sl == 0 && e < n) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += red[k][ex];
      out[e] = t;
    }
    __syncthreads();
  }
}

// The same sum (same order: slice sl takes chunks sl, sl + 8, ..., the slices meet in ascending order) with 16-byte loads: 32 lanes
// x float4 = 128 consecutive outputs per slice and trip.  The 4-byte form moved the 32 MB of partials of a 1 024 x 512 weight
// gradient (16 row slabs) at 2 TB/s -- 16.5 us per launch, twice per classification step, 27 launches per segmentation step.
// n % 4 == 0 (every weight matrix the step reduces; the caller checks and keeps the scalar form otherwise).
__device__ __forceinline__ void reduce_partials_body4(int bid, int nb, int chunks, long long n, const float *__restrict__ partial,
                                                      float *__restrict__ out) {
  __shared__ float4 red4[8][32];
  const int ex = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const long long n4 = n >> 2;
  const float4 *p4 = reinterpret_cast<const float4 *>(partial);
  for (long long e0 = (long long)bid * 32; e0 < n4; e0 += (long long)nb * 32) {
    const long long e = e0 + ex;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < n4) {
      int c = sl;
      for (; c + 56 < chunks; c += 64) {          // 8 loads of 16 bytes in flight
        float4 a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = p4[(long long)(c + 8 * u) * n4 + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) { s.x += a[u].x; s.y += a[u].y; s.z += a[u].z; s.w += a[u].w; }
      }
      for (; c < chunks; c += 8) { const float4 a = p4[(long long)c * n4 + e]; s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w; }
    }
    red4[sl][ex] = s;
    __syncthreads();
    if (sl == 0 && e < n4) {
      float4 t = red4[0][ex];
#pragma unroll
      for (int k = 1; k < 8; ++k) { const float4 a = red4[k][ex]; t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w; }
      reinterpret_cast<float4 *>(out)[e] = t;
    }
    __syncthreads();
  }
}
__device__ __forceinline__ void reduce_partials_any(int bid, int nb, int chunks, long long n, const float *__restrict__ partial,
                                                    float *__restrict__ out) {
  // the 16-byte form for the WIDE gradients (few row slabs, many outputs); a narrow first-layer gradient is the opposite -- 512 slabs
  // of 512 outputs -- and wants its outputs spread over as many workgroups as possible: the 4-byte form gives it 4 x the workgroups
  // and 128 slabs per trip (with the 16-byte form such a reduction ran on 4 workgroups for 138 us)
  if (chunks <= 64 && n >= 4096 && (n & 3) == 0 && (((uintptr_t)partial | (uintptr_t)out) & 15) == 0)
    reduce_partials_body4(bid, nb, chunks, n, partial, out);
  else reduce_partials_body(bid, nb, chunks, n, partial, out);
}

__global__ void __launch_bounds__(GM_THREADS)
reduce_partials_kernel(int chunks, long long n, const float *__restrict__ partial, float *__restrict__ out) {
  reduce_partials_any(blockIdx.x, gridDim.x, chunks, n, partial, out);
}

// ---- BatchNorm statistics -> affine.  Workgroup = 8 channels x 32 slices of the partial rows: consecutive
// lanes read consecutive channels (64-byte rows of doubles), slices are combined through LDS in a
// fixed order (deterministic).
template <int NS>   // number of statistics reduced together
__device__ __forceinline__ void reduce_stat_rows(int cb, int c, int nblk, int nstat, const int (&which)[NS],
                                                 const double *__restrict__ partial, double (&out)[NS], bool &owner) {
  __shared__ double red[32][8][NS];
  const int ex = threadIdx.x & 7, sl = threadIdx.x >> 3;      // 8 channels (64 contiguous bytes) x 32 slices
  const int ch = cb * 8 + ex;                                  // cb: this workgroup's block of 8 channels
  double acc[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) acc[s] = 0.0;
  if (ch < c)
    for (int b = sl; b < nblk; b += 512) {     // 16 partial rows per trip, all loads in flight together: the kernel is
      double v[16][NS];                         // one memory latency deep at the usual 512 rows (it sits between every
#pragma unroll                                  // two GEMMs of a stack, so its latency is on the step's critical path)
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int s = 0; s < NS; ++s)
          v[u][s] = (b + 32 * u < nblk) ? partial[((long long)(b + 32 * u) * nstat + which[s]) * c + ch] : 0.0;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        double t[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = (v[4 * q][s] + v[4 * q + 1][s]) + (v[4 * q + 2][s] + v[4 * q + 3][s]);
        acc[s] += (t[0] + t[1]) + (t[2] + t[3]);
      }
    }
#pragma unroll
  for (int s = 0; s < NS; ++s) red[sl][ex][s] = acc[s];
  __syncthreads();
  owner = (sl == 0) && (ch < c);
  if (owner)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 32; ++k) t += red[k][ex][s];
      out[s] = t;
    }
}

__global__ void __launch_bounds__(256)
bn_finalize_kernel(int c, long long rows, int nblk, const double *__restrict__ partial,
                   const float *__restrict__ gamma, const float *__restrict__ beta, float eps, float momentum,
                   float *__restrict__ scale, float *__restrict__ shift, float *__restrict__ mean_out,
                   float *__restrict__ invstd_out, float *__restrict__ running_mean,
                   float *__restrict__ running_var) {
  const int which[2] = {0, 1};
  double sq[2];
  bool owner;
  // the owner's channel vectors are requested BEFORE the reduction (one memory round trip less behind it: these launches are a
  // chain of dependent latencies, ~5 us each whatever their size)
  // (unconditional loads at a clamped channel, absent vectors read `scale` instead: a load under a branch is waited for at the end of
  // its block, in front of the reduction's own loads)
  const int ch = blockIdx.x * 8 + (threadIdx.x & 7), chc = ch < c ? ch : c - 1;
  const float g_ = (gamma ? gamma : scale)[chc], b_ = (beta ? beta : scale)[chc];
  const float rm_ = (running_mean ? running_mean : scale)[chc], rv_ = (running_var ? running_var : scale)[chc];
  reduce_stat_rows<2>(blockIdx.x, c, nblk, 2, which, partial, sq, owner);
  if (!owner) return;
  const double mean = sq[0] / (double)rows;
  double var = sq[1] / (double)rows - mean * mean;
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  const double g = gamma ? (double)g_ : 1.0, bt = beta ? (double)b_ : 0.0;
  scale[ch] = (float)(g * invstd);
  shift[ch] = (float)(bt - mean * g * invstd);
  mean_out[ch] = (float)mean;
  invstd_out[ch] = (float)invstd;
  if (running_mean) {
    const double unbiased = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
    running_mean[ch] = (float)((1.0 - momentum) * (double)rm_ + momentum * mean);
    running_var[ch] = (float)((1.0 - momentum) * (double)rv_ + momentum * unbiased);
  }
}

// ---- BatchNorm backward sums -> coefficients of dy = p*dz + q*y + r ------------------------------
// which: 1 -> dgamma from stat row 1, 2 -> from stat row 2 (second branch of the two-branch first layer)
__global__ void __launch_bounds__(256)
bn_bwd_finalize_kernel(int c, long long rows, int nblk, int nstat, int which, const double *__restrict__ partial,
                       const float *__restrict__ scale, const float *__restrict__ mean,
                       const float *__restrict__ invstd, float *__restrict__ p, float *__restrict__ q,
                       float *__restrict__ r, float *__restrict__ dgamma, float *__restrict__ dbeta) {
  const int sel[2] = {0, which};
  double v[2];
  bool owner;
  reduce_stat_rows<2>(blockIdx.x, c, nblk, nstat, sel, partial, v, owner);
  if (!owner) return;
  const int ch = blockIdx.x * 8 + (threadIdx.x & 7);
  const double db = v[0], dg = v[1];
  const double s = scale[ch], is = invstd[ch], mu = mean[ch], m = (double)rows;
  // dy = s * (dz - db/m - yhat * dg/m),  yhat = (y - mu) * is
  const double qq = -s * is * dg / m;
  p[ch] = (float)s;
  q[ch] = (float)qq;
  r[ch] = (float)(-s * db / m - qq * mu);
  if (dgamma) dgamma[ch] = (float)dg;
  if (dbeta) dbeta[ch] = (float)db;
}

// The weight-gradient reduction of layer l and the BatchNorm-backward finalize of layer l-1 sit next to each other in
// the backward chain and are both a few microseconds of work behind ~5 us of graph-node latency: one launch, the first
// `nfin` workgroups finalize, the others reduce (no barrier spans both roles).
__global__ void __launch_bounds__(256)
bn_bwd_finalize_reduce_kernel(int c, long long rows, int nblk, int nstat, int which, const double *__restrict__ partial,
                              const float *__restrict__ scale, const float *__restrict__ mean,
                              const float *__restrict__ invstd, float *__restrict__ p, float *__restrict__ q,
                              float *__restrict__ r, float *__restrict__ dgamma, float *__restrict__ dbeta,
                              int nfin, int red_chunks, long long red_n, const float *__restrict__ red_partial,
                              float *__restrict__ red_out) {
  if ((int)blockIdx.x >= nfin) {
    reduce_partials_any(blockIdx.x - nfin, gridDim.x - nfin, red_chunks, red_n, red_partial, red_out);
    return;
  }
  const int sel[2] = {0, which};
  double v[2];
  bool owner;
  reduce_stat_rows<2>(blockIdx.x, c, nblk, nstat, sel, partial, v, owner);
  if (!owner) return;
  const int ch = blockIdx.x * 8 + (threadIdx.x & 7);
  const double db = v[0], dg = v[1];
  const double s = scale[ch], is = invstd[ch], mu = mean[ch], m = (double)rows;
  const double qq = -s * is * dg / m;
  p[ch] = (float)s;
  q[ch] = (float)qq;
  r[ch] = (float)(-s * db / m - qq * mu);
  if (dgamma) dgamma[ch] = (float)dg;
  if (dbeta) dbeta[ch] = (float)db;
}

// ---- several of those small jobs in ONE launch ---------------------------------------------------------------------
// Every one of them is a few microseconds of work behind ~5 us of graph-node latency, and a stack's backward issues them
// in pairs (the two BatchNorms of a two-branch first layer, the two first-layer weight gradients): rs_backward_tail takes
// up to RS_TAIL_FIN_MAX BatchNorm-backward finalizes and up to RS_TAIL_RED_MAX weight-gradient reductions, workgroup
// ranges [start[i], start[i + 1]) per job; rs_bn_finalize_batch the forward statistics of up to RS_BN_BATCH_MAX layers.
struct TailStarts { int fin[RS_TAIL_FIN_MAX + 1]; int red[RS_TAIL_RED_MAX + 1]; };

__global__ void __launch_bounds__(256)
backward_tail_kernel(rs_backward_tail_work w, TailStarts st) {
  const int b = blockIdx.x;
  if (b < st.fin[w.nfin]) {
    int i = 0;
    for (int k = 1; k < w.nfin; ++k) if (b >= st.fin[k]) i = k;
    const rs_bn_bwd_item &it = w.fin[i];
    const int cb = b - st.fin[i];
    const int sel[2] = {0, it.which};
    double v[2];
    bool owner;
    const int ch = cb * 8 + (threadIdx.x & 7), chc = ch < it.c ? ch : it.c - 1;
    const float s_ = it.scale[chc], is_ = it.invstd[chc], mu_ = it.mean[chc];       // requested before the reduction (see bn_finalize_kernel)
    reduce_stat_rows<2>(cb, it.c, it.nblk, it.nstat, sel, it.partial, v, owner);
    if (!owner) return;
    const double db = v[0], dg = v[1];
    const double s = s_, is = is_, mu = mu_, m = it.rows_dev ? (double)min(it.rows, (long long)*it.rows_dev) : (double)it.rows;
    const double qq = -s * is * dg / m;                 // dy = s * (dz - db/m - yhat * dg/m),  yhat = (y - mu) * is
    it.p[ch] = (float)s;
    it.q[ch] = (float)qq;
    it.r[ch] = (float)(-s * db / m - qq * mu);
    if (it.dgamma) it.dgamma[ch] = (float)dg;
    if (it.dbeta) it.dbeta[ch] = (float)db;
    return;
  }
  int j = 0;
  for (int k = 1; k < w.nred; ++k) if (b >= st.red[k]) j = k;
  reduce_partials_any(b - st.red[j], st.red[j + 1] - st.red[j], w.red[j].chunks, w.red[j].n, w.red[j].partial, w.red[j].out);
}

struct BnBatch { rs_bn_item it[RS_BN_BATCH_MAX]; int start[RS_BN_BATCH_MAX + 1]; int n; };

__global__ void __launch_bounds__(256)
bn_finalize_batch_kernel(BnBatch w) {
  const int b = blockIdx.x;
  int i = 0;
  for (int k = 1; k < w.n; ++k) if (b >= w.start[k]) i = k;
  const rs_bn_item &it = w.it[i];
  const int cb = b - w.start[i];
  const int which[2] = {0, 1};
  double sq[2];
  bool owner;
  const int ch = cb * 8 + (threadIdx.x & 7), chc = ch < it.c ? ch : it.c - 1;      // requested before the reduction (see bn_finalize_kernel)
  const float g_ = (it.gamma ? it.gamma : it.scale)[chc], b_ = (it.beta ? it.beta : it.scale)[chc];
  const float rm_ = (it.running_mean ? it.running_mean : it.scale)[chc], rv_ = (it.running_var ? it.running_var : it.scale)[chc];
  reduce_stat_rows<2>(cb, it.c, it.nblk, 2, which, it.partial, sq, owner);
  if (!owner) return;
  const long long nrows = it.rows_dev ? min(it.rows, (long long)*it.rows_dev) : it.rows;      // (a packed batch under a captured capacity: the count is device data)
  const double rows = (double)nrows;
  const double mean = sq[0] / rows;
  double var = sq[1] / rows - mean * mean;
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)it.eps);
  const double g = it.gamma ? (double)g_ : 1.0, bt = it.beta ? (double)b_ : 0.0;
  it.scale[ch] = (float)(g * invstd);
  it.shift[ch] = (float)(bt - mean * g * invstd);
  it.save_mean[ch] = (float)mean;
  it.save_invstd[ch] = (float)invstd;
  if (it.running_mean) {
    const double unbiased = nrows > 1 ? var * rows / (rows - 1.0) : var;
    it.running_mean[ch] = (float)((1.0 - it.momentum) * (double)rm_ + it.momentum * mean);
    it.running_var[ch] = (float)((1.0 - it.momentum) * (double)rv_ + it.momentum * unbiased);
  }
}

// element e of the pooled layer's pre-activation y: fp32, or bf16 (bf16 activation storage) behind the same pointer type
template <bool BF> __device__ __forceinline__ float ldy(const float *y, long long e) {
  if constexpr (BF) return __uint_as_float((unsigned)reinterpret_cast<const unsigned short *>(y)[e] << 16);
  else return y[e];
}

// ---- pooling over nsample, fused with the last BatchNorm + ReLU -----------------------------------
// out[g][c] = max_k relu(scale*y[g*ns+k][c] + shift), arg = first k attaining it
template <bool BF>
__global__ void __launch_bounds__(GM_THREADS)
pool_max_kernel(long long groups, int ns, int c, int relu, const int *__restrict__ offsets, const float *__restrict__ y,
                const float *__restrict__ scale, const float *__restrict__ shift, float *__restrict__ out,
                int *__restrict__ arg) {
  const long long total = groups * c;
  for (long long e = (long long)blockIdx.x * GM_THREADS + threadIdx.x; e < total; e += (long long)gridDim.x * GM_THREADS) {
    const long long g = e / c;
    const int ch = (int)(e - g * c);
    const float s = scale ? scale[ch] : 1.f, t = shift ? shift[ch] : 0.f;
    float best = -INFINITY; int bi = 0;
    const long long base = offsets ? offsets[g] : g * ns;             // ragged (compacted) or dense groups
    const int len = offsets ? offsets[g + 1] - offsets[g] : ns;
    if (!offsets) {                                                  // dense groups: a plain loop the compiler pipelines itself
      for (int k = 0; k < len; ++k) {                                // (measured on the segmentation step: 4-row batches 65 us, this 51 us)
        float z = fmaf(s, ldy<BF>(y, (base + k) * c + ch), t);
        if (relu) z = fmaxf(z, 0.f);
        if (z > best) { best = z; bi = k; }
      }
    } else
    for (int k0 = 0; k0 < len; k0 += 4) {                            // ragged groups, 4 rows in flight (one per trip was a chain of `len` round trips)
      float z[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) z[u] = ldy<BF>(y, (base + min(k0 + u, len - 1)) * c + ch);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float zz = fmaf(s, z[u], t);
        if (relu) zz = fmaxf(zz, 0.f);
        if (k0 + u < len && zz > best) { best = zz; bi = k0 + u; }
      }
    }
    out[e] = best;
    if (arg) arg[e] = bi;
  }
}

// Few, long, dense groups (the group_all stage: 32 groups x 128 rows x 1024 channels): one thread per (group, channel)
// is 32 K threads walking 128 rows each -- half the chip, a 128-deep load chain (39 us for 16 MB).  Here a workgroup
// takes 64 channels of one group, its 4 waves take every 4th row with 4 loads in flight, and the slices meet in LDS;
// the first row attaining the maximum wins, as in the sequential scan.
template <bool BF>
__global__ void __launch_bounds__(GM_THREADS)
pool_max_long_kernel(int ns, int c, int relu, const float *__restrict__ y, const float *__restrict__ scale,
                     const float *__restrict__ shift, float *__restrict__ out, int *__restrict__ arg) {
  __shared__ float bz[4][64];
  __shared__ int bk[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int ch = blockIdx.y * 64 + tx;
  const long long g = blockIdx.x;
  float best = -INFINITY; int bi = 0x7fffffff;
  if (ch < c) {
    const float s = scale ? scale[ch] : 1.f, t = shift ? shift[ch] : 0.f;
    const long long col = g * ns * c + ch;
    for (int k0 = ty; k0 < ns; k0 += 16) {
      float z[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) z[u] = (k0 + 4 * u < ns) ? ldy<BF>(y, col + (long long)(k0 + 4 * u) * c) : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (k0 + 4 * u >= ns) break;
        float v = fmaf(s, z[u], t);
        if (relu) v = fmaxf(v, 0.f);
        if (v > best) { best = v; bi = k0 + 4 * u; }
      }
    }
  }
  bz[ty][tx] = best; bk[ty][tx] = bi;
  __syncthreads();
  if (ty == 0 && ch < c) {
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float v = bz[w][tx]; const int k = bk[w][tx];
      if (v > best || (v == best && k < bi)) { best = v; bi = k; }
    }
    out[g * c + ch] = best; arg[g * c + ch] = bi == 0x7fffffff ? 0 : bi;
  }
}

// out = relu(scale * (scale >= 0 ? ymax : ymin) + shift), arg = matching index (resolves the fused pooling)
__global__ void __launch_bounds__(GM_THREADS)
pool_select_kernel(long long groups, int c, const float *__restrict__ ymax, const float *__restrict__ ymin,
                   const int *__restrict__ amax, const int *__restrict__ amin, const float *__restrict__ scale,
                   const float *__restrict__ shift, float *__restrict__ out, int *__restrict__ arg) {
  const long long total = groups * c;
  for (long long e = (long long)blockIdx.x * GM_THREADS + threadIdx.x; e < total; e += (long long)gridDim.x * GM_THREADS) {
    const int ch = (int)(e % c);
    const float s = scale[ch];
    const bool up = s >= 0.f;
    out[e] = fmaxf(fmaf(s, up ? ymax[e] : ymin[e], shift[ch]), 0.f);
    arg[e] = up ? amax[e] : amin[e];
  }
}

// v[g][c] = dout * (out > 0)  (out == NULL: v = dout); partial sums {sum v, sum v * yhat[arg row]} per column (BatchNorm backward
// of the pooled layer computed from G x C data only)
template <bool BF>
__global__ void __launch_bounds__(GM_THREADS)
pool_max_bwd_kernel(long long groups_arg, const int *__restrict__ groups_dev, int ns, int c, const int *__restrict__ offsets, const float *__restrict__ dout, long long ldd,
                    const float *__restrict__ out, const int *__restrict__ arg, const float *__restrict__ y,
                    const float *__restrict__ mean, const float *__restrict__ invstd, float *__restrict__ v,
                    double *__restrict__ partial, int partial_blocks) {
  const long long groups = groups_dev ? min(groups_arg, (long long)*groups_dev) : groups_arg;      // (groups beyond the device count: not read, not written, not summed)
  // workgroup = 64 columns x 4 group lanes; column sums stay in registers over the group loop and are
  // combined across the 4 lanes through LDS (fixed order)
  __shared__ double red[4][64][2];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int ch = blockIdx.y * 64 + tx;
  double s0 = 0.0, s1 = 0.0;
  if (ch < c) {
    const float mu = mean[ch], is = invstd[ch];
    // two groups per trip: a group is a chain of two dependent round trips (out / arg / offsets -> the y row they select), and a
    // thread sees only 2-4 groups -- one after the other that was 4-8 exposed round trips per launch
    const long long gstep = (long long)gridDim.x * 4;
    for (long long g = (long long)blockIdx.x * 4 + ty; g < groups; g += 2 * gstep) {
      const long long g1 = g + gstep;
      const bool ok1 = g1 < groups;
      const long long e0 = g * c + ch, e1 = (ok1 ? g1 : g) * c + ch;
      const float o0 = out ? out[e0] : 1.f, o1 = out ? out[e1] : 1.f;
      const float d0 = dout[g * ldd + ch], d1 = dout[(ok1 ? g1 : g) * ldd + ch];
      const int a0 = arg ? arg[e0] : 0, a1 = arg ? arg[e1] : 0;
      const long long b0 = offsets ? (long long)offsets[g] : g * ns, b1 = offsets ? (long long)offsets[ok1 ? g1 : g] : (ok1 ? g1 : g) * ns;
      const float y0 = ldy<BF>(y, (b0 + a0) * c + ch), y1 = ldy<BF>(y, (b1 + a1) * c + ch);
      const float v0 = o0 > 0.f ? d0 : 0.f;                        // out == NULL: the pooled layer ended without a ReLU
      if (v) v[e0] = v0;                                           // (v == NULL: only the sums -- v would be dout itself)
      s0 += (double)v0;
      s1 += (double)(v0 * ((y0 - mu) * is));
      if (ok1) {                                                  // (same order of the sums as one group per trip)
        const float v1 = o1 > 0.f ? d1 : 0.f;
        if (v) v[e1] = v1;
        s0 += (double)v1;
        s1 += (double)(v1 * ((y1 - mu) * is));
      }
    }
  }
  red[ty][tx][0] = s0; red[ty][tx][1] = s1;
  __syncthreads();
  if (ty == 0 && ch < c) {
    partial[((long long)blockIdx.x * 2 + 0) * c + ch] = (red[0][tx][0] + red[1][tx][0]) + (red[2][tx][0] + red[3][tx][0]);
    partial[((long long)blockIdx.x * 2 + 1) * c + ch] = (red[0][tx][1] + red[1][tx][1]) + (red[2][tx][1] + red[3][tx][1]);
    // the finalize kernel sums `partial_blocks` rows: the ones no workgroup owns read as zero (no memset launch)
    for (int pb = blockIdx.x + gridDim.x; pb < partial_blocks; pb += gridDim.x) {
      partial[((long long)pb * 2 + 0) * c + ch] = 0.0;
      partial[((long long)pb * 2 + 1) * c + ch] = 0.0;
    }
  }
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

int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v ? atoi(v) : dflt;
}

int persistent_blocks(long long tiles, int tiles_n, int bm, bool fwd) {
  static const int slots128 = env_int("RS_GEMM_SLOTS", 512), slots64 = env_int("RS_GEMM_SLOTS64", 512);
  static const int slots64f = env_int("RS_GEMM_SLOTS64_FWD", 0);   // forward instances (<= 168 VGPRs) can run 3 per CU
  const int slots = bm == 64 ? ((fwd && slots64f > 0) ? slots64f : slots64) : slots128;
  long long want = slots / (tiles_n > 0 ? tiles_n : 1);   // ~2 (tall tiles) / 3-4 (short tiles) workgroups per CU over the whole grid
  if (want < 64) want = 64;
  return (int)(tiles < want ? tiles : want);
}

bool uses_b(int mode) { return mode == OPM_RELU2 || mode == OPM_AFF2 || mode == OPM_POOLED; }
bool aligned_to(const void *p, int bytes) { return p == nullptr || ((uintptr_t)p % (uintptr_t)bytes) == 0; }

// widest vector (4, 2, 1 floats) every access of this operand over `cols` columns is aligned for
int pick_vec(const RowOperand &o, int cols) {
  for (int v = 4; v >= 2; v >>= 1) {
    if (cols % v) continue;
    bool ok = aligned_to(o.a, 4 * v) && (o.lda % v == 0);
    if (uses_b(o.mode)) ok = ok && aligned_to(o.b, 4 * v) && (o.ldb % v == 0);
    if (o.mode != OPM_ID && o.mode != OPM_BCAST)
      ok = ok && aligned_to(o.s1, 4 * v) && aligned_to(o.t1, 4 * v) && aligned_to(o.s2, 4 * v) && aligned_to(o.t2, 4 * v);
    if (o.mode == OPM_POOLED) ok = ok && aligned_to(o.arg, 4 * v);
    if (ok) return v;
  }
  return 1;
}

int check_operand(const char *who, const RowOperand *o, long long rows) {
  RS_REQUIRE(o, "%s: operand descriptor is NULL", who);
  RS_REQUIRE(o->mode >= OPM_ID && o->mode <= OPM_BCAST, "%s: unknown operand mode %d", who, o->mode);
  RS_REQUIRE(o->a, "%s: operand tensor is NULL", who);
  if (o->mode == OPM_RELU1) RS_REQUIRE(o->s1 && o->t1, "%s: RELU1 operand needs scale/shift", who);
  if (o->mode == OPM_RELU2) RS_REQUIRE(o->b && o->s1 && o->t1 && o->s2 && o->t2, "%s: RELU2 operand needs two tensors and two scale/shift pairs", who);
  if (o->mode == OPM_AFF2) RS_REQUIRE(o->b && o->s1 && o->t1 && o->s2, "%s: AFF2 operand needs dz, y and p/q/r", who);
  if (o->mode == OPM_POOLED) RS_REQUIRE(o->b && o->s1 && o->t1 && o->s2 && o->arg && (o->ns > 0 || (o->grp && o->slot)), "%s: POOLED operand needs v, arg, y, p/q/r and nsample or (grp, slot)", who);
  if (o->mode == OPM_BCAST) RS_REQUIRE(o->ns > 0, "%s: BCAST operand needs nsample", who);
  if (o->mode == OPM_POOLED || o->mode == OPM_BCAST) RS_REQUIRE(rows < 2147483647LL, "%s: pooled operands need rows < 2^31", who);
  return RS_OK;
}

thread_local bool g_gemm_refused = false;      // set by launch_gemm_m when a tile's LDS does not fit the device

template <int BM, int BN, int V, bool BF>
void launch_gemm_m(dim3 grid, hipStream_t st, long long rows, const int *rows_dev, int kdim, int cols, const RowOperand &E,
                   const float *w, int ldw, const Epilogue &ep) {
  const size_t lds = (BF && RS_SPLIT) ? sizeof(float) * 2 * 12 * ((BM * 4 + 16) + (BN * 4 + 16))      // unit 4: three parts of four planes per operand and stage
                                      : sizeof(float) * (2 * AStage<BM>::SIZE + 2 * WStage<BN>::SIZE);
  if (lds > (size_t)rs_lds_limit()) {      // (fail with a message, not at dispatch; gemm_rows_impl returns the error)
    rs_set_error("rs_mlp_gemm_rows: the %d x %d tile needs %zu bytes of LDS per workgroup, the device offers %d", BM, BN, lds, rs_lds_limit());
    g_gemm_refused = true;
    return;
  }
  // wave-specialised instances (8 waves: 4 MFMA + 4 loader, see the kernel): fp32, 64-row tiles, vector operands
  static const int ws_on = env_int("RS_GEMM_WS", 0);       // measured slower on the fused backward instances and equal on the forward ones: DESIGN.md 5
  if constexpr (BM == 64 && !BF && V >= 2) {
    if (ws_on) {
#define RS_GW(M_) hipLaunchKernelGGL((gemm_rows_kernel<BM, BN, V, M_, BF, true>), grid, dim3(2 * GM_THREADS), lds, st, rows, rows_dev, kdim, cols, E, w, ldw, ep)
      switch (E.mode) {
        case OPM_ID: RS_GW(OPM_ID); break;
        case OPM_RELU1: RS_GW(OPM_RELU1); break;
        case OPM_RELU2: RS_GW(OPM_RELU2); break;
        case OPM_AFF2: RS_GW(OPM_AFF2); break;
        case OPM_POOLED: if (E.grp) RS_GW(OPM_POOLED_RAGGED); else RS_GW(OPM_POOLED_DENSE); break;
        default: RS_GW(OPM_BCAST); break;
      }
#undef RS_GW
      return;
    }
  }
#define RS_G(M_) hipLaunchKernelGGL((gemm_rows_kernel<BM, BN, V, M_, BF>), grid, dim3(GM_THREADS), lds, st, rows, rows_dev, kdim, cols, E, w, ldw, ep)
  if (V == 1) { RS_G(-1); return; }                 // odd sizes: one generic (runtime-mode) kernel
  switch (E.mode) {
    case OPM_ID: RS_G(OPM_ID); break;
    case OPM_RELU1: RS_G(OPM_RELU1); break;
    case OPM_RELU2: RS_G(OPM_RELU2); break;
    case OPM_AFF2: RS_G(OPM_AFF2); break;
    case OPM_POOLED: if (E.grp) RS_G(OPM_POOLED_RAGGED); else RS_G(OPM_POOLED_DENSE); break;
    default: RS_G(OPM_BCAST); break;
  }
#undef RS_G
}
template <int BM, int BN>
void launch_gemm(bool bf, int v, dim3 grid, hipStream_t st, long long rows, const int *rows_dev, int kdim, int cols, const RowOperand &E,
                 const float *w, int ldw, const Epilogue &ep) {
#if RS_TU_HAS_BF16
  if (bf && v == 4) { launch_gemm_m<BM, BN, 4, true>(grid, st, rows, rows_dev, kdim, cols, E, w, ldw, ep); return; }
  if (bf && v == 2) { launch_gemm_m<BM, BN, 2, true>(grid, st, rows, rows_dev, kdim, cols, E, w, ldw, ep); return; }
#endif
#if !RS_TU_BF16_ONLY      // (the bf16-only unit calls with bf = true: vector operands never get here)
  if (v == 4) { launch_gemm_m<BM, BN, 4, false>(grid, st, rows, rows_dev, kdim, cols, E, w, ldw, ep); return; }
  if (v == 2) { launch_gemm_m<BM, BN, 2, false>(grid, st, rows, rows_dev, kdim, cols, E, w, ldw, ep); return; }
#endif
  (void)bf;
  if constexpr (BM == 128) launch_gemm_m<BM, BN, 1, false>(grid, st, rows, rows_dev, kdim, cols, E, w, ldw, ep);   // scalar operands: tall tile only, fp32 (bf16 staging packs pairs)
}

template <int WN, int WK, int TN, int TK, int VP, int VQ, bool BF>
void launch_wgrad_m(dim3 grid, hipStream_t st, long long rows, const int *rows_dev, int ncols, int kcols, const RowOperand &P,
                    const RowOperand &Q, float *partial) {
  const size_t lds = sizeof(float) * ((BF && RS_SPLIT && (WN * TN + WK * TK) * 32 > 192) ? 1 : 2) * ((BF && RS_SPLIT) ? 3 * (WG_BR / 2) : WG_BR) *
                     (WN * TN * 32 + WK * TK * 32);
#define RS_WG(PM_, QM_) hipLaunchKernelGGL((wgrad_kernel<WN, WK, TN, TK, VP, VQ, PM_, QM_, BF>), grid, dim3(GM_THREADS), lds, st, rows, rows_dev, ncols, kcols, P, Q, partial)
#define RS_WGQ(PM_) do { if (Q.mode == OPM_ID) RS_WG(PM_, OPM_ID); else if (Q.mode == OPM_RELU1) RS_WG(PM_, OPM_RELU1); else if (Q.mode == OPM_RELU2) RS_WG(PM_, OPM_RELU2); else RS_WG(-1, -1); } while (0)
  if (VP == 1 || VQ == 1) { RS_WG(-1, -1); return; }
  if (P.mode == OPM_AFF2) RS_WGQ(OPM_AFF2);
  else if (P.mode == OPM_POOLED) { if (P.grp) RS_WGQ(OPM_POOLED_RAGGED); else RS_WGQ(OPM_POOLED_DENSE); }
  else if (P.mode == OPM_BCAST) RS_WGQ(OPM_BCAST);
  else RS_WG(-1, -1);
#undef RS_WGQ
#undef RS_WG
}
// bf16 staging pairs the rows a thread owns: needs vector operands and an even number of vectors per thread and stage
template <int WN, int WK, int TN, int TK, int VP, int VQ>
constexpr bool wgrad_bf16_ok() {
  return VP >= 2 && VQ >= 2 && (WG_BR * WN * TN * 32 / VP / GM_THREADS) % 2 == 0 &&
         ((WG_BR * WK * TK * 32 / VQ + GM_THREADS - 1) / GM_THREADS) % 2 == 0;
}
template <int WN, int WK, int TN, int TK, int VP, int VQ>
void launch_wgrad_p(bool bf, dim3 grid, hipStream_t st, long long rows, const int *rows_dev, int ncols, int kcols, const RowOperand &P,
                    const RowOperand &Q, float *partial) {
  constexpr bool ok = wgrad_bf16_ok<WN, WK, TN, TK, VP, VQ>();
  if constexpr (ok && RS_TU_HAS_BF16) {
    if (bf) { launch_wgrad_m<WN, WK, TN, TK, VP, VQ, true>(grid, st, rows, rows_dev, ncols, kcols, P, Q, partial); return; }
  }
  (void)bf;
  if constexpr (!RS_TU_BF16_ONLY || !ok)      // (the bf16-only unit needs the fp32 instance only where bf16 staging is impossible)
    launch_wgrad_m<WN, WK, TN, TK, VP, VQ, false>(grid, st, rows, rows_dev, ncols, kcols, P, Q, partial);
}
template <int WN, int WK, int TN, int TK>
void launch_wgrad(bool bf, int vp, int vq, dim3 grid, hipStream_t st, long long rows, const int *rows_dev, int ncols, int kcols, const RowOperand &P,
                  const RowOperand &Q, float *partial) {
  if (vp == 1 || vq == 1) launch_wgrad_p<WN, WK, TN, TK, 1, 1>(false, grid, st, rows, rows_dev, ncols, kcols, P, Q, partial);
  else if (vp == 4 && vq == 4) launch_wgrad_p<WN, WK, TN, TK, 4, 4>(bf, grid, st, rows, rows_dev, ncols, kcols, P, Q, partial);
  else if (vp == 4) launch_wgrad_p<WN, WK, TN, TK, 4, 2>(bf, grid, st, rows, rows_dev, ncols, kcols, P, Q, partial);
  else launch_wgrad_p<WN, WK, TN, TK, 2, 2>(bf, grid, st, rows, rows_dev, ncols, kcols, P, Q, partial);   // (2,4) runs as (2,2)
}

}  // namespace

#if RS_MLP_TU == 0
extern "C" int rs_sp_gemm_rows(long long, const int *, int, int, const rs_row_operand *, const float *, int, const rs_mlp_epilogue *, void *);
extern "C" int rs_sp_wgrad(long long, const int *, int, int, const rs_row_operand *, const rs_row_operand *, float *, int, float *, void *);
static bool split3_on() { static const int on = env_int("RS_GEMM_SPLIT3", 1); return on != 0; }
extern "C" int rs_mlp_gemm_split3(void) { return split3_on() ? 1 : 0; }
static bool split3_wide_on() { static const int on = env_int("RS_WGRAD_SPLIT3_WIDE", 2); return on != 0; }
#elif RS_MLP_TU == 2
extern "C" int rs_mlp_gemm_split3(void) { return 0; }      // experiment builds carry no unit 4
#endif

static int gemm_rows_impl(bool bf, long long rows, const int *rows_dev, int kdim, int cols, const rs_row_operand *x,
                          const float *w, int ldw, const rs_mlp_epilogue *epi, void *stream) {
  RS_REQUIRE(rows >= 0 && kdim >= 0 && cols >= 0, "rs_mlp_gemm_rows: negative size");
  if (rows == 0 || cols == 0) return RS_OK;
  RS_REQUIRE(kdim > 0, "rs_mlp_gemm_rows: empty reduction dimension");
  RS_REQUIRE(w && epi && epi->out, "rs_mlp_gemm_rows: null pointer");
  RS_REQUIRE(ldw % 4 == 0 && ldw >= kdim && aligned_to(w, 16),
             "rs_mlp_gemm_rows: weights must be n-major (cols x ldw) with a 16-byte aligned base and ldw %% 4 == 0 (ldw=%d, kdim=%d)", ldw, kdim);
  int rc = check_operand("rs_mlp_gemm_rows", x, rows);
  if (rc != RS_OK) return rc;
  Epilogue ep;
  static_cast<rs_mlp_epilogue &>(ep) = *epi;
  ep.w3_rows = (epi->w3 && epi->ldw3 > 0) ? (int)(epi->w3_part / epi->ldw3) : 0;
  RowOperand E = *x;
  if (E.ns <= 0) E.ns = 1;
  const int epi_mode = ep.mode;
  RS_REQUIRE(epi_mode >= EPI_STORE && epi_mode <= EPI_MASK, "rs_mlp_gemm_rows: unknown epilogue %d", epi_mode);
  if (epi_mode != EPI_STORE) RS_REQUIRE(ep.partial && ep.partial_blocks > 0, "rs_mlp_gemm_rows: statistics need a partial buffer");
  if (epi_mode == EPI_MASK) RS_REQUIRE(ep.my1 && ep.ms1 && ep.mt1 && ep.mean1 && ep.invstd1, "rs_mlp_gemm_rows: mask epilogue needs the producing layer's y/scale/shift/mean/invstd");
  if (epi_mode == EPI_MASK && ep.my2) RS_REQUIRE(ep.ms2 && ep.mt2 && ep.mean2 && ep.invstd2, "rs_mlp_gemm_rows: second mask branch incomplete");
  if (epi_mode != EPI_MASK) { ep.my1 = nullptr; ep.my2 = nullptr; }
  // groups of 32 rows are pooled straight from the accumulators wherever the epilogue is the direct one (64-row tiles, 128-row tiles
  // of <= 64 columns); other group sizes, and 128 x 128 tiles, through the C tile in LDS
  const bool pool32 = ep.pool_ns == 32 && epi_mode == EPI_STATS;
  if (ep.pool_ns > 0) {
    RS_REQUIRE(!rows_dev || (pool32 && !ep.row_mult && !E.mult),
               "rs_mlp_gemm_rows: fused pooling needs dense groups (not a compacted row set; a device row count only with groups of 32: whole groups, a multiple of the wave's 32-row tile)");
    RS_REQUIRE(ep.pool_max && ep.pool_min && ep.pool_amax && ep.pool_amin, "rs_mlp_gemm_rows: fused pooling needs its four outputs");
    const int rpt = GM_BM / (GM_THREADS / (cols <= 32 ? 32 : (cols <= 64 ? 64 : 128)));   // (pooling keeps 128-wide tiles)
    RS_REQUIRE(rows % ep.pool_ns == 0 && (pool32 || rpt % ep.pool_ns == 0),
               "rs_mlp_gemm_rows: fused pooling needs nsample (%d) to divide %d rows per thread", ep.pool_ns, rpt);
  }
  if (RS_STORE_BF16) {       // unit 3: the call's flags must be the storage-role table (top of this file)
    const bool masked = epi_mode == EPI_MASK;
    RS_REQUIRE((E.a_bf16 != 0) == sb_a(E.mode) && (E.mode == OPM_ID || E.mode == OPM_RELU1 || E.mode == OPM_BCAST || (E.b_bf16 != 0) == sb_b(E.mode)) &&
               (ep.out_bf16 != 0) == sb_out(E.mode) && (!masked || (ep.my1_bf16 && (!ep.my2 || ep.my2_bf16))) && !(masked && sb_out(E.mode)),
               "rs_mlp_gemm_rows_bf16: bf16 storage needs y tensors (RELU1/RELU2 a, every b, masks, forward outputs) bf16 and the rest fp32 "
               "(mode %d, a %d b %d out %d my1 %d my2 %d)", E.mode, E.a_bf16, E.b_bf16, ep.out_bf16, ep.my1_bf16, ep.my2_bf16);
  } else {
    RS_REQUIRE(!(E.a_bf16 || E.b_bf16 || ep.out_bf16 || ep.my1_bf16 || ep.my2_bf16), "rs_mlp_gemm_rows: bf16 tensors are taken by rs_mlp_gemm_rows_bf16 only");
  }
  hipStream_t st0 = (hipStream_t)stream;
  // kdim <= 16, plain operand, rows on 16- or 8-byte boundaries, fp32 output: the LDS-free wave-per-32-rows kernel -- odd kdim
  // included (round 4: the 3 position channels of the segmentation rows went to the streaming kernel below first, 29.5 us for 67 MB)
  static const int narrow_on = env_int("RS_GEMM_NARROW", 1);
  if (narrow_on && !RS_STORE_BF16 && !bf && kdim <= 16 && cols <= 128 && E.mode == OPM_ID && epi_mode != EPI_MASK && ep.pool_ns == 0) {
    const int kl = kdim <= 8 ? 1 : 2;
    if (aligned_to(E.a, 8) && E.lda % 2 == 0 && E.lda >= 4 && rows > 0) {
      int gxn = (int)((rows + 127) / 128);
      if (gxn > 1024) gxn = 1024;
      if (epi_mode != EPI_STORE && gxn > ep.partial_blocks) gxn = ep.partial_blocks;
      if (gxn < 1) gxn = 1;
      const dim3 grid(gxn);
      const int ct = cols <= 32 ? 1 : (cols <= 64 ? 2 : 4);
#define RS_GN(CT_, KL_) hipLaunchKernelGGL((gemm_narrow_kernel<CT_, KL_>), grid, dim3(GM_THREADS), 0, st0, rows, rows_dev, kdim, cols, E, w, ldw, ep)
      if (kl == 1) { if (ct == 1) RS_GN(1, 1); else if (ct == 2) RS_GN(2, 1); else RS_GN(4, 1); }
      else { if (ct == 1) RS_GN(1, 2); else if (ct == 2) RS_GN(2, 2); else RS_GN(4, 2); }
#undef RS_GN
      RS_CHECK_LAUNCH("rs_mlp_gemm_rows");
      return RS_OK;
    }
  }
  static const int small_on = env_int("RS_GEMM_SMALL", 1);
  // Only where the MFMA kernel would have to take its scalar-load generic instance (an odd kdim or an unaligned
  // operand, e.g. the 3 position channels of a 19-channel row: 166 us against 29 us at 524288 rows); with float2 /
  // float4 operands the MFMA kernel is the faster one even at kdim = 6 (14 us against 15-18 us at 66 584 rows).
  if (small_on && kdim <= WS_KP && E.mode == OPM_ID && epi_mode != EPI_MASK && ep.pool_ns == 0 && pick_vec(E, kdim) == 1) {
    const int nb = cols <= 32 ? 32 : (cols <= 64 ? 64 : 128);
    int gxs = (int)((rows + 63) / 64);
    if (gxs > 512) gxs = 512;
    if (epi_mode != EPI_STORE && gxs > ep.partial_blocks) gxs = ep.partial_blocks;
    if (gxs < 1) gxs = 1;
    const dim3 grid(gxs, rs_cdiv(cols, nb));
    if (nb == 32) hipLaunchKernelGGL(gemm_small_kernel<32>, grid, dim3(GM_THREADS), 0, st0, rows, rows_dev, kdim, cols, E, w, ldw, ep);
    else if (nb == 64) hipLaunchKernelGGL(gemm_small_kernel<64>, grid, dim3(GM_THREADS), 0, st0, rows, rows_dev, kdim, cols, E, w, ldw, ep);
    else hipLaunchKernelGGL(gemm_small_kernel<128>, grid, dim3(GM_THREADS), 0, st0, rows, rows_dev, kdim, cols, E, w, ldw, ep);
    RS_CHECK_LAUNCH("rs_mlp_gemm_rows");
    return RS_OK;
  }
  const int v = pick_vec(E, kdim);
#if RS_MLP_TU == 0
  // the launches of the tiled kernel with vector operands run unit 4's split-product instances (RS_GEMM_SPLIT3=0: the fp32 MFMA ones below)
  // ... when the caller handed the weights' three-part image along (rs_mlp_epilogue.w3, written by rs_pack_weights); without one: fp32 MFMAs
  if (!bf && split3_on() && v >= 2 && epi->w3) {
    RS_REQUIRE(epi->ldw3 >= ((kdim + 31) & ~31) && epi->ldw3 % 32 == 0 && epi->w3_part >= (long long)cols * epi->ldw3 && epi->w3_part % epi->ldw3 == 0 && aligned_to(epi->w3, 16),
               "rs_mlp_gemm_rows: the weights' three-part image needs ldw3 (%d) a multiple of 32 >= kdim (%d), parts (%lld elements) of whole rows >= cols (%d) and a 16-byte aligned base",
               epi->ldw3, kdim, epi->w3_part, cols);
    return rs_sp_gemm_rows(rows, rows_dev, kdim, cols, x, w, ldw, epi, stream);
  }
#endif
  // tile height: 64-row tiles (2 x 2 waves) wherever the layout allows -- twice the workgroups, three per CU
  static const int bm64_on = env_int("RS_GEMM_BM64", 1);
  int bm = (bm64_on && cols > 32 && v >= 2 && (ep.pool_ns == 0 || pool32)) ? 64 : GM_BM;
  long long tiles = (rows + bm - 1) / bm;
  int bn = cols <= 32 ? 32 : (cols <= 64 ? 64 : 128);
  static const int bn_small = env_int("RS_GEMM_BN64_BELOW", 256), bn_small64 = env_int("RS_GEMM_BN64_BELOW64", 512);
  if (bn == 128 && tiles * rs_cdiv(cols, 128) < (bm == 64 ? bn_small64 : bn_small) && ep.pool_ns == 0) bn = 64;   // few rows (group_all stage): 2x the workgroups
  static const int gtail64 = env_int("RS_GEMM_TAIL64", 0);      // a ragged last column tile (138 = 128 + 10): 3 x 64 columns of work instead of 2 x 128 (measured: no gain, off by default)
  if (gtail64 && bn == 128 && bm == 64 && ep.pool_ns == 0 && (cols % 128) != 0 && (cols % 128) <= 64) bn = 64;
  static const int bn32_below = env_int("RS_GEMM_BN32_BELOW", 256);
  if (bm == GM_BM && bn == 64 && cols > 64 && tiles * rs_cdiv(cols, 64) < bn32_below && ep.pool_ns == 0) bn = 32;   // still under one workgroup per CU: 4096 x 512 -> 256 runs 22 us instead of 30
  // (32-row tiles -- 1 x 4 waves, 42 KB of LDS, 168 VGPRs for three workgroups per CU -- were measured for the compacted
  // sa1 / sa2 launches, whose 1043 / 753 tiles of 64 rows are 3 / 2 rounds on 512 workgroups with a mostly empty last
  // round: 1.86 ms/step against 1.83, 768 / 1024 / 512 slots alike.  The per-tile fixed costs outweigh the finer rounds.)
  const int tiles_n = rs_cdiv(cols, bn);
  int gx = persistent_blocks(tiles, tiles_n, bm, E.mode <= OPM_RELU2);
  if (epi_mode != EPI_STORE) gx = gx < ep.partial_blocks ? gx : ep.partial_blocks;
  const dim3 grid(gx, tiles_n);
  hipStream_t st = (hipStream_t)stream;
  if (bm == 64) {
    if (bn == 64) launch_gemm<64, 64>(bf, v, grid, st, rows, rows_dev, kdim, cols, E, w, ldw, ep);
    else launch_gemm<64, 128>(bf, v, grid, st, rows, rows_dev, kdim, cols, E, w, ldw, ep);
  } else if (bn == 32) launch_gemm<128, 32>(bf, v, grid, st, rows, rows_dev, kdim, cols, E, w, ldw, ep);
  else if (bn == 64) launch_gemm<128, 64>(bf, v, grid, st, rows, rows_dev, kdim, cols, E, w, ldw, ep);
  else launch_gemm<128, 128>(bf, v, grid, st, rows, rows_dev, kdim, cols, E, w, ldw, ep);
  if (g_gemm_refused) { g_gemm_refused = false; return RS_ERR_ARG; }
  RS_CHECK_LAUNCH("rs_mlp_gemm_rows");
  return RS_OK;
}

#if !RS_TU_BF16_ONLY
extern "C" int rs_mlp_gemm_rows(long long rows, const int *rows_dev, int kdim, int cols, const rs_row_operand *x,
                                const float *w, int ldw, const rs_mlp_epilogue *epi, void *stream) {
  return gemm_rows_impl(false, rows, rows_dev, kdim, cols, x, w, ldw, epi, stream);
}
#endif
#if RS_MLP_TU == 4
// the split-product instances behind the fp32 entry point of unit 0 (not part of the ABI)
extern "C" __attribute__((visibility("hidden"))) int rs_sp_gemm_rows(long long rows, const int *rows_dev, int kdim, int cols, const rs_row_operand *x,
                                                                     const float *w, int ldw, const rs_mlp_epilogue *epi, void *stream) {
  return gemm_rows_impl(true, rows, rows_dev, kdim, cols, x, w, ldw, epi, stream);
}
#elif RS_MLP_TU == 3
// bf16 activation storage: the instances of this unit behind the public entry point of unit 1 (not part of the ABI)
extern "C" __attribute__((visibility("hidden"))) int rs_sb_gemm_rows(long long rows, const int *rows_dev, int kdim, int cols, const rs_row_operand *x,
                                                                     const float *w, int ldw, const rs_mlp_epilogue *epi, void *stream) {
  return gemm_rows_impl(true, rows, rows_dev, kdim, cols, x, w, ldw, epi, stream);
}
#elif RS_TU_HAS_BF16
#if RS_MLP_TU == 1
extern "C" int rs_sb_gemm_rows(long long, const int *, int, int, const rs_row_operand *, const float *, int, const rs_mlp_epilogue *, void *);
#endif
// Mixed precision (BASELINE configs[4]): same contract, operands rounded to bf16 at the LDS commit, bf16 MFMA with
// fp32 accumulation.  Launches the fp32 instance where the layout forces scalar operand loads or the narrow
// streaming kernel applies (kdim <= 16, unaligned: no matrix pipe involved).  A call that marks tensors as bf16
// (a_bf16 / b_bf16 / out_bf16 / my*_bf16) runs the bf16-storage instances.
extern "C" int rs_mlp_gemm_rows_bf16(long long rows, const int *rows_dev, int kdim, int cols, const rs_row_operand *x,
                                     const float *w, int ldw, const rs_mlp_epilogue *epi, void *stream) {
#if RS_MLP_TU == 1
  if (x && epi && (x->a_bf16 || x->b_bf16 || epi->out_bf16 || epi->my1_bf16 || epi->my2_bf16))
    return rs_sb_gemm_rows(rows, rows_dev, kdim, cols, x, w, ldw, epi, stream);
#endif
  return gemm_rows_impl(true, rows, rows_dev, kdim, cols, x, w, ldw, epi, stream);
}

#endif

static int wgrad_impl(bool bf, long long rows, const int *rows_dev, int ncols, int kcols, const rs_row_operand *p,
                      const rs_row_operand *q, float *partial, int chunks, float *dw, void *stream) {
  RS_REQUIRE(rows >= 0 && ncols >= 0 && kcols >= 0 && chunks > 0, "rs_mlp_wgrad: bad size");
  if (ncols == 0 || kcols == 0) return RS_OK;
  RS_REQUIRE(partial, "rs_mlp_wgrad: null pointer");
  int rc = check_operand("rs_mlp_wgrad(P)", p, rows);
  if (rc != RS_OK) return rc;
  rc = check_operand("rs_mlp_wgrad(Q)", q, rows);
  if (rc != RS_OK) return rc;
  RowOperand P = *p, Q = *q;
  if (RS_STORE_BF16) {
    auto role_ok = [](const RowOperand &o) {
      return (o.a_bf16 != 0) == sb_a(o.mode) && (o.mode == OPM_ID || o.mode == OPM_RELU1 || o.mode == OPM_BCAST || (o.b_bf16 != 0) == sb_b(o.mode));
    };
    RS_REQUIRE(role_ok(P) && role_ok(Q), "rs_mlp_wgrad_bf16: bf16 storage needs y tensors (RELU1/RELU2 a, every b) bf16 and the rest fp32 "
               "(P mode %d a %d b %d, Q mode %d a %d b %d)", P.mode, P.a_bf16, P.b_bf16, Q.mode, Q.a_bf16, Q.b_bf16);
  } else {
    RS_REQUIRE(!(P.a_bf16 || P.b_bf16 || Q.a_bf16 || Q.b_bf16), "rs_mlp_wgrad: bf16 tensors are taken by rs_mlp_wgrad_bf16 only");
  }
  if (P.ns <= 0) P.ns = 1;
  if (Q.ns <= 0) Q.ns = 1;
  hipStream_t st = (hipStream_t)stream;
  const int vp = pick_vec(P, ncols), vq = pick_vec(Q, kcols);
  static const int small_on = env_int("RS_WGRAD_SMALL", 1);
  static const int wnarrow_on = env_int("RS_WGRAD_NARROW", 1);
  static const int wide_form = env_int("RS_WGRAD_SPLIT3_WIDE", 2);      // (unit 4) 2: the 128 x 128 block on one LDS stage; 1: 128 x 64 blocks
  // a ragged last block (138 = 128 + 10 columns of Q: the 10 normal channels next to 128 features) costs a whole 128-wide block:
  // 64-wide blocks then do 192 columns' work instead of 256 (RS_WGRAD_TAIL64=1; measured round 6, profiles/r06/tail64_ab.txt: no gain -- these launches are not bound by the matrix pipe -- so off by default)
  static const int tail64 = env_int("RS_WGRAD_TAIL64", 0);
  const bool ragged_tail = tail64 && RS_SPLIT && kcols > 64 && (kcols % 128) != 0 && (kcols % 128) <= 64;
  // (kcols <= 32 with <= 64 columns of P through this kernel -- the 32 / 64-column layers of the segmentation step's 524 288-row
  // stage, whose 32 x 32 / 64 x 32 products keep one or two of the tiled kernel's four waves on the matrix pipe -- was measured in
  // round 4: 3.72 against 3.67 ms per step; its one-float-per-lane loads cost more than the idle waves.)
  if (wnarrow_on && !RS_STORE_BF16 && !bf && kcols <= WS_KP && (Q.mode == OPM_ID || Q.mode == OPM_RELU1) &&
      (P.mode == OPM_AFF2 || P.mode == OPM_POOLED || P.mode == OPM_BCAST || P.mode == OPM_ID)) {
    // narrow gradient on the matrix pipe (rows = MFMA k), 64 columns of P per workgroup
    const int ct = ncols <= 32 ? 1 : 2;
    const dim3 grid(chunks, rs_cdiv(ncols, 32 * ct));
#define RS_WN(CT_, PM_, QM_) hipLaunchKernelGGL((wgrad_narrow_kernel<CT_, PM_, QM_>), grid, dim3(GM_THREADS), 0, st, rows, rows_dev, ncols, kcols, P, Q, partial)
#define RS_WNQ(CT_, PM_) do { if (Q.mode == OPM_ID) RS_WN(CT_, PM_, OPM_ID); else RS_WN(CT_, PM_, OPM_RELU1); } while (0)
#define RS_WNP(CT_) do { if (P.mode == OPM_AFF2) RS_WNQ(CT_, OPM_AFF2); else if (P.mode == OPM_POOLED) RS_WNQ(CT_, OPM_POOLED); \
                         else if (P.mode == OPM_BCAST) RS_WNQ(CT_, OPM_BCAST); else RS_WNQ(CT_, OPM_ID); } while (0)
    if (ct == 1) RS_WNP(1); else RS_WNP(2);
#undef RS_WNP
#undef RS_WNQ
#undef RS_WN
  } else
  if (small_on && kcols <= WS_KP && vp == 4 && (Q.mode == OPM_ID || Q.mode == OPM_RELU1) &&
      (P.mode == OPM_AFF2 || P.mode == OPM_POOLED || P.mode == OPM_BCAST || P.mode == OPM_ID)) {
    // narrow gradient (first-layer branches): streaming kernel, no matrix pipe
    const int nb = ncols <= 32 ? 32 : (ncols <= 64 ? 64 : 128);
    const dim3 grid(chunks, rs_cdiv(ncols, nb));
    const size_t lds = sizeof(float) * (size_t)(GM_THREADS / (nb / 4)) * nb * WS_KP;
#define RS_WS(NB_, PM_, QM_) hipLaunchKernelGGL((wgrad_small_kernel<NB_, 4, PM_, QM_>), grid, dim3(GM_THREADS), lds, st, rows, rows_dev, ncols, kcols, P, Q, partial)
#define RS_WSQ(NB_, PM_) do { if (Q.mode == OPM_ID) RS_WS(NB_, PM_, OPM_ID); else RS_WS(NB_, PM_, OPM_RELU1); } while (0)
#define RS_WSP(NB_) do { if (P.mode == OPM_AFF2) RS_WSQ(NB_, OPM_AFF2); else if (P.mode == OPM_POOLED) RS_WSQ(NB_, OPM_POOLED); \
                         else if (P.mode == OPM_BCAST) RS_WSQ(NB_, OPM_BCAST); else RS_WSQ(NB_, OPM_ID); } while (0)
    if (nb == 32) RS_WSP(32); else if (nb == 64) RS_WSP(64); else RS_WSP(128);
#undef RS_WSP
#undef RS_WSQ
#undef RS_WS
  } else
#if RS_MLP_TU == 0
  // (products wider than 64 columns of Q, RS_WGRAD_SPLIT3_WIDE: 2 (default) unit 4's 128 x 128 output block on ONE LDS stage -- two stages
  // of three parts are 96 KB, one workgroup per CU: 83 -> 126 us at 4096 x 1024 x 512; one stage and a second barrier per 32 rows: 76 us,
  // classification step 1.342 -> 1.298 ms; 1: unit 4's 128 x 64 blocks, 89 us; 0: the fp32 instances)
  if (!bf && split3_on() && vp >= 2 && vq >= 2 && (kcols <= 64 || split3_wide_on())) {
    return rs_sp_wgrad(rows, rows_dev, ncols, kcols, p, q, partial, chunks, dw, stream);
  } else
#endif
  if (kcols > 64 && (!RS_SPLIT || wide_form == 2) && !ragged_tail) {          // 128 x 128 output block: waves 2 x 2, 2 x 2 tiles each
      launch_wgrad<2, 2, 2, 2>(bf, vp, vq, dim3(chunks, rs_cdiv(ncols, 128), rs_cdiv(kcols, 128)), st, rows, rows_dev, ncols, kcols, P, Q, partial);
  } else if (kcols > 32) {   // 128 x 64: waves 4 x 1, 1 x 2 tiles (unit 4: also the wider products, 64 columns of Q per workgroup)
    launch_wgrad<4, 1, 1, 2>(bf, vp, vq, dim3(chunks, rs_cdiv(ncols, 128), rs_cdiv(kcols, 64)), st, rows, rows_dev, ncols, kcols, P, Q, partial);
  } else {                   // 128 x 32: waves 4 x 1, 1 x 1 tile
    launch_wgrad<4, 1, 1, 1>(bf, vp, vq, dim3(chunks, rs_cdiv(ncols, 128), 1), st, rows, rows_dev, ncols, kcols, P, Q, partial);
  }
  if (dw) {      // dw == NULL: the caller reduces `partial` itself (rs_reduce_partials / rs_bn_backward_finalize_reduce)
    const long long n = (long long)ncols * kcols;
    long long rb = (n + 31) / 32;
    if (rb > 2048) rb = 2048;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((int)rb), dim3(GM_THREADS), 0, st, chunks, n, partial, dw);
  }
  RS_CHECK_LAUNCH("rs_mlp_wgrad");
  return RS_OK;
}

#if !RS_TU_BF16_ONLY
extern "C" int rs_mlp_wgrad(long long rows, const int *rows_dev, int ncols, int kcols, const rs_row_operand *p,
                            const rs_row_operand *q, float *partial, int chunks, float *dw, void *stream) {
  return wgrad_impl(false, rows, rows_dev, ncols, kcols, p, q, partial, chunks, dw, stream);
}
#endif
#if RS_MLP_TU == 4
extern "C" __attribute__((visibility("hidden"))) int rs_sp_wgrad(long long rows, const int *rows_dev, int ncols, int kcols, const rs_row_operand *p,
                                                                 const rs_row_operand *q, float *partial, int chunks, float *dw, void *stream) {
  return wgrad_impl(true, rows, rows_dev, ncols, kcols, p, q, partial, chunks, dw, stream);
}
#elif RS_MLP_TU == 3
extern "C" __attribute__((visibility("hidden"))) int rs_sb_wgrad(long long rows, const int *rows_dev, int ncols, int kcols, const rs_row_operand *p,
                                                                 const rs_row_operand *q, float *partial, int chunks, float *dw, void *stream) {
  return wgrad_impl(true, rows, rows_dev, ncols, kcols, p, q, partial, chunks, dw, stream);
}
#elif RS_TU_HAS_BF16
#if RS_MLP_TU == 1
extern "C" int rs_sb_wgrad(long long, const int *, int, int, const rs_row_operand *, const rs_row_operand *, float *, int, float *, void *);
#endif
// Mixed precision: both operands rounded to bf16 after their fp32 prologue, bf16 MFMA, fp32 accumulation inside a
// row slab; the slabs' partial products and their fixed-order sum stay fp32.  The narrow streaming kernel
// (kcols <= 16) and the layouts that force scalar loads or one vector per thread (kcols <= 32, float4) run in fp32.
// Operands that mark tensors as bf16 run the bf16-storage instances.
extern "C" int rs_mlp_wgrad_bf16(long long rows, const int *rows_dev, int ncols, int kcols, const rs_row_operand *p,
                                 const rs_row_operand *q, float *partial, int chunks, float *dw, void *stream) {
#if RS_MLP_TU == 1
  if (p && q && (p->a_bf16 || p->b_bf16 || q->a_bf16 || q->b_bf16))
    return rs_sb_wgrad(rows, rows_dev, ncols, kcols, p, q, partial, chunks, dw, stream);
#endif
  return wgrad_impl(true, rows, rows_dev, ncols, kcols, p, q, partial, chunks, dw, stream);
}

#endif

#if !RS_TU_BF16_ONLY      // everything below: fp32 unit only
// ---- padded copies of up to RS_PACK_MAX conv weights (cout, cin) in one launch --------------------------------
// transpose = 0:  dst[j*ld + k] = src[j*cin + k]  (k < cin, else 0), ld >= cin    -- forward operand of rs_mlp_gemm_rows
//                 when cin is not a multiple of 4 (otherwise the conv weight is used in place);
// transpose = 1:  dst[k*ld + j] = src[j*cin + k]  (j < cout, else 0), ld >= cout  -- data-gradient operand (dY . W).
// The three-part image (dst3) is TILE-ORDERED (round 6): dst3[q][k / 8][row][k % 8], row = the operand's n (outer) index, k its reduction
// (inner) index -- 16-byte units of 8 consecutive k, the units of one k-octet contiguous over rows: what one global_load_lds_dwordx4 of the
// row GEMM moves into an LDS plane.  A thread of this kernel writes whole units (three 16-byte stores, consecutive lanes = consecutive rows).
__device__ __forceinline__ void split_store_unit(const float (&v)[8], unsigned short *dst3, long long part, long long unit) {
  unsigned w[3][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float d[3];
    split_bf16(v[2 * i], v[2 * i + 1], d);
#pragma unroll
    for (int q = 0; q < 3; ++q) w[q][i] = __float_as_uint(d[q]);
  }
#pragma unroll
  for (int q = 0; q < 3; ++q)
    *reinterpret_cast<uint4 *>(dst3 + q * part + unit * 8) = make_uint4(w[q][0], w[q][1], w[q][2], w[q][3]);
}

__global__ void __launch_bounds__(GM_THREADS)
pack_weights_kernel(rs_pack_weights_args a) {
  const int e = blockIdx.y;
  const int cout = a.cout[e], cin = a.cin[e], ld = a.ld[e], ld3 = a.ld3[e];
  const float *__restrict__ src = a.src[e];
  float *__restrict__ dst = a.dst[e];
  unsigned short *__restrict__ dst3 = reinterpret_cast<unsigned short *>(a.dst3[e]);
  if (a.transpose[e]) {
    // 32 x 32 tiles through LDS: rows of src are read along k, rows of dst written along j -- both coalesced.  (Element-wise,
    // consecutive lanes read src[j * cin + k] for consecutive j: one 4-byte word from each of 64 cache lines per load, the
    // 512 x 1 024 weight as 32 x its bytes; the launch took 13 us.)
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int wide = dst3 ? max(ld, ld3) : ld;                   // (ld3 = round32(cout) >= ld = round4(cout))
    const int tk = (cin + 31) >> 5, tj = (wide + 31) >> 5;
    const long long part = (long long)cin * ld3;
    for (int t = blockIdx.x; t < tk * tj; t += gridDim.x) {
      const int k0 = (t / tj) << 5, j0 = (t % tj) << 5;
#pragma unroll
      for (int p2 = 0; p2 < 4; ++p2) {
        const int j = j0 + ty + 8 * p2, k = k0 + tx;
        tile[ty + 8 * p2][tx] = (j < cout && k < cin) ? src[j * cin + k] : 0.f;
      }
      __syncthreads();
      if (dst) {
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) {
          const int k = k0 + ty + 8 * p2, j = j0 + tx;
          if (k < cin && j < ld) dst[k * ld + j] = tile[tx][ty + 8 * p2];
        }
      }
      // image rows = k (the data gradient's n), reduction index = j: the tile holds 4 j-octets x 32 rows = 128 units
      if (dst3 && threadIdx.x < 128) {
        const int jo = threadIdx.x >> 5, k = k0 + tx;
        if (k < cin && j0 + 8 * jo < ld3) {
          float v[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = tile[8 * jo + i][tx];
          split_store_unit(v, dst3, part, (long long)((j0 >> 3) + jo) * cin + k);
        }
      }
      __syncthreads();
    }
  } else {
    const long long part = (long long)cout * ld3;
    if (dst) {
      const int total = cout * ld;
      for (int i = blockIdx.x * GM_THREADS + threadIdx.x; i < total; i += gridDim.x * GM_THREADS) {
        const int j = i / ld, k = i - j * ld;
        dst[i] = k < cin ? src[j * cin + k] : 0.f;
      }
    }
    if (dst3) {
      const int units = (ld3 >> 3) * cout;                       // unit = (k-octet ko, row j), rows fastest
      for (int u = blockIdx.x * GM_THREADS + threadIdx.x; u < units; u += gridDim.x * GM_THREADS) {
        const int ko = u / cout, j = u - ko * cout;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (8 * ko + i) < cin ? src[j * cin + 8 * ko + i] : 0.f;
        split_store_unit(v, dst3, part, u);
      }
    }
  }
}

extern "C" int rs_pack_weights(const rs_pack_weights_args *args, void *stream) {
  RS_REQUIRE(args && args->n >= 0 && args->n <= RS_PACK_MAX, "rs_pack_weights: bad descriptor count");
  if (args->n == 0) return RS_OK;
  int biggest = 0;
  for (int e = 0; e < args->n; ++e) {
    const int inner = args->transpose[e] ? args->cout[e] : args->cin[e];
    const int outer = args->transpose[e] ? args->cin[e] : args->cout[e];
    RS_REQUIRE(args->src[e] && (args->dst[e] || args->dst3[e]) && args->ld[e] >= inner && args->ld[e] % 4 == 0,
               "rs_pack_weights: entry %d invalid (ld=%d cout=%d cin=%d transpose=%d)", e, args->ld[e], args->cout[e],
               args->cin[e], args->transpose[e]);
    RS_REQUIRE(!args->dst3[e] || (args->ld3[e] >= args->ld[e] && args->ld3[e] % 32 == 0),
               "rs_pack_weights: entry %d: the split image needs ld3 (%d) >= ld (%d) and ld3 %% 32 == 0", e, args->ld3[e], args->ld[e]);
    const int wide = args->dst3[e] ? max(args->ld[e], args->ld3[e]) : args->ld[e];
    biggest = max(biggest, args->transpose[e] ? 8 * GM_THREADS * (rs_cdiv(args->cin[e], 32) * rs_cdiv(wide, 32)) / 32 : outer * wide);
  }
  int gx = rs_cdiv(biggest, GM_THREADS);        // (a transposed entry: ~a workgroup per four of its 32 x 32 tiles)
  if (gx > 128) gx = 128;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(gx, args->n), dim3(GM_THREADS), 0, (hipStream_t)stream, *args);
  RS_CHECK_LAUNCH("rs_pack_weights");
  return RS_OK;
}

extern "C" int rs_reduce_partials(int nblk, long long n, const float *partial, float *out, void *stream) {
  RS_REQUIRE(nblk > 0 && n >= 0, "rs_reduce_partials: bad size");
  if (n == 0) return RS_OK;
  RS_REQUIRE(partial && out, "rs_reduce_partials: null pointer");
  long long rb = (n + 31) / 32;
  if (rb > 2048) rb = 2048;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((int)rb), dim3(GM_THREADS), 0, (hipStream_t)stream, nblk, n, partial, out);
  RS_CHECK_LAUNCH("rs_reduce_partials");
  return RS_OK;
}

extern "C" int rs_bn_finalize(int c, long long rows, int nblk, const double *partial, const float *gamma,
                              const float *beta, float eps, float momentum, float *scale, float *shift,
                              float *save_mean, float *save_invstd, float *running_mean, float *running_var,
                              void *stream) {
  RS_REQUIRE(c >= 0 && rows > 0 && nblk > 0, "rs_bn_finalize: bad size");
  if (c == 0) return RS_OK;
  RS_REQUIRE(partial && scale && shift && save_mean && save_invstd, "rs_bn_finalize: null pointer");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(rs_cdiv(c, 8)), dim3(256), 0, (hipStream_t)stream, c, rows, nblk,
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
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(rs_cdiv(c, 8)), dim3(256), 0, (hipStream_t)stream, c, rows, nblk,
                     nstat, which, partial, scale, mean, invstd, p, q, r, dgamma, dbeta);
  RS_CHECK_LAUNCH("rs_bn_backward_finalize");
  return RS_OK;
}

extern "C" int rs_bn_backward_finalize_reduce(int c, long long rows, int nblk, int nstat, int which, const double *partial,
                                              const float *scale, const float *mean, const float *invstd, float *p,
                                              float *q, float *r, float *dgamma, float *dbeta, int red_chunks,
                                              long long red_n, const float *red_partial, float *red_out, void *stream) {
  RS_REQUIRE(c > 0 && rows > 0 && nblk > 0 && nstat >= 2 && which >= 1 && which < nstat, "rs_bn_backward_finalize_reduce: bad size");
  RS_REQUIRE(partial && scale && mean && invstd && p && q && r, "rs_bn_backward_finalize_reduce: null pointer");
  RS_REQUIRE(red_chunks > 0 && red_n > 0 && red_partial && red_out, "rs_bn_backward_finalize_reduce: empty reduction");
  const int nfin = rs_cdiv(c, 8);
  long long rb = (red_n + 31) / 32;
  if (rb > 2048) rb = 2048;
  hipLaunchKernelGGL(bn_bwd_finalize_reduce_kernel, dim3(nfin + (int)rb), dim3(256), 0, (hipStream_t)stream, c, rows, nblk,
                     nstat, which, partial, scale, mean, invstd, p, q, r, dgamma, dbeta, nfin, red_chunks, red_n, red_partial,
                     red_out);
  RS_CHECK_LAUNCH("rs_bn_backward_finalize_reduce");
  return RS_OK;
}

extern "C" int rs_backward_tail(const rs_backward_tail_work *work, void *stream) {
  RS_REQUIRE(work, "rs_backward_tail: null pointer");
  rs_backward_tail_work w = *work;
  RS_REQUIRE(w.nfin >= 0 && w.nfin <= RS_TAIL_FIN_MAX && w.nred >= 0 && w.nred <= RS_TAIL_RED_MAX, "rs_backward_tail: %d finalizes / %d reductions exceed %d / %d",
             w.nfin, w.nred, RS_TAIL_FIN_MAX, RS_TAIL_RED_MAX);
  if (w.nfin == 0 && w.nred == 0) return RS_OK;
  TailStarts st = {};
  int at = 0;
  for (int i = 0; i < w.nfin; ++i) {
    const rs_bn_bwd_item &it = w.fin[i];
    RS_REQUIRE(it.c > 0 && it.rows > 0 && it.nblk > 0 && it.nstat >= 2 && it.which >= 1 && it.which < it.nstat, "rs_backward_tail: bad size (finalize %d)", i);
    RS_REQUIRE(it.partial && it.scale && it.mean && it.invstd && it.p && it.q && it.r, "rs_backward_tail: null pointer (finalize %d)", i);
    st.fin[i] = at;
    at += rs_cdiv(it.c, 8);
  }
  for (int i = w.nfin; i <= RS_TAIL_FIN_MAX; ++i) st.fin[i] = at;
  for (int j = 0; j < w.nred; ++j) {
    const rs_reduce_item &it = w.red[j];
    RS_REQUIRE(it.chunks > 0 && it.n > 0 && it.partial && it.out, "rs_backward_tail: empty reduction %d", j);
    long long rb = (it.n + 31) / 32;
    if (rb > 2048 / (w.nred > 1 ? 2 : 1)) rb = 2048 / (w.nred > 1 ? 2 : 1);
    st.red[j] = at;
    at += (int)rb;
  }
  for (int j = w.nred; j <= RS_TAIL_RED_MAX; ++j) st.red[j] = at;
  hipLaunchKernelGGL(backward_tail_kernel, dim3(at), dim3(256), 0, (hipStream_t)stream, w, st);
  RS_CHECK_LAUNCH("rs_backward_tail");
  return RS_OK;
}

extern "C" int rs_bn_finalize_batch(const rs_bn_item *items, int n, void *stream) {
  RS_REQUIRE(items && n >= 0 && n <= RS_BN_BATCH_MAX, "rs_bn_finalize_batch: %d items (at most %d)", n, RS_BN_BATCH_MAX);
  if (n == 0) return RS_OK;
  BnBatch w = {};
  int at = 0;
  for (int i = 0; i < n; ++i) {
    const rs_bn_item &it = items[i];
    RS_REQUIRE(it.c > 0 && it.rows > 0 && it.nblk > 0, "rs_bn_finalize_batch: bad size (item %d)", i);
    RS_REQUIRE(it.partial && it.scale && it.shift && it.save_mean && it.save_invstd, "rs_bn_finalize_batch: null pointer (item %d)", i);
    RS_REQUIRE((it.running_mean == nullptr) == (it.running_var == nullptr), "rs_bn_finalize_batch: running_mean and running_var come together (item %d)", i);
    w.it[i] = it;
    w.start[i] = at;
    at += rs_cdiv(it.c, 8);
  }
  for (int i = n; i <= RS_BN_BATCH_MAX; ++i) w.start[i] = at;
  w.n = n;
  hipLaunchKernelGGL(bn_finalize_batch_kernel, dim3(at), dim3(256), 0, (hipStream_t)stream, w);
  RS_CHECK_LAUNCH("rs_bn_finalize_batch");
  return RS_OK;
}

extern "C" int rs_pool_max(long long groups, int nsample, int c, int relu, const int *offsets, const float *y,
                           int y_bf16, const float *scale, const float *shift, float *out, int *arg, void *stream) {
  RS_REQUIRE(groups >= 0 && nsample > 0 && c >= 0, "rs_pool_max: bad size");
  if (groups == 0 || c == 0) return RS_OK;
  RS_REQUIRE(y && out && (arg || nsample == 1), "rs_pool_max: null pointer (arg may be NULL for nsample = 1 only)");
  if (!offsets && nsample >= 64 && groups <= 65535 && groups * c <= (1 << 18)) {
    if (y_bf16) hipLaunchKernelGGL(pool_max_long_kernel<true>, dim3((int)groups, rs_cdiv(c, 64)), dim3(GM_THREADS), 0, (hipStream_t)stream,
                                   nsample, c, relu, y, scale, shift, out, arg);
    else hipLaunchKernelGGL(pool_max_long_kernel<false>, dim3((int)groups, rs_cdiv(c, 64)), dim3(GM_THREADS), 0, (hipStream_t)stream,
                            nsample, c, relu, y, scale, shift, out, arg);
    RS_CHECK_LAUNCH("rs_pool_max");
    return RS_OK;
  }
  long long blocks = (groups * c + GM_THREADS - 1) / GM_THREADS;
  if (blocks > 2048) blocks = 2048;
  if (y_bf16) hipLaunchKernelGGL(pool_max_kernel<true>, dim3((int)blocks), dim3(GM_THREADS), 0, (hipStream_t)stream, groups, nsample, c,
                                 relu, offsets, y, scale, shift, out, arg);
  else hipLaunchKernelGGL(pool_max_kernel<false>, dim3((int)blocks), dim3(GM_THREADS), 0, (hipStream_t)stream, groups, nsample, c,
                          relu, offsets, y, scale, shift, out, arg);
  RS_CHECK_LAUNCH("rs_pool_max");
  return RS_OK;
}

extern "C" int rs_pool_select(long long groups, int c, const float *ymax, const float *ymin, const int *amax,
                              const int *amin, const float *scale, const float *shift, float *out, int *arg,
                              void *stream) {
  RS_REQUIRE(groups >= 0 && c >= 0, "rs_pool_select: bad size");
  if (groups == 0 || c == 0) return RS_OK;
  RS_REQUIRE(ymax && ymin && amax && amin && scale && shift && out && arg, "rs_pool_select: null pointer");
  long long blocks = (groups * c + GM_THREADS - 1) / GM_THREADS;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pool_select_kernel, dim3((int)blocks), dim3(GM_THREADS), 0, (hipStream_t)stream, groups, c, ymax, ymin,
                     amax, amin, scale, shift, out, arg);
  RS_CHECK_LAUNCH("rs_pool_select");
  return RS_OK;
}

extern "C" int rs_pool_max_backward(long long groups, int nsample, int c, const int *offsets, const float *dout, long long ldd,
                                    const float *out, const int *arg, const float *y, int y_bf16, const float *mean,
                                    const float *invstd, float *v, double *partial, int partial_blocks, const int *groups_dev, void *stream) {
  RS_REQUIRE(groups >= 0 && nsample > 0 && c >= 0 && partial_blocks > 0, "rs_pool_max_backward: bad size");
  if (groups == 0 || c == 0) return RS_OK;
  RS_REQUIRE(dout && (arg || (nsample == 1 && !offsets)) && y && mean && invstd && (v || !out) && partial, "rs_pool_max_backward: null pointer (arg may be NULL for dense groups of one row only; v for a layer without ReLU: it would equal dout)");
  if (ldd <= 0) ldd = c;
  RS_REQUIRE(ldd >= c, "rs_pool_max_backward: row stride of dout %lld < %d channels", ldd, c);
  const long long want = (groups + 3) / 4;
  int gx = (int)(want < partial_blocks ? want : partial_blocks);
  hipStream_t st = (hipStream_t)stream;
  if (y_bf16) hipLaunchKernelGGL(pool_max_bwd_kernel<true>, dim3(gx, rs_cdiv(c, 64)), dim3(GM_THREADS), 0, st, groups, groups_dev, nsample, c,
                                 offsets, dout, ldd, out, arg, y, mean, invstd, v, partial, partial_blocks);
  else hipLaunchKernelGGL(pool_max_bwd_kernel<false>, dim3(gx, rs_cdiv(c, 64)), dim3(GM_THREADS), 0, st, groups, groups_dev, nsample, c,
                          offsets, dout, ldd, out, arg, y, mean, invstd, v, partial, partial_blocks);
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
#endif   // !RS_TU_BF16_ONLY
