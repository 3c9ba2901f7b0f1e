// fps.hip — farthest point sampling for gfx950 (wave64).
//
// What it computes: farthest_point_sample(cuda=False) of the reference
// (classification/modules/pointnet2_utils.py:47-75), i.e.
//   picks[0] = start;  dist[:] = 1e10
//   for i in 0..m-1:  out[i] = cur
//                     d[k]   = ((x_k-cx)^2 + (y_k-cy)^2) + (z_k-cz)^2     (products rounded separately)
//                     dist[k] = d[k] < dist[k] ? d[k] : dist[k]
//                     cur    = lowest k with dist[k] == max(dist)
// The reference CUDA kernel (classification/modules/pointops/src/sampling/sampling_cuda_kernel.cu:58-168)
// keeps dist in global memory (re-read every iteration), reduces through a 10-level shared
// memory tree with a __syncthreads per level and resolves ties by thread id.  Here:
//   * one workgroup per cloud; every thread owns PPT *consecutive* points whose coordinates and
//     running distance live in registers for the whole launch (no global / LDS traffic in the loop);
//   * the per-iteration arg-max is: per-lane max -> wave max by DPP + gfx950 permlane swaps
//     (rs_wave_max_u32 on the float bit pattern, distances are >= 0) -> v_cmp ballot ->
//     s_ff1 picks the lowest lane -> v_readlane fetches that lane's slot and coordinates.
//     Because points are laid out blocked (thread t owns [t*PPT, (t+1)*PPT)), "lowest lane,
//     lowest slot" IS "lowest point index" -- the reference CPU tie rule -- for free;
//   * with more than one wave, each wave publishes (max, index, xyz) to a double-buffered LDS
//     slot and ONE s_barrier per iteration separates publish from the (redundant, uniform)
//     cross-wave pick; with one wave there is no barrier and no LDS at all.
// It is a latency chain of m-1 dependent reductions, not an HBM- or MFMA-bound kernel;
// DESIGN.md §5 gives the per-iteration budget.
#include "rs_common.h"
#include <math.h>
#include <stdlib.h>

namespace {

struct FpsSeg {   // which rows of xyz / idx this workgroup owns
  int row0;       // first point row
  int n;          // number of points
  int out0;       // first output slot
  int m;          // number of picks
  int start;      // local index of the first pick
  int idx_base;   // added to local indices on output (packed batches emit global rows)
};

__device__ __forceinline__ FpsSeg fps_segment(int blk, int n, int m, const int *start,
                                              const int *offset, const int *new_offset) {
  FpsSeg s;
  if (offset) {   // packed batch (segmentation): rows [offset[blk-1], offset[blk])
    s.row0 = blk ? offset[blk - 1] : 0;
    s.n = offset[blk] - s.row0;
    s.out0 = blk ? new_offset[blk - 1] : 0;
    s.m = new_offset[blk] - s.out0;
    s.start = 0;
    s.idx_base = s.row0;
  } else {
    s.row0 = blk * n;
    s.n = n;
    s.out0 = blk * m;
    s.m = m;
    s.start = start ? start[blk] : 0;
    s.idx_base = 0;
  }
  return s;
}

// signed max over the wave (distance bit patterns are >= 0, padding slots hold -1.0f = a negative integer)
__device__ __forceinline__ int rs_wave_max_i32(int v) {
  v = max(v, (int)rs_dpp<RS_DPP_QUAD_XOR1>((unsigned)v));
  v = max(v, (int)rs_dpp<RS_DPP_QUAD_XOR2>((unsigned)v));
  v = max(v, (int)rs_dpp<RS_DPP_ROW_HALF_MIRROR>((unsigned)v));
  v = max(v, (int)rs_dpp<RS_DPP_ROW_MIRROR>((unsigned)v));
  auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
  v = max((int)r[0], (int)r[1]);
  auto s = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
  v = max((int)s[0], (int)s[1]);
  return v;
}

// Priority position p = tid * PPT + j ("lowest lane, lowest slot" wins a tie) -> local point index.
//   tie_bs == 0 (classification, torch.max rule): p itself -- lowest index among equal distances.
//   tie_bs  > 0 (packed batches): what the reference kernel's strided scan + shared-memory tree computes
//   (segmentation/modules/pointops/src/sampling/sampling_cuda_kernel.cu:44-58 and __update :7-12; pinned against
//   the kernel itself in tests/test_oracle_ref.py): thread t of `tie_bs` scans rows t, t + bs, ... keeping the first
//   maximum, and the tree keeps the lower slot at every level, i.e. the thread with the lowest BIT-REVERSED id wins.
//   Position p = rank * q + jj enumerates (bit-reversed rank, scan order): point = bitrev(rank) + jj * bs.
__device__ __forceinline__ int fps_point_of(int p, int tie_bs, int tie_q, int tie_shift) {
  if (tie_bs == 0) return p;
  const int rank = p / tie_q, jj = p - rank * tie_q;
  if (rank >= tie_bs) return 0x3fffffff;
  const int t = tie_bs > 1 ? (int)(__brev((unsigned)rank) >> tie_shift) : 0;
  return t + jj * tie_bs;
}

template <int PPT, int NW, bool TIE>     // points per thread, waves per workgroup (blockDim.x == 64 * NW), packed tie rule
__global__ void __launch_bounds__(64 * NW)
fps_reg_kernel(int n_arg, int m_arg, const float *__restrict__ xyz, const int *__restrict__ start,
               const int *__restrict__ offset, const int *__restrict__ new_offset,
               int *__restrict__ idx_out, int tie_bs, int tie_q, int tie_shift, const int *__restrict__ tie_n_dev, int guard) {
  __shared__ int2 red_key[2][16];     // (max distance bits, point index) per wave, double buffered
  __shared__ float4 red_xyz[2][16];

  const FpsSeg seg = fps_segment(blockIdx.x, n_arg, m_arg, start, offset, new_offset);
  const int n = seg.n, m = seg.m;
  if (m <= 0 || n <= 0) return;
  const float *pts = xyz + (size_t)seg.row0 * 3;
  int *out = idx_out + seg.out0;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  if (TIE && tie_n_dev) {             // sectorized FPS: the largest sector is only known on the device (rs_sectorize)
    const int nr = max(*tie_n_dev, 1);
    const int bits = min(31 - __clz(nr), 10);      // opt_n_threads: min(2^floor(log2 n), 1024)
    tie_bs = 1 << bits;
    tie_q = (nr + tie_bs - 1) >> bits;
    tie_shift = 32 - bits;
    // guard (sectorized FPS whose host-side bound -- the whole cloud -- exceeds this kernel's positions): the launch pairs this kernel
    // with fps_global_kernel, and the largest SECTOR, known on the device only, decides which of the two works (uniformly: the
    // position -> point map depends on the largest sector, not on this workgroup's)
    if (guard && tie_bs * tie_q > PPT * 64 * NW) return;
  }

  float px[PPT], py[PPT], pz[PPT], md[PPT];
  constexpr bool TABLE = TIE && PPT <= 16;       // TIE: the point each slot holds, as a table up to 16 points per lane; above that the ONE lane
  int kid[TABLE ? PPT : 1];                      // that publishes recomputes it (an integer division per pick: 1.21 -> 1.30 us per pick at 4 096 rows)
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int p = TIE ? fps_point_of(tid * PPT + j, tie_bs, tie_q, tie_shift) : tid * PPT + j;
    if constexpr (TABLE) kid[j] = p;
    const int pl = p < n ? p : 0;     // padding: a real point's coordinates, distance pinned at -1 -> below every real
    px[j] = pts[pl * 3 + 0];          // distance (>= +0) in the signed comparison, so it never wins, not even a tie at 0
    py[j] = pts[pl * 3 + 1];
    pz[j] = pts[pl * 3 + 2];
    md[j] = p < n ? 1e10f : -1.0f;    // pointnet2_utils.py:65 / tmp = 1e10 (pointops.py:45)
  }

  int cur = seg.start;
  if (cur < 0 || cur >= n) cur = 0;
  float cx = pts[cur * 3 + 0], cy = pts[cur * 3 + 1], cz = pts[cur * 3 + 2];

  for (int it = 0; it < m; ++it) {
    if (tid == 0) out[it] = cur + seg.idx_base;
    if (it == m - 1) break;

    // distance update + per-lane max.  Two points per instruction where the lane owns an even number of them:
    // v_pk_add_f32 / v_pk_mul_f32 round each half exactly like the scalar forms (no contraction), and the update is
    // the longest VALU stretch of a pick.
    int lmax = (int)0x80000000;
    if constexpr (PPT % 2 == 0) {
      typedef float v2f __attribute__((ext_vector_type(2)));
      const v2f cx2 = {cx, cx}, cy2 = {cy, cy}, cz2 = {cz, cz};
#pragma unroll
      for (int j = 0; j < PPT; j += 2) {
        const v2f x2 = {px[j], px[j + 1]}, y2 = {py[j], py[j + 1]}, z2 = {pz[j], pz[j + 1]};
        const v2f dx = x2 - cx2, dy = y2 - cy2, dz = z2 - cz2;
        const v2f d = (dx * dx + dy * dy) + dz * dz;
        // running minimum on the bit patterns (distances are >= +0, padding is -1.0f: integer order = float order);
        // v_min_i32 instead of the canonicalise + v_min_f32 pair fminf needs
        md[j] = __int_as_float(min(__float_as_int(d.x), __float_as_int(md[j])));
        md[j + 1] = __int_as_float(min(__float_as_int(d.y), __float_as_int(md[j + 1])));
        lmax = max(lmax, max(__float_as_int(md[j]), __float_as_int(md[j + 1])));     // v_max3_i32
      }
    } else {
#pragma unroll
      for (int j = 0; j < PPT; ++j) {
        const float dx = px[j] - cx, dy = py[j] - cy, dz = pz[j] - cz;
        const float d = (dx * dx + dy * dy) + dz * dz;
        md[j] = d < md[j] ? d : md[j];
        lmax = max(lmax, __float_as_int(md[j]));
      }
    }
    const int wmax = __builtin_amdgcn_readfirstlane(rs_wave_max_i32(lmax));

    // lowest slot in this lane that holds the wave max (2 instructions per slot); the point and its coordinates are
    // picked by the ONE lane that publishes them (the selects used to run once in the main flow and, in the packed
    // variant, a second time inside the publishing branch: 64 extra instructions per pick at 16 points per lane)
    int slot = PPT;
#pragma unroll
    for (int j = PPT - 1; j >= 0; --j) slot = (__float_as_int(md[j]) == wmax) ? j : slot;
    const unsigned long long cand = __ballot(slot < PPT);
    const int wl = __ffsll((long long)cand) - 1;   // lowest lane holding the max (cand != 0 always)
    auto pick = [&](int &sk, float &sx, float &sy, float &sz) {
      sk = TABLE ? kid[0] : (TIE ? fps_point_of(tid * PPT + slot, tie_bs, tie_q, tie_shift) : tid * PPT + slot); sx = px[0]; sy = py[0]; sz = pz[0];
#pragma unroll
      for (int j = 1; j < PPT; ++j) {
        const bool hit = slot == j;
        if constexpr (TABLE) sk = hit ? kid[j] : sk;
        sx = hit ? px[j] : sx; sy = hit ? py[j] : sy; sz = hit ? pz[j] : sz;
      }
    };

    if (NW == 1) {
      int sk; float sx, sy, sz;
      pick(sk, sx, sy, sz);
      cur = __builtin_amdgcn_readlane(sk, wl);
      cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sx), wl));
      cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sy), wl));
      cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sz), wl));
    } else {
      // the winning lane publishes its candidate itself (no v_readlane round trips), one barrier, then every lane
      // fetches the NW candidates with independent broadcast LDS reads and compares them in registers
      const int par = it & 1;
      if (lane == wl) {
        int sk; float sx, sy, sz;
        pick(sk, sx, sy, sz);
        red_key[par][wave] = make_int2(wmax, sk);
        red_xyz[par][wave] = make_float4(sx, sy, sz, 0.f);
      }
      __syncthreads();
      int2 k[NW]; float4 c[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) { k[w] = red_key[par][w]; c[w] = red_xyz[par][w]; }
      int bkey = k[0].x, bidx = k[0].y; float bx = c[0].x, by = c[0].y, bz = c[0].z;
#pragma unroll
      for (int w = 1; w < NW; ++w) {         // ascending waves + strict '>' keeps the lowest position on ties
        const bool better = k[w].x > bkey;
        bkey = better ? k[w].x : bkey;
        bidx = better ? k[w].y : bidx;
        bx = better ? c[w].x : bx; by = better ? c[w].y : by; bz = better ? c[w].z : bz;
      }
      cur = bidx; cx = bx; cy = by; cz = bz;
      // the other parity buffer is only rewritten after the NEXT barrier, so no second barrier
    }
  }
}

// Round 6: clouds / sectors of 16 385 .. 24 576 positions (the reference trains S3DIS on clouds of up to 80 000 points: 4 sectors of
// ~20 000 at the first stage, 20 000-point clouds at the second).  Coordinates in registers (3 PPT of the 128 VGPRs a 1 024-thread
// workgroup would have per lane -- hence 512 threads, 48 / 64 points per lane, 256 VGPRs), the RUNNING DISTANCE in LDS ([slot][thread]:
// conflict-free, 96 / 128 KB), nothing global inside the loop.
// fps_global_kernel re-reads coordinates and distances through L1 / L2 for every pick -- 320 KB per pick at 20 000 rows, 6.7 us per pick,
// 75 % of an S3DIS-sized training step.  Same ownership, tie rules and padding as fps_reg_kernel (position p = tid * PPT + j; lowest
// lane, lowest slot wins): the picks are the same, bit for bit.  The cross-wave pick: one candidate per lane of a 16-lane row, four
// DPP steps on (distance bits, 15 - wave), then ONE broadcast read of the winner's point -- not 16 candidates in registers per lane.
template <int PPT, int NT, bool TIE>      // points per lane, threads (512: 8 waves = 2 per SIMD = 256 VGPRs each; 1 024 threads leave 128 and spill)
__global__ void __launch_bounds__(NT)
fps_lds_kernel(int n_arg, int m_arg, const float *__restrict__ xyz, const int *__restrict__ start,
               const int *__restrict__ offset, const int *__restrict__ new_offset,
               int *__restrict__ idx_out, int tie_bs, int tie_q, int tie_shift, const int *__restrict__ tie_n_dev, int guard) {
  extern __shared__ float fps_dl[];     // [PPT][NT] running distances
  constexpr int NW = NT / 64;
  __shared__ int2 red_key[2][16];
  const FpsSeg seg = fps_segment(blockIdx.x, n_arg, m_arg, start, offset, new_offset);
  const int n = seg.n, m = seg.m;
  if (m <= 0 || n <= 0) return;
  const float *pts = xyz + (size_t)seg.row0 * 3;
  int *out = idx_out + seg.out0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (TIE && tie_n_dev) {
    const int nr = max(*tie_n_dev, 1);
    const int bits = min(31 - __clz(nr), 10);
    tie_bs = 1 << bits;
    tie_q = (nr + tie_bs - 1) >> bits;
    tie_shift = 32 - bits;
  }
  if (guard) {      // paired launch (fps_dispatch): 1 = this kernel works only when the largest sector fits its positions
    if (tie_bs * tie_q > PPT * NT) return;
  }
  float px[PPT], py[PPT], pz[PPT];
  float *dl = fps_dl + tid;
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int p = TIE ? fps_point_of(tid * PPT + j, tie_bs, tie_q, tie_shift) : tid * PPT + j;
    const int pl = p < n ? p : 0;
    px[j] = pts[pl * 3 + 0]; py[j] = pts[pl * 3 + 1]; pz[j] = pts[pl * 3 + 2];
    dl[j * NT] = p < n ? 1e10f : -1.0f;
  }
  int cur = seg.start;
  if (cur < 0 || cur >= n) cur = 0;
  float cx = pts[cur * 3 + 0], cy = pts[cur * 3 + 1], cz = pts[cur * 3 + 2];
  for (int it = 0; it < m; ++it) {
    if (tid == 0) out[it] = cur + seg.idx_base;
    if (it == m - 1) break;
    int lmax = (int)0x80000000, slot = 0;
#pragma unroll
    for (int g = 0; g < PPT; g += 8) {      // eight slots at a time: their LDS reads in flight together, but not all PPT of them (registers)
      float o[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) o[u] = dl[(g + u) * NT];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = g + u;
        const float dx = px[j] - cx, dy = py[j] - cy, dz = pz[j] - cz;
        const float d = (dx * dx + dy * dy) + dz * dz;
        const int v = min(__float_as_int(d), __float_as_int(o[u]));            // (distances >= +0, padding -1.0f: integer order = float order)
        dl[j * NT] = __int_as_float(v);
        if (v > lmax) { lmax = v; slot = j; }                                  // strict: the lowest slot among equals
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const int wmax = __builtin_amdgcn_readfirstlane(rs_wave_max_i32(lmax));
    const unsigned long long cand = __ballot(lmax == wmax);
    const int wl = __ffsll((long long)cand) - 1;
    const int par = it & 1;
    // (the winner's coordinates are re-read from the cloud after the cross-wave pick, ~1 us of L2 latency per pick: selecting them out of
    //  3 PPT registers -- in a second pass or riding with the maximum -- makes the compiler spill 100-170 of the registers this kernel lives on)
    if (lane == wl)
      red_key[par][wave] = make_int2(wmax, TIE ? fps_point_of(tid * PPT + slot, tie_bs, tie_q, tie_shift) : tid * PPT + slot);
    __syncthreads();
    int hi = (lane & 15) < NW ? red_key[par][lane & 15].x : (int)0x80000000, lo = 15 - (lane & 15);      // greatest distance, then the lowest wave
    auto step = [&](int ohi, int olo) {
      const bool take = ohi > hi || (ohi == hi && olo > lo);
      hi = take ? ohi : hi; lo = take ? olo : lo;
    };
    step((int)rs_dpp<RS_DPP_QUAD_XOR1>((unsigned)hi), (int)rs_dpp<RS_DPP_QUAD_XOR1>((unsigned)lo));
    step((int)rs_dpp<RS_DPP_QUAD_XOR2>((unsigned)hi), (int)rs_dpp<RS_DPP_QUAD_XOR2>((unsigned)lo));
    step((int)rs_dpp<RS_DPP_ROW_HALF_MIRROR>((unsigned)hi), (int)rs_dpp<RS_DPP_ROW_HALF_MIRROR>((unsigned)lo));
    step((int)rs_dpp<RS_DPP_ROW_MIRROR>((unsigned)hi), (int)rs_dpp<RS_DPP_ROW_MIRROR>((unsigned)lo));
    const int ww = __builtin_amdgcn_readfirstlane(15 - lo);
    cur = __builtin_amdgcn_readfirstlane(red_key[par][ww].y);
    cx = pts[cur * 3 + 0]; cy = pts[cur * 3 + 1]; cz = pts[cur * 3 + 2];
    // (the other parity's slots are rewritten only after the NEXT barrier: no second barrier)
  }
}

// Fallback for clouds too large for registers: distances in `temp` (global), one pass per pick, 1024 threads.
// Classification: blocked ownership, lowest index on ties.  Packed batches: the reference kernel's own strided
// ownership (such clouds have > 8192 rows, so its block size is 1024 too) and its bit-reversed-thread tie rule.
__global__ void __launch_bounds__(1024)
fps_global_kernel(int n_arg, int m_arg, const float *__restrict__ xyz, const int *__restrict__ start,
                  const int *__restrict__ offset, const int *__restrict__ new_offset,
                  float *__restrict__ temp, int *__restrict__ idx_out, int tie, int tie_n, const int *__restrict__ tie_n_dev, int skip_positions) {
  __shared__ uint2 red_key[2][16];
  __shared__ unsigned red_tie[2][16];
  if (skip_positions && tie_n_dev) {      // paired with a guarded fps_reg_kernel: that one works when the largest sector fits its positions
    const int nr = max(*tie_n_dev, 1);
    const int bits = min(31 - __clz(nr), 10);
    if ((((nr + (1 << bits) - 1) >> bits) << bits) <= skip_positions) return;
  }
  const FpsSeg seg = fps_segment(blockIdx.x, n_arg, m_arg, start, offset, new_offset);
  const int n = seg.n, m = seg.m;
  if (m <= 0 || n <= 0) return;
  const float *pts = xyz + (size_t)seg.row0 * 3;
  float *dist = temp + seg.row0;
  int *out = idx_out + seg.out0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int per = (n + blockDim.x - 1) / blockDim.x;
  const int p0 = tie ? tid : tid * per, p1 = tie ? n : min(n, tid * per + per), pstep = tie ? (int)blockDim.x : 1;
  // the reference thread of row k is k mod bs, bs = opt_n_threads(largest cloud) <= 1024 = blockDim: thread tid here scans
  // rows of reference thread tid mod bs (several of ours share one when bs < 1024: lowest row among them wins)
  const int nref = tie_n_dev ? max(*tie_n_dev, 1) : max(tie_n, 1);
  const int tbits = min(31 - __clz(nref), 10);
  const unsigned trank = tie ? (tbits ? (__brev((unsigned)tid & ((1u << tbits) - 1u)) >> (32 - tbits)) : 0u) : (unsigned)tid;
  for (int p = p0; p < p1; p += pstep) dist[p] = 1e10f;
  int cur = seg.start;
  if (cur < 0 || cur >= n) cur = 0;
  for (int it = 0; it < m; ++it) {
    if (tid == 0) out[it] = cur + seg.idx_base;
    if (it == m - 1) break;
    const float cx = pts[cur * 3 + 0], cy = pts[cur * 3 + 1], cz = pts[cur * 3 + 2];
    unsigned lmax = 0u; int larg = 0x7fffffff;
    // eight rows per trip, all their loads requested before the first is used (round 6: one row per trip was a chain of dependent
    // L2 round trips -- 20 of them per pick at 20 000 rows, 6.7 us per pick, 75 % of an S3DIS-sized training step)
    for (int pb = p0; pb < p1; pb += 8 * pstep) {
      float x[8], y[8], z[8], o[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = min(pb + u * pstep, p1 - 1 >= p0 ? (p0 + ((p1 - 1 - p0) / pstep) * pstep) : p0);      // (clamped to this thread's last row)
        x[u] = pts[p * 3 + 0]; y[u] = pts[p * 3 + 1]; z[u] = pts[p * 3 + 2]; o[u] = dist[p];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = pb + u * pstep;
        if (p >= p1) break;
        const float dx = x[u] - cx, dy = y[u] - cy, dz = z[u] - cz;
        const float d = (dx * dx + dy * dy) + dz * dz;
        const float v = d < o[u] ? d : o[u];
        dist[p] = v;
        const unsigned vb = __float_as_uint(v);
        if (vb > lmax || larg == 0x7fffffff) { lmax = vb; larg = p; }
      }
    }
    if (p0 >= p1) { lmax = 0u; larg = 0x7fffffff; }
    const bool has = larg != 0x7fffffff;
    const unsigned wmax = (unsigned)__builtin_amdgcn_readfirstlane((int)rs_wave_max_u32(has ? lmax : 0u));
    // among the lanes holding the wave maximum: lowest index (classification) / lowest bit-reversed thread id (packed)
    const bool top = has && lmax == wmax;
    const unsigned key = top ? (tie ? trank : (unsigned)larg) : 0x7fffffffu;
    const unsigned wkey = rs_wave_min_u32(key);
    // (packed) several lanes may share the winning reference thread: the lowest row among them
    const unsigned warg = rs_wave_min_u32((top && key == wkey) ? (unsigned)larg : 0x7fffffffu);
    const int par = it & 1;
    if (lane == 0) { red_key[par][wave] = make_uint2(wmax, warg); red_tie[par][wave] = wkey; }
    __syncthreads();
    unsigned bkey = 0u, bidx = 0x7fffffffu, btie = 0x7fffffffu;
    for (int w = 0; w < nwaves; ++w) {
      const uint2 k = red_key[par][w];
      const unsigned t = red_tie[par][w];
      const bool better = (k.y != 0x7fffffffu) && (bidx == 0x7fffffffu || k.x > bkey || (k.x == bkey && t < btie) ||
                                                   (k.x == bkey && t == btie && k.y < bidx));
      bkey = better ? k.x : bkey;
      bidx = better ? k.y : bidx;
      btie = better ? t : btie;
    }
    cur = (int)bidx;
  }
}

int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

struct FpsTie { int bs, q, shift; const int *n_dev; int guard; };

template <int PPT, int NW>
void launch_reg2(int blocks, int n, int m, const float *xyz, const int *start, const int *offset,
                 const int *new_offset, int *idx, FpsTie tie, hipStream_t st) {
  if (tie.bs > 0)
    hipLaunchKernelGGL((fps_reg_kernel<PPT, NW, true>), dim3(blocks), dim3(64 * NW), 0, st, n, m, xyz, start, offset,
                       new_offset, idx, tie.bs, tie.q, tie.shift, tie.n_dev, tie.guard);
  else
    hipLaunchKernelGGL((fps_reg_kernel<PPT, NW, false>), dim3(blocks), dim3(64 * NW), 0, st, n, m, xyz, start, offset,
                       new_offset, idx, 0, 1, 0, (const int *)nullptr, 0);
}
template <int PPT>
void launch_reg(int blocks, int waves, int n, int m, const float *xyz, const int *start,
                const int *offset, const int *new_offset, int *idx, FpsTie tie, hipStream_t st) {
  switch (waves) {
    case 1: launch_reg2<PPT, 1>(blocks, n, m, xyz, start, offset, new_offset, idx, tie, st); break;
    case 2: launch_reg2<PPT, 2>(blocks, n, m, xyz, start, offset, new_offset, idx, tie, st); break;
    case 4: launch_reg2<PPT, 4>(blocks, n, m, xyz, start, offset, new_offset, idx, tie, st); break;
    case 8: launch_reg2<PPT, 8>(blocks, n, m, xyz, start, offset, new_offset, idx, tie, st); break;
    default: launch_reg2<PPT, 16>(blocks, n, m, xyz, start, offset, new_offset, idx, tie, st); break;
  }
}

// opt_n_threads of the reference (segmentation/modules/pointops/src/cuda_utils.h:10-13), same double arithmetic
int ref_block_size(int work_size) {
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int t = 1 << pow_2;
  if (t > 1024) t = 1024;
  return t < 1 ? 1 : t;
}

// n_max: the largest cloud a workgroup can meet; packed: the tie rule of the reference kernel launched for n_max rows
#ifdef RS_EXP_FAKE_FPS      // measurement builds only (tools/build_exp.sh): every (n / m)-th row instead of the sampling, to time a step without the sampling's launches
__global__ void fake_fps_kernel(int n, int m, const int *__restrict__ offset, const int *__restrict__ new_offset, int *__restrict__ idx) {
  const int c = blockIdx.x;
  const int s = offset ? (c ? offset[c - 1] : 0) : c * n, e = offset ? offset[c] : (c + 1) * n;
  const int ns = new_offset ? (c ? new_offset[c - 1] : 0) : c * m, ne = new_offset ? new_offset[c] : (c + 1) * m;
  const int cnt = ne - ns, rows = e - s;
  for (int j = threadIdx.x; j < cnt; j += blockDim.x) idx[ns + j] = (offset ? s : 0) + (int)((long long)j * rows / (cnt > 0 ? cnt : 1));
}
#endif

int fps_dispatch(int blocks, int n_max, int n, int m, const float *xyz, const int *start,
                 const int *offset, const int *new_offset, float *temp, int *idx, bool packed, const int *n_dev, hipStream_t st) {
#ifdef RS_EXP_FAKE_FPS
  hipLaunchKernelGGL(fake_fps_kernel, dim3(blocks), dim3(256), 0, st, n, m, offset, new_offset, idx);
  return RS_OK;
#endif
  FpsTie tie = {0, 1, 0, n_dev, 0};
  int positions = n_max;              // priority positions a workgroup must hold
  if (packed) {
    tie.bs = ref_block_size(n_max);
    tie.q = (n_max + tie.bs - 1) / tie.bs;
    int bits = 0;
    while ((1 << bits) < tie.bs) ++bits;
    tie.shift = 32 - bits;
    positions = tie.bs * tie.q;
    // largest cloud known only as a bound (n_dev holds the real one): bs * q < n + bs <= 2 n, and < n + 1024 once n >= 1024
    if (n_dev) positions = n_max >= 1024 ? ((n_max + 1023) / 1024 + 1) * 1024 : 2 * n_max;
  }
  // Waves per workgroup: the loop is a latency chain, so fewer waves (no cross-wave hop, or a short one)
  // win as long as the per-lane point count stays small (profiles/: RS_FPS_WAVES sweeps).
  int waves = env_int("RS_FPS_WAVES", 0);
  if (waves <= 0) waves = positions <= 512 ? 1 : (positions <= 1024 ? 2 : (positions <= 4096 ? 4 : (positions <= 8192 ? 8 : 16)));
  int w = 1;
  while (w < waves && w < 16) w <<= 1;       // power of two
  waves = w;
  int ppt = (positions + waves * 64 - 1) / (waves * 64);
  while (ppt > 16 && waves < 16) { waves *= 2; ppt = (positions + waves * 64 - 1) / (waves * 64); }
  // Round 6: 24 points per lane at 16 waves (coordinates + running distance = 96 of the 128 VGPRs a 1 024-thread workgroup has per lane):
  // 24 576 positions in registers -- the 20 000-point clouds / sectors of the reference's S3DIS batches (80 000 points, 4 sectors at
  // the first stage, stride 4) -- instead of 16 384; above that the distances live in `temp` (fps_global_kernel: 6.7 us per pick at
  // 20 000 points against ~2 here).  RS_FPS_REG24=0: the round-5 limit.
  static const int reg24 = env_int("RS_FPS_REG24", 0);      // (off: at 1 024 threads the 24-point instances need ~180 VGPRs and spill 140 of them)
  const int ppt_max = reg24 ? 24 : 16;
  // 16 385 .. 24 576 positions: coordinates in registers, distances in LDS (fps_lds_kernel; RS_FPS_LDS=0: fps_global_kernel as in round 5)
  static const int lds_on = env_int("RS_FPS_LDS", 1);
  if (lds_on && ppt > ppt_max && waves == 16) {
    const bool bound_only = packed && n_dev;      // sectorized: `positions` is a bound from the whole cloud, the largest sector lives on the device
    if (positions <= 48 * 512 || bound_only) {
      // 48 points per lane x 512 threads = 24 576 positions (64 per lane spill ~100 registers: built, measured, not kept)
      FpsTie g = tie;
      g.guard = (bound_only && positions > 48 * 512) ? 1 : 0;
      const size_t lds = (size_t)48 * 512 * sizeof(float);
      if (tie.bs > 0) hipLaunchKernelGGL((fps_lds_kernel<48, 512, true>), dim3(blocks), dim3(512), lds, st, n, m, xyz, start, offset, new_offset, idx, g.bs, g.q, g.shift, g.n_dev, g.guard);
      else hipLaunchKernelGGL((fps_lds_kernel<48, 512, false>), dim3(blocks), dim3(512), lds, st, n, m, xyz, start, offset, new_offset, idx, 0, 1, 0, (const int *)nullptr, 0);
      if (!g.guard) return RS_OK;
      // paired: the largest sector (device) decides -- this kernel when it fits 24 576 positions, fps_global_kernel otherwise
      if (!temp) { rs_set_error("rs_furthestsampling: n=%d needs the `temp` scratch (one float per row)", n_max); return RS_ERR_ARG; }
      hipLaunchKernelGGL(fps_global_kernel, dim3(blocks), dim3(1024), 0, st, n, m, xyz, start, offset, new_offset, temp, idx, 1, n_max, n_dev, 48 * 512);
      return RS_OK;
    }
  }
  if (ppt > ppt_max) {   // the distances no longer fit the register file
    if (!temp) { rs_set_error("rs_furthestsampling: n=%d needs the `temp` scratch (one float per row)", n_max); return RS_ERR_ARG; }
    int skip = 0;
    if (reg24 && packed && n_dev) {
      // sectorized FPS: the host bounds a sector by its whole cloud; the largest sector (on the device) usually fits the registers.
      // Both kernels are launched, the device value decides which one works (the other's workgroups return at once).
      FpsTie g = tie;
      g.guard = 1;
      launch_reg<24>(blocks, 16, n, m, xyz, start, offset, new_offset, idx, g, st);
      skip = 24 * 1024;
    }
    hipLaunchKernelGGL(fps_global_kernel, dim3(blocks), dim3(1024), 0, st, n, m, xyz, start, offset, new_offset, temp, idx, packed ? 1 : 0, n_max, n_dev, skip);
    return RS_OK;
  }
  if (ppt <= 1) launch_reg<1>(blocks, waves, n, m, xyz, start, offset, new_offset, idx, tie, st);
  else if (ppt <= 2) launch_reg<2>(blocks, waves, n, m, xyz, start, offset, new_offset, idx, tie, st);
  else if (ppt <= 4) launch_reg<4>(blocks, waves, n, m, xyz, start, offset, new_offset, idx, tie, st);
  else if (ppt <= 8) launch_reg<8>(blocks, waves, n, m, xyz, start, offset, new_offset, idx, tie, st);
  else if (ppt <= 16) launch_reg<16>(blocks, waves, n, m, xyz, start, offset, new_offset, idx, tie, st);
  else launch_reg<24>(blocks, waves, n, m, xyz, start, offset, new_offset, idx, tie, st);
  return RS_OK;
}

}  // namespace

extern "C" int rs_furthestsampling(int b, int n, int m, const float *xyz, const int *start,
                                   float *temp, int *idx, void *stream) {
  RS_REQUIRE(b >= 0 && n >= 0 && m >= 0, "rs_furthestsampling: negative size (b=%d n=%d m=%d)", b, n, m);
  if (b == 0 || m == 0) return RS_OK;
  RS_REQUIRE(n > 0, "rs_furthestsampling: empty cloud with m=%d picks", m);
  RS_REQUIRE(xyz && idx, "rs_furthestsampling: null pointer");
  int rc = fps_dispatch(b, n, n, m, xyz, start, nullptr, nullptr, temp, idx, false, nullptr, (hipStream_t)stream);
  if (rc != RS_OK) return rc;
  RS_CHECK_LAUNCH("rs_furthestsampling");
  return RS_OK;
}

extern "C" int rs_furthestsampling_offset(int b, int n_max, const float *xyz, const int *offset,
                                          const int *new_offset, float *temp, int *idx, void *stream) {
  RS_REQUIRE(b >= 0 && n_max >= 0, "rs_furthestsampling_offset: negative size");
  if (b == 0 || n_max == 0) return RS_OK;
  RS_REQUIRE(xyz && offset && new_offset && idx, "rs_furthestsampling_offset: null pointer");
  int rc = fps_dispatch(b, n_max, 0, 0, xyz, nullptr, offset, new_offset, temp, idx, true, nullptr, (hipStream_t)stream);
  if (rc != RS_OK) return rc;
  RS_CHECK_LAUNCH("rs_furthestsampling_offset");
  return RS_OK;
}


/* Packed FPS over the sectors rs_sectorize produced: like rs_furthestsampling_offset, but the largest cloud is known to
 * the host only as the bound n_bound; the true value (which fixes the reference kernel's block size and therefore its
 * tie rule) is read from n_max_dev on the device. */
extern "C" int rs_furthestsampling_sectors(int b, int n_bound, const int *n_max_dev, const float *xyz, const int *offset,
                                           const int *new_offset, float *temp, int *idx, void *stream) {
  RS_REQUIRE(b >= 0 && n_bound >= 0, "rs_furthestsampling_sectors: negative size");
  if (b == 0 || n_bound == 0) return RS_OK;
  RS_REQUIRE(xyz && offset && new_offset && idx && n_max_dev, "rs_furthestsampling_sectors: null pointer");
  int rc = fps_dispatch(b, n_bound, 0, 0, xyz, nullptr, offset, new_offset, temp, idx, true, n_max_dev, (hipStream_t)stream);
  if (rc != RS_OK) return rc;
  RS_CHECK_LAUNCH("rs_furthestsampling_sectors");
  return RS_OK;
}
