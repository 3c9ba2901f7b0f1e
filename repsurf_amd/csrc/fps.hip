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
               int *__restrict__ idx_out, int tie_bs, int tie_q, int tie_shift, const int *__restrict__ tie_n_dev) {
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
  }

  float px[PPT], py[PPT], pz[PPT], md[PPT];
  int kid[PPT];                       // TIE: the point each slot holds (the blocked layout needs no table)
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int p = TIE ? fps_point_of(tid * PPT + j, tie_bs, tie_q, tie_shift) : tid * PPT + j;
    kid[j] = p;
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
      sk = TIE ? kid[0] : tid * PPT + slot; sx = px[0]; sy = py[0]; sz = pz[0];
#pragma unroll
      for (int j = 1; j < PPT; ++j) {
        const bool hit = slot == j;
        if (TIE) sk = hit ? kid[j] : sk;
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

// Fallback for clouds too large for registers: distances in `temp` (global), one pass per pick, 1024 threads.
// Classification: blocked ownership, lowest index on ties.  Packed batches: the reference kernel's own strided
// ownership (such clouds have > 8192 rows, so its block size is 1024 too) and its bit-reversed-thread tie rule.
__global__ void __launch_bounds__(1024)
fps_global_kernel(int n_arg, int m_arg, const float *__restrict__ xyz, const int *__restrict__ start,
                  const int *__restrict__ offset, const int *__restrict__ new_offset,
                  float *__restrict__ temp, int *__restrict__ idx_out, int tie, int tie_n, const int *__restrict__ tie_n_dev) {
  __shared__ uint2 red_key[2][16];
  __shared__ unsigned red_tie[2][16];
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
    for (int p = p0; p < p1; p += pstep) {
      const float dx = pts[p * 3 + 0] - cx, dy = pts[p * 3 + 1] - cy, dz = pts[p * 3 + 2] - cz;
      const float d = (dx * dx + dy * dy) + dz * dz;
      const float o = dist[p];
      const float v = d < o ? d : o;
      dist[p] = v;
      const unsigned vb = __float_as_uint(v);
      if (vb > lmax || larg == 0x7fffffff) { lmax = vb; larg = p; }
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

struct FpsTie { int bs, q, shift; const int *n_dev; };

template <int PPT, int NW>
void launch_reg2(int blocks, int n, int m, const float *xyz, const int *start, const int *offset,
                 const int *new_offset, int *idx, FpsTie tie, hipStream_t st) {
  if (tie.bs > 0)
    hipLaunchKernelGGL((fps_reg_kernel<PPT, NW, true>), dim3(blocks), dim3(64 * NW), 0, st, n, m, xyz, start, offset,
                       new_offset, idx, tie.bs, tie.q, tie.shift, tie.n_dev);
  else
    hipLaunchKernelGGL((fps_reg_kernel<PPT, NW, false>), dim3(blocks), dim3(64 * NW), 0, st, n, m, xyz, start, offset,
                       new_offset, idx, 0, 1, 0, (const int *)nullptr);
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
  FpsTie tie = {0, 1, 0, n_dev};
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
  if (ppt > 16) {   // > 16384 positions per cloud: distances no longer fit the register file
    if (!temp) { rs_set_error("rs_furthestsampling: n=%d needs the `temp` scratch (one float per row)", n_max); return RS_ERR_ARG; }
    hipLaunchKernelGGL(fps_global_kernel, dim3(blocks), dim3(1024), 0, st, n, m, xyz, start, offset, new_offset, temp, idx, packed ? 1 : 0, n_max, n_dev);
    return RS_OK;
  }
  if (ppt <= 1) launch_reg<1>(blocks, waves, n, m, xyz, start, offset, new_offset, idx, tie, st);
  else if (ppt <= 2) launch_reg<2>(blocks, waves, n, m, xyz, start, offset, new_offset, idx, tie, st);
  else if (ppt <= 4) launch_reg<4>(blocks, waves, n, m, xyz, start, offset, new_offset, idx, tie, st);
  else if (ppt <= 8) launch_reg<8>(blocks, waves, n, m, xyz, start, offset, new_offset, idx, tie, st);
  else launch_reg<16>(blocks, waves, n, m, xyz, start, offset, new_offset, idx, tie, st);
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
