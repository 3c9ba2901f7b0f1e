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

template <int PPT, int NW>     // points per thread, waves per workgroup (blockDim.x == 64 * NW)
__global__ void __launch_bounds__(64 * NW)
fps_reg_kernel(int n_arg, int m_arg, const float *__restrict__ xyz, const int *__restrict__ start,
               const int *__restrict__ offset, const int *__restrict__ new_offset,
               int *__restrict__ idx_out) {
  __shared__ uint2 red_key[2][16];    // (max distance bits, point index) per wave, double buffered
  __shared__ float4 red_xyz[2][16];

  const FpsSeg seg = fps_segment(blockIdx.x, n_arg, m_arg, start, offset, new_offset);
  const int n = seg.n, m = seg.m;
  if (m <= 0 || n <= 0) return;
  const float *pts = xyz + (size_t)seg.row0 * 3;
  int *out = idx_out + seg.out0;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  float px[PPT], py[PPT], pz[PPT], md[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int p = tid * PPT + j;
    if (p < n) {
      px[j] = pts[p * 3 + 0];
      py[j] = pts[p * 3 + 1];
      pz[j] = pts[p * 3 + 2];
      md[j] = 1e10f;   // pointnet2_utils.py:65
    } else {           // padding: distance pinned at 0 and highest indices -> never beats a real point
      px[j] = py[j] = pz[j] = 0.f;
      md[j] = 0.f;
    }
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
    unsigned lmax = 0u;
    if constexpr (PPT % 2 == 0) {
      typedef float v2f __attribute__((ext_vector_type(2)));
      const v2f cx2 = {cx, cx}, cy2 = {cy, cy}, cz2 = {cz, cz};
#pragma unroll
      for (int j = 0; j < PPT; j += 2) {
        const v2f x2 = {px[j], px[j + 1]}, y2 = {py[j], py[j + 1]}, z2 = {pz[j], pz[j + 1]};
        const v2f dx = x2 - cx2, dy = y2 - cy2, dz = z2 - cz2;
        const v2f d = (dx * dx + dy * dy) + dz * dz;
        md[j] = fminf(d.x, md[j]);
        md[j + 1] = fminf(d.y, md[j + 1]);
        lmax = max(lmax, max(__float_as_uint(md[j]), __float_as_uint(md[j + 1])));
      }
    } else {
#pragma unroll
      for (int j = 0; j < PPT; ++j) {
        const float dx = px[j] - cx, dy = py[j] - cy, dz = pz[j] - cz;
        const float d = (dx * dx + dy * dy) + dz * dz;
        md[j] = d < md[j] ? d : md[j];
        lmax = max(lmax, __float_as_uint(md[j]));
      }
    }
    const unsigned wmax = (unsigned)__builtin_amdgcn_readfirstlane((int)rs_wave_max_u32(lmax));

    // lowest slot in this lane that holds the wave max, and its coordinates
    int slot = PPT;
    float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
    for (int j = PPT - 1; j >= 0; --j) {
      const bool hit = __float_as_uint(md[j]) == wmax;
      slot = hit ? j : slot;
      sx = hit ? px[j] : sx;
      sy = hit ? py[j] : sy;
      sz = hit ? pz[j] : sz;
    }
    const unsigned long long cand = __ballot(slot < PPT);
    const int wl = __ffsll((long long)cand) - 1;   // lowest lane holding the max (cand != 0 always)
    const int wslot = __builtin_amdgcn_readlane(slot, wl);
    const float wx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sx), wl));
    const float wy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sy), wl));
    const float wz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sz), wl));
    const int widx = (wave * 64 + wl) * PPT + wslot;

    if (NW == 1) {
      cur = widx; cx = wx; cy = wy; cz = wz;
    } else {
      const int par = it & 1;
      if (lane == 0) {
        red_key[par][wave] = make_uint2(wmax, (unsigned)widx);
        red_xyz[par][wave] = make_float4(wx, wy, wz, 0.f);
      }
      __syncthreads();
      // all NW candidates are fetched with independent LDS reads (one wait), then compared in registers
      uint2 k[NW]; float4 c[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) { k[w] = red_key[par][w]; c[w] = red_xyz[par][w]; }
      unsigned bkey = k[0].x; int bidx = (int)k[0].y; float bx = c[0].x, by = c[0].y, bz = c[0].z;
#pragma unroll
      for (int w = 1; w < NW; ++w) {         // ascending waves + strict '>' keeps the lowest index on ties
        const bool better = k[w].x > bkey;
        bkey = better ? k[w].x : bkey;
        bidx = better ? (int)k[w].y : bidx;
        bx = better ? c[w].x : bx; by = better ? c[w].y : by; bz = better ? c[w].z : bz;
      }
      cur = bidx; cx = bx; cy = by; cz = bz;
      // the other parity buffer is only rewritten after the NEXT barrier, so no second barrier
    }
  }
}

// Fallback for clouds too large for registers: distances in `temp` (global), one pass per pick.
__global__ void __launch_bounds__(1024)
fps_global_kernel(int n_arg, int m_arg, const float *__restrict__ xyz, const int *__restrict__ start,
                  const int *__restrict__ offset, const int *__restrict__ new_offset,
                  float *__restrict__ temp, int *__restrict__ idx_out) {
  __shared__ uint2 red_key[2][16];
  const FpsSeg seg = fps_segment(blockIdx.x, n_arg, m_arg, start, offset, new_offset);
  const int n = seg.n, m = seg.m;
  if (m <= 0 || n <= 0) return;
  const float *pts = xyz + (size_t)seg.row0 * 3;
  float *dist = temp + seg.row0;
  int *out = idx_out + seg.out0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  // blocked ownership so that "lowest thread, lowest local position" == lowest index
  const int per = (n + blockDim.x - 1) / blockDim.x;
  const int p0 = tid * per, p1 = min(n, p0 + per);
  for (int p = p0; p < p1; ++p) dist[p] = 1e10f;
  int cur = seg.start;
  if (cur < 0 || cur >= n) cur = 0;
  for (int it = 0; it < m; ++it) {
    if (tid == 0) out[it] = cur + seg.idx_base;
    if (it == m - 1) break;
    const float cx = pts[cur * 3 + 0], cy = pts[cur * 3 + 1], cz = pts[cur * 3 + 2];
    unsigned lmax = 0u; int larg = 0x7fffffff;
    for (int p = p0; p < p1; ++p) {
      const float dx = pts[p * 3 + 0] - cx, dy = pts[p * 3 + 1] - cy, dz = pts[p * 3 + 2] - cz;
      const float d = (dx * dx + dy * dy) + dz * dz;
      const float o = dist[p];
      const float v = d < o ? d : o;
      dist[p] = v;
      const unsigned vb = __float_as_uint(v);
      if (vb > lmax || larg == 0x7fffffff) { lmax = vb; larg = p; }
    }
    if (p0 >= p1) { lmax = 0u; larg = 0x7fffffff; }
    const unsigned wmax = (unsigned)__builtin_amdgcn_readfirstlane((int)rs_wave_max_u32(lmax));
    const unsigned warg = rs_wave_min_u32(lmax == wmax ? (unsigned)larg : 0x7fffffffu);
    const int par = it & 1;
    if (lane == 0) red_key[par][wave] = make_uint2(wmax, warg);
    __syncthreads();
    unsigned bkey = 0u; unsigned bidx = 0x7fffffffu;
    for (int w = 0; w < nwaves; ++w) {
      const uint2 k = red_key[par][w];
      const bool better = (k.y != 0x7fffffffu) && (bidx == 0x7fffffffu || k.x > bkey);
      bkey = better ? k.x : bkey;
      bidx = better ? k.y : bidx;
    }
    cur = (int)bidx;
  }
}

int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

template <int PPT, int NW>
void launch_reg2(int blocks, int n, int m, const float *xyz, const int *start, const int *offset,
                 const int *new_offset, int *idx, hipStream_t st) {
  hipLaunchKernelGGL((fps_reg_kernel<PPT, NW>), dim3(blocks), dim3(64 * NW), 0, st, n, m, xyz, start, offset,
                     new_offset, idx);
}
template <int PPT>
void launch_reg(int blocks, int waves, int n, int m, const float *xyz, const int *start,
                const int *offset, const int *new_offset, int *idx, hipStream_t st) {
  switch (waves) {
    case 1: launch_reg2<PPT, 1>(blocks, n, m, xyz, start, offset, new_offset, idx, st); break;
    case 2: launch_reg2<PPT, 2>(blocks, n, m, xyz, start, offset, new_offset, idx, st); break;
    case 4: launch_reg2<PPT, 4>(blocks, n, m, xyz, start, offset, new_offset, idx, st); break;
    case 8: launch_reg2<PPT, 8>(blocks, n, m, xyz, start, offset, new_offset, idx, st); break;
    default: launch_reg2<PPT, 16>(blocks, n, m, xyz, start, offset, new_offset, idx, st); break;
  }
}

// n_max: the largest cloud a workgroup can meet
int fps_dispatch(int blocks, int n_max, int n, int m, const float *xyz, const int *start,
                 const int *offset, const int *new_offset, float *temp, int *idx, hipStream_t st) {
  // Waves per workgroup: the loop is a latency chain, so fewer waves (no cross-wave hop, or a short one)
  // win as long as the per-lane point count stays small: measured best 1 wave up to 512 points, 2 waves
  // at 1024 (profiles/: RS_FPS_WAVES sweep).
  int waves = env_int("RS_FPS_WAVES", 0);
  if (waves <= 0) waves = n_max <= 512 ? 1 : (n_max <= 1024 ? 2 : (n_max <= 4096 ? 4 : (n_max <= 8192 ? 8 : 16)));
  int w = 1;
  while (w < waves && w < 16) w <<= 1;       // power of two
  waves = w;
  int ppt = (n_max + waves * 64 - 1) / (waves * 64);
  while (ppt > 16 && waves < 16) { waves *= 2; ppt = (n_max + waves * 64 - 1) / (waves * 64); }
  if (ppt > 16) {   // > 16384 points per cloud: distances no longer fit the register file
    if (!temp) { rs_set_error("rs_furthestsampling: n=%d needs the `temp` scratch (b*n floats)", n_max); return RS_ERR_ARG; }
    hipLaunchKernelGGL(fps_global_kernel, dim3(blocks), dim3(1024), 0, st, n, m, xyz, start, offset, new_offset, temp, idx);
    return RS_OK;
  }
  if (ppt <= 1) launch_reg<1>(blocks, waves, n, m, xyz, start, offset, new_offset, idx, st);
  else if (ppt <= 2) launch_reg<2>(blocks, waves, n, m, xyz, start, offset, new_offset, idx, st);
  else if (ppt <= 4) launch_reg<4>(blocks, waves, n, m, xyz, start, offset, new_offset, idx, st);
  else if (ppt <= 8) launch_reg<8>(blocks, waves, n, m, xyz, start, offset, new_offset, idx, st);
  else launch_reg<16>(blocks, waves, n, m, xyz, start, offset, new_offset, idx, st);
  return RS_OK;
}

}  // namespace

extern "C" int rs_furthestsampling(int b, int n, int m, const float *xyz, const int *start,
                                   float *temp, int *idx, void *stream) {
  RS_REQUIRE(b >= 0 && n >= 0 && m >= 0, "rs_furthestsampling: negative size (b=%d n=%d m=%d)", b, n, m);
  if (b == 0 || m == 0) return RS_OK;
  RS_REQUIRE(n > 0, "rs_furthestsampling: empty cloud with m=%d picks", m);
  RS_REQUIRE(xyz && idx, "rs_furthestsampling: null pointer");
  int rc = fps_dispatch(b, n, n, m, xyz, start, nullptr, nullptr, temp, idx, (hipStream_t)stream);
  if (rc != RS_OK) return rc;
  RS_CHECK_LAUNCH("rs_furthestsampling");
  return RS_OK;
}

extern "C" int rs_furthestsampling_offset(int b, int n_max, const float *xyz, const int *offset,
                                          const int *new_offset, float *temp, int *idx, void *stream) {
  RS_REQUIRE(b >= 0 && n_max >= 0, "rs_furthestsampling_offset: negative size");
  if (b == 0 || n_max == 0) return RS_OK;
  RS_REQUIRE(xyz && offset && new_offset && idx, "rs_furthestsampling_offset: null pointer");
  int rc = fps_dispatch(b, n_max, 0, 0, xyz, nullptr, offset, new_offset, temp, idx, (hipStream_t)stream);
  if (rc != RS_OK) return rc;
  RS_CHECK_LAUNCH("rs_furthestsampling_offset");
  return RS_OK;
}
