// umbrella_fan.h — the per-point triangle fan of the umbrella-surface constructor, shared by the
// classification kernel (dense clouds, nearest neighbour dropped) and the segmentation kernel
// (packed clouds, the point itself stays in the ring, azimuth taken after a fixed rotation).
//
// Input: G ring offsets (neighbour - query) in kNN order.  Steps, one thread per point, all in
// registers (classification/modules/repsurface_utils.py:121-131, recons_utils.py:27-57,82-90,
// 108-124,152-176, polar_utils.py:10-31; the segmentation copies of those files are identical up to
// the pieces selected by the template flags):
//   1. key_j = atan2(y, x)/(2*pi) + 0.5 of the (optionally rotated) offset; stable sort by key.
//      Keys closer than RS_PHI_TIE are ordered by the exact sign of the 2-D cross product of the
//      coordinates the key was computed from (fp64: products of floats are exact).
//   2. triangles (origin, s_j, s_{j+1 mod G}): normal = unit cross product, centroid, polar form of the
//      centroid, constant <n, c>/sqrt(3); sign rule (first triangle's x-component positive), per-cloud flip.
//   3. degenerate (NaN-normal) triangles take normal / centroid / constant of the first valid one.
// Arithmetic order follows the PyTorch CPU kernels operation by operation (tests/golden/probe.json).
#pragma once
#include "rs_common.h"
#include <math.h>

#define RS_PHI_TIE 4.8e-7f          // 8 ulp at 1.0 in normalised-azimuth units
#define RS_PI_F 3.14159274101257324f      // float(np.pi)
#define RS_TWO_PI_F 6.28318548202514648f  // float(2*np.pi)
#define RS_SQRT3_F 1.73205077648162842f   // torch.sqrt(torch.Tensor([3]))

// "b goes before a" for two fan neighbours a (earlier position) and b (later position).
__device__ __forceinline__ bool rs_phi_before(float ka, float xa, float ya, float kb, float xb, float yb) {
  const float diff = kb - ka;
  if (__builtin_expect(fabsf(diff) <= RS_PHI_TIE, 0)) {
    const double cr = (double)xa * (double)yb - (double)xb * (double)ya;   // > 0: b is counter-clockwise of a
    return cr < 0.0;
  }
  return diff < 0.f;
}

// ROT: sort key from the offset rotated by segmentation/modules/repsurface_utils.py:71-74
//      (xyz @ [[0.5,-0.5,0.7071],[0.7071,0.7071,0],[-0.5,0.5,0.7071]]; torch.matmul with K=3 is the
//      sequential FMA chain fma(z,R2j, fma(y,R1j, x*R0j)) on the CPU build, probed).
// SEG_ORDER: channel order [polar(3), normal(3), const(1), centroid(3)] (segmentation :320) instead of
//      [centroid(3), polar(3), normal(3), const(1)] (classification :290).
template <int G, bool ROT, bool SEG_ORDER>
__device__ __forceinline__ void rs_fan_features(float (&ox)[G], float (&oy)[G], float (&oz)[G], float flip,
                                                float *__restrict__ orow) {
  float key[G], kx[G], ky[G];
#pragma unroll
  for (int j = 0; j < G; ++j) {
    if (ROT) {
      kx[j] = rs_fma(oz[j], -0.5f, rs_fma(oy[j], 0.7071f, ox[j] * 0.5f));
      ky[j] = rs_fma(oz[j], 0.5f, rs_fma(oy[j], 0.7071f, ox[j] * -0.5f));
    } else {
      kx[j] = ox[j]; ky[j] = oy[j];
    }
    key[j] = atan2f(ky[j], kx[j]) / RS_TWO_PI_F + 0.5f;    // xyz2sphere(...)[..., 2]
  }
  // stable odd-even transposition sort by azimuth (argsort)
#pragma unroll
  for (int round = 0; round < G; ++round) {
#pragma unroll
    for (int j = (round & 1); j + 1 < G; j += 2) {
      const bool sw = ROT ? rs_phi_before(key[j], kx[j], ky[j], key[j + 1], kx[j + 1], ky[j + 1])
                          : rs_phi_before(key[j], ox[j], oy[j], key[j + 1], ox[j + 1], oy[j + 1]);
      const float tk = key[j], tx = ox[j], ty = oy[j], tz = oz[j];
      key[j] = sw ? key[j + 1] : tk; ox[j] = sw ? ox[j + 1] : tx; oy[j] = sw ? oy[j + 1] : ty; oz[j] = sw ? oz[j + 1] : tz;
      key[j + 1] = sw ? tk : key[j + 1]; ox[j + 1] = sw ? tx : ox[j + 1]; oy[j + 1] = sw ? ty : oy[j + 1]; oz[j + 1] = sw ? tz : oz[j + 1];
      if (ROT) {
        const float ux_ = kx[j], uy_ = ky[j];
        kx[j] = sw ? kx[j + 1] : ux_; ky[j] = sw ? ky[j + 1] : uy_;
        kx[j + 1] = sw ? ux_ : kx[j + 1]; ky[j + 1] = sw ? uy_ : ky[j + 1];
      }
    }
  }

  // triangle fan (origin, s_j, s_{j+1}): normal / centroid / polar / constant
  float ux[G], uy[G], uz[G], cx[G], cy[G], cz[G], rho[G], th[G], ph[G], pos[G];
  bool bad[G];
#pragma unroll
  for (int j = 0; j < G; ++j) {
    const int j2 = (j + 1 == G) ? 0 : j + 1;
    const float ax = ox[j], ay = oy[j], az = oz[j], bx = ox[j2], by = oy[j2], bz = oz[j2];
    const float nx = rs_fma(ay, bz, -(az * by));     // torch.cross (contracted on the CPU build)
    const float ny = rs_fma(az, bx, -(ax * bz));
    const float nz = rs_fma(ax, by, -(ay * bx));
    const float len = sqrtf(rs_fma(nz, nz, rs_fma(ny, ny, nx * nx)));   // torch.norm
    ux[j] = nx / len; uy[j] = ny / len; uz[j] = nz / len;
    cx[j] = ((0.f + ax) + bx) / 3.f; cy[j] = ((0.f + ay) + by) / 3.f; cz[j] = ((0.f + az) + bz) / 3.f;
  }
  // keep x_n of the FIRST triangle positive (NaN -> -1), then the per-cloud random flip
  const float pm = (ux[0] > 0.f) ? 1.f : -1.f;
#pragma unroll
  for (int j = 0; j < G; ++j) {
    ux[j] = (ux[j] * pm) * flip; uy[j] = (uy[j] * pm) * flip; uz[j] = (uz[j] * pm) * flip;
    const float r = sqrtf(rs_sqnorm(cx[j], cy[j], cz[j]));
    rho[j] = r;
    th[j] = (r == 0.f) ? 0.f : acosf(cz[j] / r) / RS_PI_F;
    ph[j] = atan2f(cy[j], cx[j]) / RS_TWO_PI_F + 0.5f;
    pos[j] = ((ux[j] * cx[j] + uy[j] * cy[j]) + uz[j] * cz[j]) / RS_SQRT3_F;
    bad[j] = (ux[j] != ux[j]) || (uy[j] != uy[j]) || (uz[j] != uz[j]);
  }
  // check_nan_umb: first valid triangle (0 when none) donates normal / centroid / constant
  float fux = ux[0], fuy = uy[0], fuz = uz[0], fcx = cx[0], fcy = cy[0], fcz = cz[0], fpos = pos[0];
  bool found = !bad[0];
#pragma unroll
  for (int j = 1; j < G; ++j) {
    const bool take = !found && !bad[j];
    fux = take ? ux[j] : fux; fuy = take ? uy[j] : fuy; fuz = take ? uz[j] : fuz;
    fcx = take ? cx[j] : fcx; fcy = take ? cy[j] : fcy; fcz = take ? cz[j] : fcz;
    fpos = take ? pos[j] : fpos;
    found = found || !bad[j];
  }
#pragma unroll
  for (int j = 0; j < G; ++j) {
    const bool r = bad[j];
    float *o = orow + j * 10;
    const float ocx = r ? fcx : cx[j], ocy = r ? fcy : cy[j], ocz = r ? fcz : cz[j];
    const float onx = r ? fux : ux[j], ony = r ? fuy : uy[j], onz = r ? fuz : uz[j];
    const float op = r ? fpos : pos[j];
    if (SEG_ORDER) {
      o[0] = rho[j]; o[1] = th[j]; o[2] = ph[j];
      o[3] = onx; o[4] = ony; o[5] = onz;
      o[6] = op;
      o[7] = ocx; o[8] = ocy; o[9] = ocz;
    } else {
      o[0] = ocx; o[1] = ocy; o[2] = ocz;
      o[3] = rho[j]; o[4] = th[j]; o[5] = ph[j];
      o[6] = onx; o[7] = ony; o[8] = onz;
      o[9] = op;
    }
  }
}
