/*
 * geom_oracle.c — CPU restatement (plain C, scalar loops) of the geometry half of the RepSurf-U
 * hot path.  TEST INFRASTRUCTURE ONLY: it is the checker for the HIP kernels in
 * repsurf_amd/csrc; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  Nothing under repsurf_amd/ imports or links it.
 *
 * Each function restates, operation by operation, what the reference's CPU/PyTorch path
 * (`cuda=False`) computes; citations are relative to /root/reference.  The floating-point
 * operation ORDER was pinned by probing the PyTorch 2.10 CPU kernels the reference runs on
 * (tests/golden/make_golden.py records the probe) and by bit-exact comparison with the
 * reference's own outputs (tests/golden/*.npz, tests/test_oracle_golden.py):
 *   torch.sum(p**2, -1)            -> (x*x + y*y) + z*z
 *   torch.matmul (K = 3)           -> fma(a2,b2, fma(a1,b1, a0*b0))
 *   torch.cross                    -> fma(a1,b2, -(a2*b1)), ...
 *   torch.norm(dim=-1) of 3        -> sqrt(fma(z,z, fma(y,y, x*x)))
 *   torch.mean of 3 rows           -> ((r0 + r1) + r2) / 3
 *   t / np.pi, t / (2*np.pi) + .5  -> t / (float)pi, t / (float)(2 pi) + 0.5f
 * Compile with -ffp-contract=off (see Makefile) so that only the fmaf() calls fuse.
 *
 * Parity status: PINNED for both sub-projects.  Classification: against the reference's executable CPU path for
 * FPS / ball query / kNN indices (bit-exact) and umbrella features (1e-6).  Segmentation (packed batches): the functions
 * restate CUDA kernels, and those kernels are EXECUTED here -- oracle/_ref compiles the reference's *_cuda_kernel.cu files
 * unmodified as host code (oracle/Makefile.ref) and tests/test_oracle_ref.py / test_oracle_seg_golden.py compare every
 * packed-batch function below with them, tie rules included.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PI_F 3.14159274101257324f      /* (float)np.pi     */
#define TWO_PI_F 6.28318548202514648f  /* (float)(2*np.pi) */
#define SQRT3_F 1.73205077648162842f   /* torch.sqrt(torch.Tensor([3])) */
#define PHI_TIE 4.8e-7f

static float sqnorm3(float x, float y, float z) { return (x * x + y * y) + z * z; }

/* square_distance(src=q, dst=p)  classification/modules/pointnet2_utils.py:15-25 */
static float sqdist_expanded(const float *q, float qq, const float *p, float pp) {
  float dot = fmaf(q[2], p[2], fmaf(q[1], p[1], q[0] * p[0]));   /* :22 torch.matmul */
  float d = -2.0f * dot;
  d = d + qq;                                                     /* :23 */
  d = d + pp;                                                     /* :24 */
  return d;
}

/* farthest_point_sample(cuda=False)  classification/modules/pointnet2_utils.py:58-75
 * xyz (b,n,3); start (b) first picks (the reference draws them with torch.randint, :66);
 * idx (b,m). */
void oracle_fps(int b, int n, int m, const float *xyz, const int *start, int *idx) {
#pragma omp parallel for schedule(dynamic) if (b > 1)          /* clouds are independent: threads only change the wall time */
  for (int bi = 0; bi < b; ++bi) {
    float *dist = (float *)malloc(sizeof(float) * (size_t)n);
    const float *pts = xyz + (size_t)bi * n * 3;
    for (int k = 0; k < n; ++k) dist[k] = 1e10f;                  /* :65 */
    int far = start ? start[bi] : 0;
    for (int i = 0; i < m; ++i) {
      idx[(size_t)bi * m + i] = far;                              /* :69 */
      const float cx = pts[far * 3], cy = pts[far * 3 + 1], cz = pts[far * 3 + 2];
      int best = 0;
      float bestv = -1.0f;
      for (int k = 0; k < n; ++k) {
        const float dx = pts[k * 3] - cx, dy = pts[k * 3 + 1] - cy, dz = pts[k * 3 + 2] - cz;
        const float d = (dx * dx + dy * dy) + dz * dz;            /* :71 */
        if (d < dist[k]) dist[k] = d;                             /* :72-73 */
        if (dist[k] > bestv) { bestv = dist[k]; best = k; }       /* :74 first maximum */
      }
      far = best;
    }
    free(dist);
  }
}

/* query_ball_point(cuda=False)  classification/modules/pointnet2_utils.py:85-99
 * radius2 = (float)(radius**2) as the tensor/scalar comparison at :90 does. */
void oracle_ballquery(int b, int n, int m, float radius2, int nsample, const float *new_xyz,
                      const float *xyz, int *idx) {
#pragma omp parallel for schedule(dynamic) if (b > 1)
  for (int bi = 0; bi < b; ++bi) {
    const float *pts = xyz + (size_t)bi * n * 3;
    for (int s = 0; s < m; ++s) {
      const float *q = new_xyz + ((size_t)bi * m + s) * 3;
      const float qq = sqnorm3(q[0], q[1], q[2]);
      int *row = idx + ((size_t)bi * m + s) * nsample;
      int cnt = 0;
      for (int k = 0; k < n && cnt < nsample; ++k) {
        const float *p = pts + k * 3;
        const float d = sqdist_expanded(q, qq, p, sqnorm3(p[0], p[1], p[2]));
        if (!(d > radius2)) row[cnt++] = k;                       /* :90-91: ascending index order */
      }
      const int first = cnt ? row[0] : 0;                         /* empty ball -> zeros (see header) */
      for (int j = cnt; j < nsample; ++j) row[j] = first;         /* :92-94 */
    }
  }
}

/* query_knn_point(cuda=False)  classification/modules/pointnet2_utils.py:109-111
 * ascending by (distance, index).  dist2 optional. */
void oracle_knn(int b, int n, int m, int k, const float *xyz, const float *new_xyz, int *idx,
                float *dist2) {
#pragma omp parallel for schedule(dynamic) if (b > 1)
  for (int bi = 0; bi < b; ++bi) {
    float *bd = (float *)malloc(sizeof(float) * (size_t)k);
    int *bi_ = (int *)malloc(sizeof(int) * (size_t)k);
    const float *pts = xyz + (size_t)bi * n * 3;
    for (int s = 0; s < m; ++s) {
      const float *q = new_xyz + ((size_t)bi * m + s) * 3;
      const float qq = sqnorm3(q[0], q[1], q[2]);
      int cnt = 0;
      for (int p = 0; p < n; ++p) {
        const float *pp = pts + p * 3;
        const float d = sqdist_expanded(q, qq, pp, sqnorm3(pp[0], pp[1], pp[2]));
        if (cnt == k && !(d < bd[k - 1])) continue;
        int j = cnt < k ? cnt++ : k - 1;
        while (j > 0 && d < bd[j - 1]) { bd[j] = bd[j - 1]; bi_[j] = bi_[j - 1]; --j; }
        bd[j] = d; bi_[j] = p;
      }
      for (int j = 0; j < k; ++j) {
        idx[((size_t)bi * m + s) * k + j] = j < cnt ? bi_[j] : 0;
        if (dist2) dist2[((size_t)bi * m + s) * k + j] = j < cnt ? bd[j] : INFINITY;
      }
    }
    free(bd); free(bi_);
  }
}

/* UmbrellaSurfaceConstructor up to the input of self.mlps, self-query case
 * classification/modules/repsurface_utils.py:276-293 (+ group_by_umbrella :112-132),
 * classification/modules/recons_utils.py:27-57, 82-90, 108-124, 152-176,
 * classification/modules/polar_utils.py:10-31.
 * feat (b, n, k-1, 10) = [centroid(3), polar(3), normal(3), const(1)];  knn_idx (b,n,k) optional;
 * near_tie (b,n) optional: 1 where two azimuth keys were closer than PHI_TIE (the order is then
 * decided by the exact cross-product sign, which PyTorch's rounding need not reproduce). */
/* The triangle fan over g ring offsets (in kNN order), shared by the classification constructor and the
 * segmentation one.  rotate: azimuth key after segmentation/modules/repsurface_utils.py:71-74 (_fixed_rotate;
 * torch.matmul K=3 = fma chain, probed).  seg_order: channels [polar, normal, const, centroid]
 * (segmentation/modules/repsurface_utils.py:320) instead of [centroid, polar, normal, const]
 * (classification/modules/repsurface_utils.py:290).  work: 6*g + 10*g floats. */
static int fan_features(int g, float *ox, float *oy, float *oz, float flip, int rotate, int seg_order,
                        float *work, float *o) {
  float *key = work, *kx = key + g, *ky = kx + g, *un = ky + g;   /* un: per triangle u(3) c(3) polar(3) pos */
  int tie = 0;
  for (int j = 0; j < g; ++j) {
    if (rotate) {
      kx[j] = fmaf(oz[j], -0.5f, fmaf(oy[j], 0.7071f, ox[j] * 0.5f));
      ky[j] = fmaf(oz[j], 0.5f, fmaf(oy[j], 0.7071f, ox[j] * -0.5f));
    } else { kx[j] = ox[j]; ky[j] = oy[j]; }
    key[j] = atan2f(ky[j], kx[j]) / TWO_PI_F + 0.5f;                              /* polar_utils.py:22,29 */
  }
  /* stable insertion sort by azimuth (argsort; stable for <= 16 keys, probed) */
  for (int i = 1; i < g; ++i) {
    int j = i;
    while (j > 0) {
      const float diff = key[j] - key[j - 1];
      int before;
      if (fabsf(diff) <= PHI_TIE) {
        const double cr = (double)kx[j - 1] * (double)ky[j] - (double)kx[j] * (double)ky[j - 1];
        before = cr < 0.0;
        tie = 1;
      } else before = diff < 0.0f;
      if (!before) break;
      float t;
      t = key[j]; key[j] = key[j - 1]; key[j - 1] = t;
      t = kx[j]; kx[j] = kx[j - 1]; kx[j - 1] = t;
      t = ky[j]; ky[j] = ky[j - 1]; ky[j - 1] = t;
      t = ox[j]; ox[j] = ox[j - 1]; ox[j - 1] = t;
      t = oy[j]; oy[j] = oy[j - 1]; oy[j - 1] = t;
      t = oz[j]; oz[j] = oz[j - 1]; oz[j - 1] = t;
      --j;
    }
  }
  for (int j = 0; j < g; ++j) {
    const int j2 = (j + 1 == g) ? 0 : j + 1;                                      /* roll(-1) */
    const float ax = ox[j], ay = oy[j], az = oz[j], bx = ox[j2], by = oy[j2], bz = oz[j2];
    const float nx = fmaf(ay, bz, -(az * by));                                    /* recons_utils.py cal_normal: cross */
    const float ny = fmaf(az, bx, -(ax * bz));
    const float nz = fmaf(ax, by, -(ay * bx));
    const float len = sqrtf(fmaf(nz, nz, fmaf(ny, ny, nx * nx)));                /* torch.norm */
    float *t = un + j * 10;
    t[0] = nx / len; t[1] = ny / len; t[2] = nz / len;
    t[3] = ((0.0f + ax) + bx) / 3.0f;                                             /* cal_center: mean of 3 vertices */
    t[4] = ((0.0f + ay) + by) / 3.0f;
    t[5] = ((0.0f + az) + bz) / 3.0f;
  }
  const float pm = (un[0] > 0.0f) ? 1.0f : -1.0f;                                 /* first triangle's x positive */
  int first = 0, found = 0;
  for (int j = 0; j < g; ++j) {
    float *t = un + j * 10;
    for (int c = 0; c < 3; ++c) t[c] = (t[c] * pm) * flip;
    const float rho = sqrtf(sqnorm3(t[3], t[4], t[5]));                           /* polar_utils.py:19 */
    t[6] = rho;
    t[7] = (rho == 0.0f) ? 0.0f : acosf(t[5] / rho) / PI_F;                       /* :21,24-25,28 */
    t[8] = atan2f(t[4], t[3]) / TWO_PI_F + 0.5f;                                  /* :22,29 */
    t[9] = ((t[0] * t[3] + t[1] * t[4]) + t[2] * t[5]) / SQRT3_F;                 /* cal_const */
    const int bad = isnan(t[0]) || isnan(t[1]) || isnan(t[2]);                    /* check_nan_umb mask */
    if (!bad && !found) { first = j; found = 1; }                                 /* argmax(~mask) */
  }
  for (int j = 0; j < g; ++j) {
    const float *t = un + j * 10;
    const int bad = isnan(t[0]) || isnan(t[1]) || isnan(t[2]);
    const float *s = bad ? un + first * 10 : t;
    float *r = o + j * 10;
    if (seg_order) {
      r[0] = t[6]; r[1] = t[7]; r[2] = t[8];                                      /* polar: never patched */
      r[3] = s[0]; r[4] = s[1]; r[5] = s[2];
      r[6] = s[9];
      r[7] = s[3]; r[8] = s[4]; r[9] = s[5];
    } else {
      r[0] = s[3]; r[1] = s[4]; r[2] = s[5];                                      /* centroid */
      r[3] = t[6]; r[4] = t[7]; r[5] = t[8];                                      /* polar: never patched */
      r[6] = s[0]; r[7] = s[1]; r[8] = s[2];                                      /* normal */
      r[9] = s[9];                                                                /* const */
    }
  }
  return tie;
}

void oracle_umbrella(int b, int n, int k, const float *xyz, const float *inv_sign, int *knn_idx,
                     float *feat, unsigned char *near_tie) {
  const int g = k - 1;
#pragma omp parallel for schedule(dynamic) if (b > 1)
  for (int bi = 0; bi < b; ++bi) {
    int *nn = (int *)malloc(sizeof(int) * (size_t)k);
    float *ox = (float *)malloc(sizeof(float) * 19 * (size_t)g);
    float *oy = ox + g, *oz = oy + g, *work = oz + g;
    const float *pts = xyz + (size_t)bi * n * 3;
    for (int q = 0; q < n; ++q) {
      oracle_knn(1, n, 1, k, pts, pts + q * 3, nn, NULL);                         /* :115 */
      if (knn_idx) memcpy(knn_idx + ((size_t)bi * n + q) * k, nn, sizeof(int) * (size_t)k);
      for (int j = 0; j < g; ++j) {                                               /* :117-123: drop the nearest */
        const float *p = pts + nn[j + 1] * 3;
        ox[j] = p[0] - pts[q * 3]; oy[j] = p[1] - pts[q * 3 + 1]; oz[j] = p[2] - pts[q * 3 + 2];
      }
      const int tie = fan_features(g, ox, oy, oz, inv_sign ? inv_sign[bi] : 1.0f, 0, 0, work,
                                   feat + ((size_t)bi * n + q) * (size_t)(g * 10));
      if (near_tie) near_tie[(size_t)bi * n + q] = (unsigned char)tie;
    }
    free(nn); free(ox);
  }
}

/* sample_and_group feature assembly  classification/modules/repsurface_utils.py:36-57
 * out (b*m*nsample, cpos+cn+cf), channels-last. */
void oracle_group_features(int b, int n, int m, int nsample, int cn, int cf, int polar,
                           const float *center, const float *new_center, const float *normal,
                           const float *feature, const int *idx, float *out) {
  const int cpos = polar ? 6 : 3, ctot = cpos + cn + cf;
#pragma omp parallel for schedule(static) if (b > 1)
  for (int bi = 0; bi < b; ++bi)
    for (int s = 0; s < m; ++s)
      for (int j = 0; j < nsample; ++j) {
        const size_t r = ((size_t)bi * m + s) * nsample + j;
        const size_t src = (size_t)bi * n + idx[r];
        float *o = out + r * ctot;
        const float *nc = new_center + ((size_t)bi * m + s) * 3;
        const float dx = center[src * 3] - nc[0], dy = center[src * 3 + 1] - nc[1], dz = center[src * 3 + 2] - nc[2];
        o[0] = dx; o[1] = dy; o[2] = dz;                                          /* :45 */
        if (polar) {                                                              /* :49-51 */
          const float rho = sqrtf(sqnorm3(dx, dy, dz));
          o[3] = rho;
          o[4] = (rho == 0.0f) ? 0.0f : acosf(dz / rho) / PI_F;
          o[5] = atan2f(dy, dx) / TWO_PI_F + 0.5f;
        }
        for (int c = 0; c < cn; ++c) o[cpos + c] = normal[src * cn + c];          /* :40 */
        for (int c = 0; c < cf; ++c) o[cpos + cn + c] = feature[src * cf + c];    /* :53 */
      }
}

/* sample_and_group_all  classification/modules/repsurface_utils.py:62-88 */
void oracle_group_all_features(int b, int n, int cn, int cf, int polar, const float *center,
                               const float *normal, const float *feature, float *out) {
  const int cpos = polar ? 6 : 3, ctot = cpos + cn + cf;
  for (size_t r = 0; r < (size_t)b * n; ++r) {
    float *o = out + r * ctot;
    const float x = center[r * 3], y = center[r * 3 + 1], z = center[r * 3 + 2];
    o[0] = x; o[1] = y; o[2] = z;
    if (polar) {
      const float rho = sqrtf(sqnorm3(x, y, z));
      o[3] = rho;
      o[4] = (rho == 0.0f) ? 0.0f : acosf(z / rho) / PI_F;
      o[5] = atan2f(y, x) / TWO_PI_F + 0.5f;
    }
    for (int c = 0; c < cn; ++c) o[cpos + c] = normal[r * cn + c];
    for (int c = 0; c < cf; ++c) o[cpos + cn + c] = feature[r * cf + c];
  }
}

/* three nearest neighbours: classification/modules/pointops/src/interpolation/
 * interpolation_cuda_kernel.cu:134-177 (sequential scan, strict '<', squared distances). */
void oracle_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx) {
  for (int bi = 0; bi < b; ++bi)
    for (int q = 0; q < n; ++q) {
      const float *u = unknown + ((size_t)bi * n + q) * 3;
      float d1 = INFINITY, d2 = INFINITY, d3 = INFINITY;
      int i1 = 0, i2 = 0, i3 = 0;
      for (int k = 0; k < m; ++k) {
        const float *p = known + ((size_t)bi * m + k) * 3;
        const float dx = u[0] - p[0], dy = u[1] - p[1], dz = u[2] - p[2];
        const float d = (dx * dx + dy * dy) + dz * dz;
        if (d < d1) { d3 = d2; i3 = i2; d2 = d1; i2 = i1; d1 = d; i1 = k; }
        else if (d < d2) { d3 = d2; i3 = i2; d2 = d; i2 = k; }
        else if (d < d3) { d3 = d; i3 = k; }
      }
      const size_t o = ((size_t)bi * n + q) * 3;
      dist2[o] = d1; dist2[o + 1] = d2; dist2[o + 2] = d3;
      idx[o] = i1; idx[o + 1] = i2; idx[o + 2] = i3;
    }
}

/* interpolation forward (channels-last): interpolation_cuda_kernel.cu:181-195 */
void oracle_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                              const float *weight, float *out) {
  for (int bi = 0; bi < b; ++bi)
    for (int q = 0; q < n; ++q) {
      const size_t r = (size_t)bi * n + q;
      for (int ch = 0; ch < c; ++ch) {
        const float *base = points + (size_t)bi * m * c + ch;
        const float v0 = weight[r * 3] * base[(size_t)idx[r * 3] * c];
        const float v1 = weight[r * 3 + 1] * base[(size_t)idx[r * 3 + 1] * c];
        const float v2 = weight[r * 3 + 2] * base[(size_t)idx[r * 3 + 2] * c];
        out[r * c + ch] = (v0 + v1) + v2;
      }
    }
}

/* Packed-batch FPS: segmentation/modules/pointops/src/sampling/sampling_cuda_kernel.cu:14-129, launched by
 * pointops.py:31-49 with n = the largest cloud of the batch and tmp = 1e10.
 * First pick = first row of the segment (:39); d = (dx*dx + dy*dy) + dz*dz on direct differences (:53), running minimum
 * in tmp (:54-55).  Arg-max = what the kernel's strided scan + shared-memory tree computes: thread tid scans rows
 * start+tid, start+tid+bs, ... keeping the first strict maximum (:56-57, best = -1 / besti = start), __update (:7-12)
 * keeps the lower slot on equal values at every level of the tree (slot t against t + bs/2, then t + bs/4, ... t + 1),
 * so among equal distances the winner is the thread whose id has the lowest BIT-REVERSED value (the last level
 * prefers even ids, the one before ids = 0 mod 4, ...), then the lowest row of that thread; tid = (row - start) mod bs,
 * bs = opt_n_threads(n) = min(2^floor(log2 n), 1024) (cuda_utils.h:10-13).
 * Pinned against the kernel itself (oracle/_ref, tests/test_oracle_ref.py), ties included. */
static int ref_opt_n_threads(int work_size) {
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int t = 1 << pow_2;
  if (t > 1024) t = 1024;
  return t < 1 ? 1 : t;
}

static int bit_reverse(int v, int bs) {
  int r = 0;
  for (int m = 1; m < bs; m <<= 1) { r = (r << 1) | (v & 1); v >>= 1; }
  return r;
}

int oracle_fps_block_size(int b, const int *offset) {
  int n_max = offset[0];
  for (int i = 1; i < b; ++i) if (offset[i] - offset[i - 1] > n_max) n_max = offset[i] - offset[i - 1];
  return ref_opt_n_threads(n_max);
}

void oracle_fps_offset(int b, const float *xyz, const int *offset, const int *new_offset, int *idx) {
  const int bs = oracle_fps_block_size(b, offset);
  for (int bi = 0; bi < b; ++bi) {
    const int r0 = bi ? offset[bi - 1] : 0, r1 = offset[bi];
    const int o0 = bi ? new_offset[bi - 1] : 0, o1 = new_offset[bi];
    if (r1 <= r0 || o1 <= o0) continue;
    float *tmp = (float *)malloc(sizeof(float) * (size_t)(r1 - r0));
    for (int k = 0; k < r1 - r0; ++k) tmp[k] = 1e10f;
    int old = r0;
    idx[o0] = r0;
    for (int j = o0 + 1; j < o1; ++j) {
      const float x1 = xyz[old * 3], y1 = xyz[old * 3 + 1], z1 = xyz[old * 3 + 2];
      float best = -1.0f;
      int besti = r0, best_tid = 0;
      for (int k = r0; k < r1; ++k) {
        const float dx = xyz[k * 3] - x1, dy = xyz[k * 3 + 1] - y1, dz = xyz[k * 3 + 2] - z1;
        const float d = (dx * dx + dy * dy) + dz * dz;
        const float d2 = d < tmp[k - r0] ? d : tmp[k - r0];
        tmp[k - r0] = d2;
        const int tid = bit_reverse((k - r0) % bs, bs);
        if (d2 > best || (d2 == best && tid < best_tid)) { best = d2; besti = k; best_tid = tid; }
      }
      old = besti;
      idx[j] = old;
    }
    free(tmp);
  }
}

/* Packed-batch kNN: segmentation/modules/pointops/src/knnquery/knnquery_cuda_kernel.cu:65-108
 * (direct differences, best_dist = 1e10 / best_idx = start initialisation :86-87, strict '<',
 * heap-sorted ascending).  PARITY UNPINNED (CUDA-only reference). */
void oracle_knn_offset(int m, int k, int b, const float *xyz, const float *new_xyz, const int *offset,
                       const int *new_offset, int *idx, float *dist2) {
  float *bd = (float *)malloc(sizeof(float) * (size_t)k);
  int *bi_ = (int *)malloc(sizeof(int) * (size_t)k);
  for (int q = 0; q < m; ++q) {
    int c = 0;
    while (c < b - 1 && q >= new_offset[c]) ++c;                                  /* get_bt_idx :51-62 */
    const int start = c ? offset[c - 1] : 0, end = offset[c];
    for (int j = 0; j < k; ++j) { bd[j] = 1e10f; bi_[j] = start; }
    const float *qq = new_xyz + (size_t)q * 3;
    for (int p = start; p < end; ++p) {
      const float dx = qq[0] - xyz[p * 3], dy = qq[1] - xyz[p * 3 + 1], dz = qq[2] - xyz[p * 3 + 2];
      const float d = (dx * dx + dy * dy) + dz * dz;
      if (!(d < bd[k - 1])) continue;
      int j = k - 1;
      while (j > 0 && d < bd[j - 1]) { bd[j] = bd[j - 1]; bi_[j] = bi_[j - 1]; --j; }
      bd[j] = d; bi_[j] = p;
    }
    for (int j = 0; j < k; ++j) { idx[(size_t)q * k + j] = bi_[j]; if (dist2) dist2[(size_t)q * k + j] = bd[j]; }
  }
  free(bd); free(bi_);
}

/* Segmentation umbrella fan (segmentation/modules/repsurface_utils.py:77-98,305-321): the k nearest
 * neighbours INCLUDING the query stay in the ring (k triangles); rotate = sort='fix'.
 * knn_idx (m,k) global rows; new_offset (b) running query ends; inv_sign (b) or NULL.
 * feat (m, k, 10) = [polar, normal, const, centroid].  The torch part of this function is pinned against
 * the reference's own code run on CPU (tests/golden/make_golden_seg.py); the kNN feeding it is the
 * restated CUDA kernel (parity unpinned). */
void oracle_umbrella_fan_offset(int m, int k, int b, int rotate, const float *xyz, const float *new_xyz,
                                const int *knn_idx, const int *new_offset, const float *inv_sign,
                                float *feat, unsigned char *near_tie) {
  float *ox = (float *)malloc(sizeof(float) * 19 * (size_t)k);
  float *oy = ox + k, *oz = oy + k, *work = oz + k;
  for (int q = 0; q < m; ++q) {
    int c = 0;
    while (c < b - 1 && q >= new_offset[c]) ++c;
    for (int j = 0; j < k; ++j) {                                                 /* :86-88 */
      const float *p = xyz + (size_t)knn_idx[(size_t)q * k + j] * 3;
      ox[j] = p[0] - new_xyz[q * 3]; oy[j] = p[1] - new_xyz[q * 3 + 1]; oz[j] = p[2] - new_xyz[q * 3 + 2];
    }
    const int tie = fan_features(k, ox, oy, oz, inv_sign ? inv_sign[c] : 1.0f, rotate, 1, work,
                                 feat + (size_t)q * (size_t)(k * 10));
    if (near_tie) near_tie[q] = (unsigned char)tie;
  }
  free(ox);
}

/* Interpolation weights: segmentation/modules/repsurface_utils.py:262-265 (dist = sqrt(dist2),
 * pointops.py:127; torch.sum over 3 = (a+b)+c, probed). */
void oracle_interp_weights(long long n, const float *dist2, float *weight) {
  for (long long r = 0; r < n; ++r) {
    const float r0 = 1.0f / (sqrtf(dist2[r * 3]) + 1e-8f);
    const float r1 = 1.0f / (sqrtf(dist2[r * 3 + 1]) + 1e-8f);
    const float r2 = 1.0f / (sqrtf(dist2[r * 3 + 2]) + 1e-8f);
    const float s = (r0 + r1) + r2;
    weight[r * 3] = r0 / s; weight[r * 3 + 1] = r1 / s; weight[r * 3 + 2] = r2 / s;
  }
}
