/*
 * repsurf_hip.h — C ABI of librepsurf_hip.so, the MI355X (gfx950) hot path of RepSurf-U.
 *
 * Every entry point is `extern "C"`, takes plain device pointers + sizes and a
 * `void* stream` (a hipStream_t; NULL = the legacy default stream), allocates
 * nothing, retains nothing across calls and never synchronises the host.
 * Return value: 0 on success, RS_ERR_ARG for an invalid argument, or
 * (RS_ERR_HIP_BASE + hipError_t) when the launch failed; rs_last_error() gives
 * the text.  The reference exits the process on a failed launch
 * (classification/modules/pointops/src/ballquery/ballquery_cuda_kernel.cu:95-100);
 * we return the code and the Python host raises RuntimeError.
 *
 * Layouts: all point data is fp32, row-major, channels-last: xyz (b, n, 3),
 * per-point features (b, n, c).  Indices are int32 like the reference CUDA
 * path (classification/modules/pointops/functions/pointops.py:44,220,313).
 * Grouped shared-MLP tiles are (rows, channels) with
 * rows = ((b * npoint) + s) * nsample + k.
 *
 * Arithmetic contract: the index-producing kernels restate, operation by
 * operation, the reference's CPU/PyTorch path (`cuda=False` branches of
 * classification/modules/pointnet2_utils.py) so that FPS / ball-query / kNN
 * indices are bit-exact against it; see DESIGN.md §3 and oracle/geom_oracle.c.
 *
 * Each declaration cites the reference interface it replaces.
 */
#ifndef REPSURF_HIP_H
#define REPSURF_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define RS_OK 0
#define RS_ERR_ARG (-1)
#define RS_ERR_HIP_BASE 1000

/* ---- library ---------------------------------------------------------- */
const char *rs_last_error(void);          /* thread-local text of the last failure */
int rs_abi_version(void);                 /* bumped on any signature change */
int rs_device_info(int *cu_count, int *wave_size, int *lds_bytes, char *arch, int arch_len);

/* ---- sampling ---------------------------------------------------------
 * Farthest point sampling.  Replaces furthestsampling_cuda_launcher(b,n,m,dataset,temp,idxs)
 * (classification/modules/pointops/src/sampling/sampling_cuda_kernel.h:17, kernel .cu:58-168)
 * with the CPU-path semantics of farthest_point_sample
 * (classification/modules/pointnet2_utils.py:47-75): start[b] is the first pick
 * (the CPU path draws it with torch.randint; NULL = index 0 like the CUDA kernel),
 * d = ((dx*dx + dy*dy) + dz*dz) with separately rounded products, running min,
 * arg-max = lowest index among maxima.  `temp` (b*n floats) is accepted for
 * signature parity and only used when n is too large for the register-resident
 * kernel (n > 16384); it may be NULL otherwise.  idx: (b, m) int32. */
int rs_furthestsampling(int b, int n, int m, const float *xyz, const int *start,
                        float *temp, int *idx, void *stream);

/* Packed-batch FPS of the segmentation path: replaces
 * furthestsampling_cuda_launcher(b, n_max, xyz, offset, new_offset, tmp, idx)
 * (segmentation/modules/pointops/src/sampling/sampling_cuda_kernel.cu:14-171):
 * cloud i owns rows [offset[i-1], offset[i]); picks new_offset[i]-new_offset[i-1]
 * samples starting from its first row; idx holds global row numbers. */
/* Exactly equal distances: the winner is the row the reference kernel's strided scan + shared-memory tree keeps --
 * reference thread t = (row - first row) mod bs, bs = opt_n_threads(n_max) (cuda_utils.h:10-13); lowest BIT-REVERSED t,
 * then the lowest row of that thread (sampling_cuda_kernel.cu:44-58, __update :7-12). */
int rs_furthestsampling_offset(int b, int n_max, const float *xyz, const int *offset,
                               const int *new_offset, float *temp, int *idx, void *stream);

/* Row gather out[b, j, :] = points[b, idx[b, j], :] on channels-last data.
 * Replaces gathering_forward/backward_cuda_launcher
 * (classification/modules/pointops/src/sampling/sampling_cuda_kernel.h:15-16),
 * which work on (b, c, n); the transposes around them
 * (classification/modules/pointnet2_utils.py:31-35) disappear. */
int rs_gather_rows(int b, int n, int m, int c, const float *points, const int *idx,
                   float *out, void *stream);
/* grad_points[b, idx[b,j], :] += grad_out[b, j, :]  (grad_points pre-zeroed by caller) */
int rs_gather_rows_backward(int b, int n, int m, int c, const float *grad_out, const int *idx,
                            float *grad_points, void *stream);
/* the same with the gathered rows' count as device data (b = 1, packed batches under a captured capacity: rs_bn_item): rows
 * [min(m, *rows_dev), m) of grad_out are not read */
int rs_gather_rows_backward_dev(int b, int n, int m, int c, const float *grad_out, const int *idx,
                                float *grad_points, const int *rows_dev, void *stream);

/* ---- ball query --------------------------------------------------------
 * Replaces ballquery_cuda_launcher_fast(b,n,m,radius,nsample,new_xyz,xyz,idx,stream)
 * (classification/modules/pointops/src/ballquery/ballquery_cuda_kernel.h:17, .cu:47-101)
 * with the semantics of query_ball_point(cuda=False)
 * (classification/modules/pointnet2_utils.py:78-99): squared distance by the
 * expanded formula of square_distance (:15-25), a point is inside when
 * NOT (d > radius2) where radius2 = float32(radius**2 computed in double); the
 * first nsample inside points in ascending index order, padded with the first.
 * A centre with an empty ball gets zeros (the CUDA kernel's pre-zeroed rows;
 * the CPU path would index out of range).  idx: (b, m, nsample) int32. */
/* cnt (optional, (b, m) int32): number of DISTINCT neighbours in each row, min(hits, nsample), at least 1;
 * slots [cnt, nsample) are the padding copies of slot 0 (input of rs_group_features_compact). */
int rs_ballquery(int b, int n, int m, float radius2, int nsample, const float *new_xyz,
                 const float *xyz, int *idx, int *cnt, void *stream);

/* Round 6: the ball query's per-cloud cell list as a REUSABLE image.  rs_ballquery_grid_build counting-sorts every cloud by cells of edge
 * >= 1.001 r into `image` (rs_ballquery_grid_bytes(b, n) bytes, 16-byte aligned: per cloud a header {lo[3], 1/cell, cells per axis[3],
 * cell count}, the points as float4 (x, y, z, |p|^2) in cell order, their u16 indices and the cells' running starts);
 * rs_ballquery_grid_query answers any set of centres against it -- rows bit-identical to rs_ballquery (same distance arithmetic,
 * threshold, order and padding; same nsample <= 64 and 64 <= n <= 4096 limits as the cell-list form of rs_ballquery).  The radius of the
 * query must be the radius the image was built for (the cell edge is derived from it); one build serves every query on the same
 * coordinates (the fused rs_ballquery rebuilds the list in every launch: 18.6 of 66 us at 2 048 clouds). */
long long rs_ballquery_grid_bytes(int b, int n);
int rs_ballquery_grid_build(int b, int n, float radius2, const float *xyz, void *image, void *stream);
int rs_ballquery_grid_query(int b, int n, int m, float radius2, int nsample, const float *new_xyz, const void *image,
                            int *idx, int *cnt, void *stream);

/* ---- kNN ----------------------------------------------------------------
 * Replaces knnquery_cuda_launcher(b,n,m,nsample,xyz,new_xyz,idx,dist2,stream)
 * (classification/modules/pointops/src/knnquery/knnquery_cuda_kernel.h:14) with the
 * semantics of query_knn_point(cuda=False) (classification/modules/pointnet2_utils.py:102-111):
 * expanded-formula distances, ascending by (distance, index).  dist2 may be NULL.
 * Any nsample <= n: register-list kernels up to 64, one wave per query by repeated minimum beyond (the reference
 * operator accepts up to 200, knnquery_cuda_kernel.cu:21-22; 100 in knnquery_heap_cuda_kernel.cu:67-68). */
int rs_knnquery(int b, int n, int m, int nsample, const float *xyz, const float *new_xyz,
                int *idx, float *dist2, void *stream);

/* Packed-batch kNN of the segmentation path: replaces knnquery_cuda_launcher
 * (segmentation/modules/pointops/src/knnquery/knnquery_cuda_kernel.cu:65-108):
 * direct-difference distances, strict '<' replacement, ascending output,
 * dist2 (m, nsample) holds squared distances (Python takes the sqrt).  Any nsample (the reference: <= 100, :86-87);
 * a cloud with fewer rows leaves the tail of a list at (1e10, first row of the cloud) like the reference. */
int rs_knnquery_offset(int m, int nsample, const float *xyz, const float *new_xyz,
                       const int *offset, const int *new_offset, int b,
                       int *idx, float *dist2, void *stream);

/* The same lists, bit for bit, through per-cloud uniform grids (round 4; csrc/grid_knn.hip): the scan above evaluates every
 * row of the cloud per query, a query's nsample nearest rows sit in a fraction of a percent of it.
 * rs_knn_grid_build: per cloud, bounding box -> cubic cells (about `per_cell` rows each, at most RS_KNN_GRID_CELLS) ->
 *   rows in cell-sorted order.  Workspaces: sorted (ntot x 4 floats: x, y, z, row bits; 16-byte aligned),
 *   starts (b x (RS_KNN_GRID_CELLS + 1) ints), grid (b x 16 floats; 16-byte aligned).
 * rs_knn_grid_query: per query the cells around its own, ring by ring, until the nsample-th distance is below the
 *   distance to every unvisited cell (or the whole grid was visited); nsample <= 64; dist2 may be NULL.
 *   new_offset: running ends of the QUERY rows per cloud (the cloud of a query selects the grid).
 *   max_queries / max_rows: the largest number of queries / searched rows any cloud holds when the HOST knows them (they size the
 *   launch and the LDS staging of a cloud's rows), 0 = unknown (m workgroup slots per cloud; clouds above 4096 rows are read
 *   from global memory either way). */
#define RS_KNN_GRID_CELLS 4096
int rs_knn_grid_build(int b, const float *xyz, const int *offset, float per_cell, float *sorted, int *starts, float *grid,
                      void *stream);
int rs_knn_grid_query(int m, int nsample, int b, int max_queries, int max_rows, const float *new_xyz, const int *new_offset,
                      const float *sorted, const int *starts, const float *grid, int *idx, float *dist2, void *stream);

/* Segmentation umbrella fan: everything UmbrellaSurfaceConstructor.forward computes between the kNN and
 * self.mlps (segmentation/modules/repsurface_utils.py:77-98 group_by_umbrella_v2 / :101-122 group_by_umbrella,
 * :305-321; segmentation/modules/recons_utils.py:10-45,48-57,84-100,128-151; polar_utils.py:10-31).
 * knn_idx (m, k) are the global rows of the k nearest neighbours of each query INCLUDING the query itself
 * (the segmentation variant does not drop it), new_offset (b) the running query ends per cloud, inv_sign (b)
 * the per-cloud +-1 of cal_normal(random_inv) (recons_utils.py:28-43) or NULL, rotate != 0 selects sort='fix'
 * (azimuth after _fixed_rotate, :71-74).  feat (m, k, 10) = [polar(3), normal(3), const(1), centroid(3)]
 * per fan triangle (:320).  k in {5, 9, 13, 17}. */
int rs_umbrella_fan_offset(int m, int k, int b, int rotate, const float *xyz, const float *new_xyz,
                           const int *knn_idx, const int *new_offset, const float *inv_sign,
                           float *feat, void *stream);

/* Inverse-distance interpolation weights (segmentation/modules/repsurface_utils.py:262-265,
 * segmentation/modules/pointops/functions/pointops.py:262-265): dist2 (n, 3) squared distances of the
 * three nearest neighbours -> weight (n, 3) = r_i / ((r_0 + r_1) + r_2), r_i = 1 / (sqrt(dist2_i) + 1e-8). */
int rs_interp_weights(long long n, const float *dist2, float *weight, void *stream);
/* rs_three_interpolate with the feature-propagation stage's skip connection and activation folded in
 * (segmentation/modules/repsurface_utils.py:266-270: interpolated + skip, ReLU): out = relu?(interpolated + add?) (add (b, n, c)
 * or NULL; relu != 0: ReLU).  Backward: the incoming gradient counts where fwd_out > 0 (fwd_out NULL: everywhere); it is
 * scattered into grad_points (b, m, c), which the caller zeroed, and -- grad_add != NULL -- also written out masked (b, n, c):
 * the gradient of `add`. */
int rs_three_interpolate_fused(int b, int c, int m, int n, const float *points, const int *idx, const float *weight,
                               const float *add, int relu, float *out, void *stream);
int rs_three_interpolate_fused_backward(int b, int c, int n, int m, const float *grad_out, const float *fwd_out,
                                        const int *idx, const float *weight, float *grad_points, float *grad_add, void *stream);
/* the same with the fine rows' count as device data (b = 1; see rs_bn_item): rows [min(n, *rows_dev), n) are not read or scattered */
int rs_three_interpolate_fused_backward_dev(int b, int c, int n, int m, const float *grad_out, const float *fwd_out,
                                            const int *idx, const float *weight, float *grad_points, float *grad_add,
                                            const int *rows_dev, void *stream);
/* Feature propagation, first layers (segmentation/modules/repsurface_utils.py:256-270; round 4): out = relu(interpolate(BN_f(points))
 * + BN_s(add)), both BatchNorms as per-channel (scale, shift) applied on the fly to the raw Linear outputs (z = fma(scale, y,
 * shift): what the BatchNorm pass computes) -- and its backward: g = grad_out * (fwd_out > 0) to grad_add, scattered with the
 * weights into grad_points (zeroed by the caller), BN_s's backward sums {sum g, sum g * (add - mean) * invstd} to
 * partial (partial_blocks, 2, c) doubles (c <= 256).  rows_dev (optional, b = 1): the fine rows' count as device data -- min(n, *rows_dev)
 * rows are read, scattered and summed (see rs_bn_item). */
int rs_three_interpolate_affine(int b, int c, int m, int n, const float *points, const float *pscale, const float *pshift,
                                const int *idx, const float *weight, const float *add, const float *ascale, const float *ashift,
                                int relu, float *out, void *stream);
int rs_three_interpolate_affine_backward(int b, int c, int n, int m, const float *grad_out, const float *fwd_out, const int *idx,
                                         const float *weight, float *grad_points, float *grad_add, const float *add,
                                         const float *add_mean, const float *add_invstd, double *partial, int partial_blocks,
                                         const int *rows_dev, void *stream);
/* The interpolation's backward towards the coarse rows as a gather (round 4) over rs_inverse_index of idx (per = 3, built with the
 * geometry): grad_points (m_rows, c) WRITTEN, ascending sums, no atomics.  g: the masked gradient (rs_three_interpolate_affine_backward
 * with grad_points = NULL makes it), or NULL: grad_out where fwd_out > 0.  partial (optional): BatchNorm-backward sums of the coarse
 * rows' BatchNorm from y / mean / invstd (c <= 256). */
int rs_three_interpolate_backward_csr(long long m_rows, int c, const float *g, const float *grad_out, const float *fwd_out,
                                      const float *weight, const int *csr_off, const int *csr_edges, float *grad_points,
                                      const float *y, const float *mean, const float *invstd, double *partial, int partial_blocks,
                                      void *stream);

/* ---- umbrella surface constructor ---------------------------------------
 * Fuses group_by_umbrella + cal_normal + cal_center + xyz2sphere + cal_const +
 * check_nan_umb (classification/modules/repsurface_utils.py:112-132,276-293;
 * classification/modules/recons_utils.py:27-57,82-90,108-124,152-176;
 * classification/modules/polar_utils.py:10-31) for the self-query case
 * (new_xyz == xyz): kNN-k, drop the nearest, sort the k-1 offsets by azimuth,
 * build the triangle fan, emit per triangle [centre(3), polar(3), normal(3), pos(1)].
 * inv_sign: (b) floats of +-1 (the per-cloud random inversion drawn by the host
 * from the CPU generator, recons_utils.py:50) or NULL.  knn_idx (b, n, k) int32 is
 * optional (NULL to skip).  feat: (b, n, k-1, 10).  k in {5, 9, 13, 17} (group_size 4/8/12/16). */
int rs_umbrella_features(int b, int n, int k, const float *xyz, const float *inv_sign,
                         int *knn_idx, float *feat, void *stream);
/* The same through per-cloud uniform grids (round 4; csrc/grid_knn.hip): the search visits the cells around a point instead of its
 * whole cloud; same lists and features, bit for bit.  offset: (b) int32 = n, 2n, ... (the packed form of the dense batch);
 * workspaces as for rs_knn_grid_build (sorted: b*n x 4 floats, starts: b x (RS_KNN_GRID_CELLS + 1) ints, grid: b x 16 floats). */
int rs_umbrella_features_grid(int b, int n, int k, const float *xyz, const int *offset, const float *inv_sign, int *knn_idx,
                              float *feat, float *sorted, int *starts, float *grid, void *stream);

/* ---- grouping ------------------------------------------------------------
 * Builds the grouped shared-MLP input of sample_and_group
 * (classification/modules/repsurface_utils.py:15-59) in one pass, replacing
 * 3x grouping_forward_cuda_launcher_fast
 * (classification/modules/pointops/src/grouping/grouping_cuda_kernel.h:19) + subtract +
 * xyz2sphere + cat: row r = (b, s, j) of `out` (rows, 3 + 3*polar + cn + cf) is
 * [center[idx]-new_center (3), polar of that offset (3, if polar), normal[idx] (cn), feature[idx] (cf)].
 * feature may be NULL (cf = 0).  idx: (b, m, nsample).
 * pos_pad zero channels follow the position block and rows are `ldo` floats apart (0 = tight): with 3 position
 * channels, pos_pad = 1 and ldo % 4 == 0 keep the feature branch of the first layer 8-byte aligned, so the row GEMM
 * and weight gradient read it with vector loads (segmentation: 3 + 10 + C channels). */
int rs_group_features(int b, int n, int m, int nsample, int cn, int cf, int polar,
                      const float *center, const float *new_center, const float *normal,
                      const float *feature, const int *idx, float *out, int pos_pad, int ldo, void *stream);
/* Backward of the gathered (normal, feature) channels: scatter-adds
 * grad_out[:, cpos:cpos+cn] into grad_normal (b,n,cn) and grad_out[:, cpos+cn:] into
 * grad_feature (b,n,cf) (either may be NULL; both pre-zeroed by the caller).
 * Replaces grouping_backward_cuda_launcher (grouping_cuda_kernel.h:17). */
int rs_group_features_backward(int b, int n, int m, int nsample, int cn, int cf, int polar,
                               const float *grad_out, const int *idx, float *grad_normal,
                               float *grad_feature, int pos_pad, int ldo, void *stream);
/* the same with the group count as device data (b = 1; see rs_bn_item): groups [min(m, *groups_dev), m) are not read or scattered */
int rs_group_features_backward_dev(int b, int n, int m, int nsample, int cn, int cf, int polar,
                                   const float *grad_out, const int *idx, float *grad_normal,
                                   float *grad_feature, int pos_pad, int ldo, const int *groups_dev, void *stream);
/* Compacted form of rs_group_features: a ball-query row is cnt distinct neighbours followed by copies of the
 * first one (classification/modules/pointnet2_utils.py:92-94) and the shared MLP maps equal rows to equal outputs, so
 * only the distinct slots are materialised.  rows of group g = [offsets[g], offsets[g+1]) with
 * offsets = exclusive scan of cnt (rs_exclusive_scan; offsets[b*m] = total rows, consumed on the device).
 * Outputs (capacity b*m*nsample rows): out (rows, ctot), mult[row] = copies the row stands for
 * (nsample-cnt+1 for slot 0, else 1), grp[row], slot[row], src[row] = cloud*n + idx (gather/scatter address). */
int rs_exclusive_scan(int n, const int *in, int *out, void *stream);
/* The bookkeeping alone (offsets by rs_exclusive_scan of cnt, then mult / grp / slot / src): it needs the ball query's
 * (idx, cnt) only -- query_ball_point's padded rows, classification/modules/pointnet2_utils.py:78-99 -- so a pipelined step
 * builds it in its geometry stage, off the critical path. */
int rs_compact_index(int b, int n, int m, int nsample, const int *idx, const int *cnt, int *offsets, int *grp, int *slot,
                     int *src, float *mult, void *stream);
/* have_index != 0: offsets / mult / grp / slot / src come from rs_compact_index (read, not written); 0: `offsets` is given
 * and the other four are built here.  fps_idx (b*m) + new_normal (b*m, cn), both or neither: the same launch also writes
 * the centres' own normal rows, new_normal[g, :] = normal[cloud(g)*n + fps_idx[g], :] (index_points(normal, fps_idx),
 * classification/modules/repsurface_utils.py:31). */
int rs_group_features_compact(int b, int n, int m, int nsample, int cn, int cf, int polar,
                              const float *center, const float *new_center, const float *normal,
                              const float *feature, const int *idx, const int *cnt, const int *offsets,
                              float *out, float *mult, int *grp, int *slot, int *src, int have_index,
                              const int *fps_idx, float *new_normal, void *stream);
/* grad_normal / grad_feature [src[row], :] += grad_out[row, gathered channels] for row < *rows_dev
 * (gradients of the copies are already summed per row: one atomic per distinct neighbour), both tensors in one launch.
 * fps_idx (b*m) + grad_new_normal (b*m rows, ldg floats apart), both or neither: the backward of the centre rows,
 * grad_normal[cloud(g)*n + fps_idx[g], :] += grad_new_normal[g*ldg + :], in the same launch. */
int rs_group_features_compact_backward(long long capacity, const int *rows_dev, int cn, int cf, int polar,
                                       const float *grad_out, const int *src, float *grad_normal,
                                       float *grad_feature, int b, int n, int m, const int *fps_idx,
                                       const float *grad_new_normal, long long ldg, void *stream);
/* The same backward as a gather (round 4): rs_compact_csr (geometry stage) inverts `src` -- csr_off (b*n + 1), csr_rows (capacity):
 * the compacted rows that name each source point, ascending; centre_of (b*n): the group whose centre the point is, or -1 (see below) --
 * and rs_group_features_compact_backward_csr WRITES every element of grad_normal / grad_feature (no zero fill, no atomics, a fixed
 * summation order). */
int rs_compact_csr(int b, int n, int m, const int *src, const int *offsets, const int *fps_idx, int *csr_off, int *centre_of,
                   int *csr_rows, void *stream);
int rs_group_features_compact_backward_csr(int b, int n, int cn, int cf, int polar, const float *grad_out, const int *csr_off,
                                           const int *csr_rows, const int *centre_of, float *grad_normal, float *grad_feature,
                                           const float *grad_new_normal, long long ldg, const int *fps_idx, int m, void *stream);
/* (round 5) centre_of[p] >= 0: the one group whose centre p is; -1: none; -2 - g: p is the centre of SEVERAL groups, g the lowest --
 * FPS repeats a row when a cloud holds fewer distinct points than picks -- and the backward adds the centre-row gradient of every group
 * of that cloud with fps_idx == p, ascending (fps_idx (b, m): required next to grad_new_normal). */
/* The inverse of ANY gather index over a packed batch (round 4): src (E edges -> global source rows), edge_ends / point_ends (b):
 * running ends of the query rows (x `per` = edges: nsample of a grouping, 3 of an interpolation) and of the source rows per cloud
 * -> csr_off (P + 1), csr_edges (E): the edges reading each source row, ascending.  overflow (1 int, zeroed by the caller) counts the
 * clouds with more than max_points (<= 16384) source rows: their lists are then undefined.  Geometry only.
 * rs_group_features_backward_csr: rs_group_features_backward (b = 1 packed layout) as a gather over it -- every element of
 * grad_normal / grad_feature written once, ascending sums, no atomics; c0 = first gathered column (cpos + pad), ldo = row pitch. */
int rs_inverse_index(int b, int per, int max_points, const int *src, const int *edge_ends, const int *point_ends, int *csr_off,
                     int *csr_edges, int *overflow, void *stream);
int rs_group_features_backward_csr(long long points, int cn, int cf, int c0, int ldo, const float *grad_out, const int *csr_off,
                                   const int *csr_edges, float *grad_normal, float *grad_feature, void *stream);
/* group_all variant (sample_and_group_all, repsurface_utils.py:62-88):
 * row (b, j) = [center (3), polar of center (3, if polar), normal (cn), feature (cf)]. */
int rs_group_all_features(int b, int n, int cn, int cf, int polar, const float *center,
                          const float *normal, const float *feature, float *out, void *stream);

/* Plain grouping out[b, s, j, :] = points[b, idx[b,s,j], :] and its backward
 * (index_points(is_group=True), classification/modules/pointnet2_utils.py:28-35). */
int rs_group_rows(int b, int n, int m, int nsample, int c, const float *points,
                  const int *idx, float *out, void *stream);
int rs_group_rows_backward(int b, int n, int m, int nsample, int c, const float *grad_out,
                           const int *idx, float *grad_points, void *stream);

/* ---- three-NN interpolation (segmentation decoder; BASELINE config 4) ----
 * Replaces nearestneighbor_cuda_launcher_fast / interpolation_forward_cuda_launcher_fast /
 * interpolation_backward_cuda_launcher
 * (classification/modules/pointops/src/interpolation/interpolation_cuda_kernel.h:18-23;
 * kernels .cu:134-195,90-114), channels-last.  dist2 (b,n,3) squared distances
 * (direct differences, like the kernel), idx (b,n,3). */
int rs_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                int *idx, void *stream);
int rs_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                         const float *weight, float *out, void *stream);
int rs_three_interpolate_backward(int b, int c, int n, int m, const float *grad_out,
                                  const int *idx, const float *weight, float *grad_points,
                                  void *stream);

/* ---- shared MLP (1x1 conv + BatchNorm(train) + ReLU [+ pool over nsample]) on fp32 MFMA ----
 * The reference runs nn.Conv2d(1x1) -> nn.BatchNorm2d -> F.relu as three framework calls per layer
 * (classification/modules/repsurface_utils.py:236-244, 296-305).  Here a layer is one row-GEMM whose
 * operand is assembled on the fly (previous BatchNorm affine + ReLU in forward, BatchNorm-backward
 * affine in backward) and whose epilogue produces the BatchNorm sums; see repsurf_amd/csrc/mlp.hip.
 * All matrices are row-major (rows, channels) with explicit leading dimensions. */

/* E[r][c] built while loading (a, b row-major with leading dimensions lda, ldb; s*, t* per column): */
enum {
  RS_OP_ID = 0,     /* E = a[r][c]                                                                  */
  RS_OP_RELU1 = 1,  /* E = relu(s1*a + t1)                       BatchNorm + ReLU of a conv output   */
  RS_OP_RELU2 = 2,  /* E = relu(s1*a + t1 + s2*b + t2)           bn_l0(mlp_l0) + bn_f0(mlp_f0), :236-239 */
  RS_OP_AFF2 = 3,   /* E = s1*a + s2*b + t1                      BatchNorm backward: a = dz, b = y   */
  RS_OP_POOLED = 4, /* dz = arg[g][c]==r%ns ? a[g][c] : 0, g=r/ns; E = s1*dz + s2*b[r][c] + t1
                       (gradient through torch.max over nsample, :244; a/arg have leading dim lda)  */
  RS_OP_BCAST = 5   /* E = a[r/ns][c]                            gradient through a sum over ns, :305 */
};
typedef struct rs_row_operand {
  const float *a; long long lda;
  const float *b; long long ldb;
  const float *s1, *t1, *s2, *t2;
  const int *arg; int ns;
  int mode;
  /* Compacted groups (duplicate ball-query slots removed, see rs_group_features_compact):
   * mult[r] = number of identical copies row r stands for (NULL = 1): RS_OP_AFF2 / RS_OP_POOLED become
   * E = s1*dz + mult*(s2*b + t1) with dz the gradient already summed over the copies;
   * grp[r], slot[r] = group and position of row r (RS_OP_POOLED with ragged groups; NULL = r/ns, r%ns). */
  const float *mult; const int *grp; const int *slot;
  /* bf16 activation storage (BASELINE configs[4], rs_mlp_*_bf16 entry points only): a / b point to bf16 tensors (same
   * shapes and leading dimensions, in elements) instead of fp32 ones.  The fp32 entry points reject a non-zero flag. */
  int a_bf16, b_bf16;
} rs_row_operand;

enum {
  RS_EPI_STORE = 0, /* out = acc + bias                                                              */
  RS_EPI_STATS = 1, /* + per-column {sum y, sum y^2} -> partial[block][2][cols]      (BN forward)     */
  RS_EPI_MASK = 2   /* out = (ms1*my1+mt1 [+ ms2*my2+mt2] > 0) ? acc : 0;  partial[block][2|3][cols] =
                       {sum out, sum out*yhat1 [, sum out*yhat2]}, yhat = (my - mean)*invstd (BN backward) */
};
typedef struct rs_mlp_epilogue {
  const float *bias; float *out; long long ldo; int mode;
  const float *my1; long long ldm1; const float *ms1, *mt1, *mean1, *invstd1;
  const float *my2; long long ldm2; const float *ms2, *mt2, *mean2, *invstd2;
  double *partial; int partial_blocks;   /* rows of the partial buffer (unused ones are zeroed) */
  /* optional fused max-pool over groups of pool_ns consecutive rows (0 = off): raw extremes of `out`
   * per (group, column) and their positions, resolved by rs_pool_select once BatchNorm's scale is known.
   * pool_ns must divide 64 / 32 / 16 for cols > 64 / > 32 / <= 32. */
  int pool_ns; float *pool_max, *pool_min; int *pool_amax, *pool_amin;
  const float *row_mult;   /* RS_EPI_STATS: per-row weight of the sums (copies a compacted row stands for; NULL = 1) */
  /* bf16 activation storage (rs_mlp_gemm_rows_bf16 only): `out` / `my1` / `my2` are bf16 tensors.  With out_bf16 the result
   * is rounded to bf16 (nearest even) FIRST and the BatchNorm sums / fused pooling see the rounded values -- what the
   * consumers of the stored tensor will read (torch.autocast does the same: a bf16 conv output feeds an fp32 BatchNorm). */
  int out_bf16, my1_bf16, my2_bf16;
  /* Optional: the weights of THIS launch already split into three bf16 parts (rs_pack_weights: dst3), TILE-ORDERED (round 6):
   * w3[q][k / 8][n][k % 8], q = 0..2 (h, m, l), n < w3_part / ldw3 rows (>= cols), k < ldw3 (a multiple of 32, zero beyond kdim),
   * parts w3_part elements apart, base 16-byte aligned -- 16-byte units of 8 consecutive k, the units of one k-octet contiguous
   * over n, so that the split-product instances of the tiled kernel (rs_mlp_gemm_split3) move a (part, k-octet) plane of their
   * weight tile global -> LDS with one global_load_lds_dwordx4 per 64 columns instead of splitting the weight tile again in every
   * row tile of every launch.  NULL: the launch runs the fp32-MFMA instances (v_mfma_f32_32x32x2_f32) -- the split-product
   * instances exist only for pre-split weights.  A non-NULL image that breaks the rules above is an error (RS_ERR_ARG).  Other
   * kernels ignore it. */
  const void *w3; int ldw3; long long w3_part;
} rs_mlp_epilogue;

/* out[rows, cols] = E[rows, kdim] . W^T,  W[n][k] = w[n*ldw + k]: weights are passed n-major, i.e. a conv weight
 * (cout, cin) as it is stored for the forward pass (in place when cin % 4 == 0) and its transpose for the data
 * gradient dY . W (rs_pack_weights makes the padded / transposed copies); base 16-byte aligned, ldw % 4 == 0,
 * entries k in [kdim, ldw) zero. */
/* rows_dev (optional, device int): actual row count of a compacted operand — read by the kernel, never by the
 * host; `rows` is then the capacity of the buffers. */
int rs_mlp_gemm_rows(long long rows, const int *rows_dev, int kdim, int cols, const rs_row_operand *x,
                     const float *w, int ldw, const rs_mlp_epilogue *epi, void *stream);
/* How rs_mlp_gemm_rows / rs_mlp_wgrad form their fp32 products on the tiled MFMA kernels (round 4).  1 (default; environment
 * RS_GEMM_SPLIT3=0 selects 0, read once per process): every operand value -- after its fp32 prologue -- is committed to LDS as
 * three bf16 parts h + m + l (each the nearest-even bf16 of what the parts before it left: 24 significant bits together) and a
 * product is six v_mfma_f32_32x32x16_bf16 (hh, hm, mh, hl, lh, mm) accumulated in fp32: 192 matrix-pipe cycles per 16 k against
 * the 512 of eight v_mfma_f32_32x32x2_f32, error against an fp64 product equal to the fp32 MFMA's (profiles/r04/gemm_split3_ab.txt;
 * the two-part / three-product form misses the 1e-5 bound: tools/probes/bf16_split_accuracy.py).  0: v_mfma_f32_32x32x2_f32.
 * Tensors, prologues, epilogues, accumulation and BatchNorm sums are fp32 either way; the row GEMM uses it for every launch of the
 * tiled kernel with vector operands, the weight gradient for every product of its tiled kernel (RS_WGRAD_SPLIT3_WIDE=0: only those of up to
 * 64 columns of Q).
 * Non-finite operands: a NaN poisons its output row in both forms; an infinity (or a finite |x| >= 2^127 (2 - 2^-8), whose bf16
 * rounds to infinity) gives +-inf under the fp32 MFMA and NaN under the split (the residual of an infinity is inf - inf) -- the
 * row is non-finite either way and no other row is touched.  Denormal operands contribute as zeros would (below 1e-30).
 * tests/test_mlp_gpu.py::test_gemm_products_with_non_finite_and_denormal_operands. */
int rs_mlp_gemm_split3(void);

/* Mixed precision (BASELINE configs[4]: "bf16 mixed precision on CDNA4 MFMA for shared MLPs"): the same contract
 * and the same fp32 tensors in HBM; the operand E (after its fp32 prologue) and the weights are rounded to bf16
 * (nearest even) when they are staged in LDS and multiplied by v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
 * Bias, BatchNorm sums, masks and pooling stay fp32.  Where the layout forces scalar operand loads, or the narrow
 * streaming kernel applies (kdim <= 16 on an unaligned operand), the fp32 instance runs.  What torch.autocast would
 * do for the reference's nn.Conv2d 1x1 (classification/modules/repsurface_utils.py:236-244), minus the bf16
 * rounding of the conv OUTPUT. */
int rs_mlp_gemm_rows_bf16(long long rows, const int *rows_dev, int kdim, int cols, const rs_row_operand *x,
                          const float *w, int ldw, const rs_mlp_epilogue *epi, void *stream);

/* Weight gradient dw[ncols][kcols] = sum_r P[r][n] * Q[r][k]; rows are split into `chunks` workgroup
 * slabs whose partial products land in partial (chunks, ncols*kcols) and are summed in a fixed order.
 * dw = NULL: only the partials are produced; the caller sums them (rs_reduce_partials or
 * rs_bn_backward_finalize_reduce). */
int rs_mlp_wgrad(long long rows, const int *rows_dev, int ncols, int kcols, const rs_row_operand *p,
                 const rs_row_operand *q, float *partial, int chunks, float *dw, void *stream);
/* Mixed precision (see rs_mlp_gemm_rows_bf16): P and Q rounded to bf16 after their fp32 prologues, bf16 MFMA with
 * fp32 accumulation inside a row slab; partial products and their fixed-order sum stay fp32. */
int rs_mlp_wgrad_bf16(long long rows, const int *rows_dev, int ncols, int kcols, const rs_row_operand *p,
                      const rs_row_operand *q, float *partial, int chunks, float *dw, void *stream);

/* BatchNorm statistics -> affine: from partial (nblk, 2, c) {sum, sumsq} over `rows` rows:
 * mean, biased var, scale = gamma/sqrt(var+eps), shift = beta - mean*scale; saves mean/invstd and,
 * when running_mean != NULL, updates running stats with `momentum` and the unbiased variance
 * (nn.BatchNorm2d training semantics). */
int rs_bn_finalize(int c, long long rows, int nblk, const double *partial, const float *gamma,
                   const float *beta, float eps, float momentum, float *scale, float *shift,
                   float *save_mean, float *save_invstd, float *running_mean, float *running_var,
                   void *stream);

/* BatchNorm backward sums -> coefficients of dy = p*dz + q*y + r (RS_OP_AFF2 / RS_OP_POOLED operands)
 * from partial (nblk, nstat, c): row 0 = sum dz, row `which` = sum dz*yhat.  dgamma/dbeta optional. */
int rs_bn_backward_finalize(int c, long long rows, int nblk, int nstat, int which, const double *partial,
                            const float *scale, const float *mean, const float *invstd, float *p,
                            float *q, float *r, float *dgamma, float *dbeta, void *stream);

/* Several of the small jobs above in one launch (each is microseconds of work behind a graph node's ~5 us latency, and a
 * stack issues them in pairs): rs_bn_finalize for up to RS_BN_BATCH_MAX layers (the two BatchNorms of a two-branch first
 * layer: classification/modules/repsurface_utils.py:236-241); rs_bn_backward_finalize for up to RS_TAIL_FIN_MAX layers
 * together with rs_reduce_partials for up to RS_TAIL_RED_MAX weight gradients. */
#define RS_BN_BATCH_MAX 4
#define RS_TAIL_FIN_MAX 2
#define RS_TAIL_RED_MAX 4
/* rows_dev (round 6, optional): the row count as DEVICE data -- min(rows, *rows_dev) rows were summed.  A packed segmentation batch
 * changes its row counts every step (segmentation/util/data_util.py:15-23) while a captured hipGraph freezes every scalar argument:
 * launches are sized for a capacity (`rows`) and read the batch's counts from a small device table the host refills before each
 * replay (repsurf_amd.graph.RaggedSegStep).  Same convention as rs_mlp_gemm_rows' rows_dev. */
typedef struct {
  int c, nblk; long long rows; const double *partial; const float *gamma, *beta; float eps, momentum;
  float *scale, *shift, *save_mean, *save_invstd, *running_mean, *running_var;
  const int *rows_dev;
} rs_bn_item;                                         /* the arguments of rs_bn_finalize (+ rows_dev) */
typedef struct {
  int c, nblk, nstat, which; long long rows; const double *partial; const float *scale, *mean, *invstd;
  float *p, *q, *r, *dgamma, *dbeta;
  const int *rows_dev;
} rs_bn_bwd_item;                                     /* the arguments of rs_bn_backward_finalize (+ rows_dev) */
typedef struct { int chunks; long long n; const float *partial; float *out; } rs_reduce_item;   /* ... of rs_reduce_partials */
typedef struct { int nfin, nred; rs_bn_bwd_item fin[RS_TAIL_FIN_MAX]; rs_reduce_item red[RS_TAIL_RED_MAX]; } rs_backward_tail_work;
int rs_bn_finalize_batch(const rs_bn_item *items, int n, void *stream);
int rs_backward_tail(const rs_backward_tail_work *work, void *stream);

/* out[g][c] = max_k f(scale*y[g*nsample+k][c] + shift), f = relu when `relu` != 0, arg = first k
 * attaining it (torch.max(new_feature, 2)[0] fused with the last BatchNorm + ReLU, :243-244);
 * scale/shift may be NULL (identity). */
/* offsets (optional, (groups+1) int32): ragged groups of a compacted row set (rows [offsets[g], offsets[g+1]));
 * NULL = dense groups of nsample rows. */
/* y_bf16 != 0: y is a bf16 tensor (bf16 activation storage, written by rs_mlp_gemm_rows_bf16 with out_bf16). */
/* nsample = 1 (ungrouped rows: the Linear-BN-ReLU rows of the segmentation decoder): arg may be NULL -- nothing is selected,
 * the kernel is the BatchNorm + ReLU pass and rs_pool_max_backward takes arg = NULL as all zeros. */
int rs_pool_max(long long groups, int nsample, int c, int relu, const int *offsets, const float *y, int y_bf16,
                const float *scale, const float *shift, float *out, int *arg, void *stream);
/* Resolves the fused pooling of rs_mlp_gemm_rows: out = relu(scale * (scale >= 0 ? ymax : ymin) + shift),
 * arg = the matching position. */
int rs_pool_select(long long groups, int c, const float *ymax, const float *ymin, const int *amax,
                   const int *amin, const float *scale, const float *shift, float *out, int *arg, void *stream);
/* v = dout * (out > 0) and the BatchNorm-backward sums of the pooled layer from (groups, c) data only:
 * partial (partial_blocks, 2, c) = {sum v, sum v * yhat[arg row]}.  out = NULL: the pooled layer ended without a
 * ReLU (rs_pool_max called with relu = 0), v = dout (v may then be NULL: only the sums).  dout: (groups, c) rows `ldd` floats apart (0 = c): the pooled
 * activations' gradient is often a column slice of a wider tensor.
 * groups_dev (optional): the group count as device data, min(groups, *groups_dev) groups are processed (see rs_bn_item). */
int rs_pool_max_backward(long long groups, int nsample, int c, const int *offsets, const float *dout, long long ldd,
                         const float *out, const int *arg, const float *y, int y_bf16, const float *mean,
                         const float *invstd, float *v, double *partial, int partial_blocks, const int *groups_dev, void *stream);
/* out[g][c] = sum_k y[g*nsample+k][c]   (umbrella aggregation 'sum', :305) */
int rs_pool_sum(long long groups, int nsample, int c, const float *y, float *out, void *stream);

/* Zero-padded copies of up to RS_PACK_MAX conv weights (cout, cin) in one launch, in the n-major layout
 * rs_mlp_gemm_rows reads:  transpose[e] = 0: dst[j*ld + k] = src[j*cin + k] (0 for cin <= k < ld) -- forward operand
 * when cin % 4 != 0;  transpose[e] = 1: dst[k*ld + j] = src[j*cin + k] (0 for cout <= j < ld) -- operand of dY . W. */
#define RS_PACK_MAX 32
typedef struct rs_pack_weights_args {
  const float *src[RS_PACK_MAX]; float *dst[RS_PACK_MAX];
  int cout[RS_PACK_MAX], cin[RS_PACK_MAX], ld[RS_PACK_MAX], transpose[RS_PACK_MAX];
  int n;
  /* dst3[e] (optional) receives the same n-major matrix as three bf16 parts h + m + l (each the nearest-even bf16 of what the
   * parts before it left), tile-ordered (round 6): dst3[e][q][k / 8][row][k % 8], k < ld3[e] (a multiple of 32 >= the inner
   * dimension, zero filled), row < outer dimension, parts (outer dimension) * ld3[e] elements apart, 16-byte aligned -- the `w3`
   * operand of rs_mlp_epilogue.  dst[e] may then be NULL (a forward weight that is used in place needs no fp32 copy). */
  void *dst3[RS_PACK_MAX]; int ld3[RS_PACK_MAX];
} rs_pack_weights_args;
int rs_pack_weights(const rs_pack_weights_args *args, void *stream);

/* out[e] = sum_b partial[b][e], b ascending (deterministic reduction of fp32 workgroup partials). */
int rs_reduce_partials(int nblk, long long n, const float *partial, float *out, void *stream);
/* rs_bn_backward_finalize and rs_reduce_partials(red_chunks, red_n, red_partial, red_out) in ONE launch: in the backward
 * chain the weight-gradient reduction of a layer (rs_mlp_wgrad called with dw = NULL leaves it to the caller) and the
 * BatchNorm-backward finalize of the layer below follow each other, each a few microseconds of work behind a
 * graph node's launch latency.  Same results as the two separate calls (same summation order). */
int rs_bn_backward_finalize_reduce(int c, long long rows, int nblk, int nstat, int which, const double *partial,
                                   const float *scale, const float *mean, const float *invstd, float *p, float *q,
                                   float *r, float *dgamma, float *dbeta, int red_chunks, long long red_n,
                                   const float *red_partial, float *red_out, void *stream);

/* ---- classifier head on <= 64 rows (repsurf_amd/csrc/head.hip) -----------------------------------------------
 * classfier = Linear-BN1d-ReLU-Dropout-Linear-BN1d-ReLU-Dropout-Linear + log_softmax
 * (classification/models/repsurf/repsurf_ssg_umb.py:32-41,56-57) and SmoothClsLoss
 * (classification/util/utils.py:55-69).  A workgroup owns 4 output columns of a layer for all rows, so the
 * BatchNorm(train) statistics are local: one launch per hidden layer forward (Linear + BN + ReLU + Dropout) and one
 * per hidden layer backward (data gradient from the next layer + Dropout/ReLU/BN backward + weight gradient).
 * Dropout masks are a hash of (seed, *step, layer, element); rs_head_output_forward advances *step once per forward,
 * backward kernels recompute the mask of `*step - step_back`. */
typedef struct rs_head_layer {
  const float *x; int ldx;                    /* input (R, K) */
  const float *w, *b, *gamma, *beta;          /* (N, K), (N), (N), (N) */
  float *running_mean, *running_var;          /* updated in place; NULL = not tracked */
  float momentum, eps, drop_p;
  float *y, *h, *mean, *invstd;               /* outputs: (R, N) pre-BN, (R, N) activation, (N), (N) */
  unsigned seed; const int *step; int layer;
  int R, K, N;
} rs_head_layer;
int rs_head_layer_forward(const rs_head_layer *l, void *stream);
/* logp (R, classes) = log_softmax(h . w^T + b); *step += 1 when step is not NULL */
int rs_head_output_forward(int rows, int k, int classes, const float *h, const float *w, const float *b,
                           float *logp, int *step, void *stream);
/* dlogits = dlogp - exp(logp) * rowsum(dlogp); dw (classes, k) = dlogits^T h; db = colsum(dlogits) */
int rs_head_output_backward(int rows, int k, int classes, const float *dlogp, const float *logp, const float *h,
                            float *dlogits, float *dw, float *db, void *stream);
typedef struct rs_head_layer_bwd {
  const float *dz_next; int n2; const float *w_next;      /* (R, N2) gradient at the next Linear's output, its weight (N2, N) */
  const float *y, *mean, *invstd, *gamma, *beta;          /* saved by the forward */
  const float *x; int ldx;                                /* this layer's input (R, K) */
  float *dz, *dw, *dgamma, *dbeta;                        /* outputs: (R, N), (N, K), (N), (N) */
  float drop_p; unsigned seed; const int *step; int layer; int step_back;
  int R, K, N;
} rs_head_layer_bwd;
int rs_head_layer_backward(const rs_head_layer_bwd *l, void *stream);
/* dx (R, K) = dz (R, N) . w (N, K): the gradient that leaves the head */
int rs_head_input_backward(int rows, int n, int k, const float *dz, const float *w, float *dx, void *stream);
/* loss[0] = -mean_r sum_j soft[r][j] logp[r][j], soft = 1-eps on target[r] and eps/(classes-1) elsewhere;
 * dlogp (R, classes) = -soft / R (d loss / d logp). */
int rs_smooth_cls_loss(int rows, int classes, float eps, const float *logp, const long long *target,
                       float *loss, float *dlogp, void *stream);
/* nn.CrossEntropyLoss(ignore_index) of the segmentation train loop (segmentation/tool/train.py:110,296) on logits (rows, classes):
 * loss[0] = mean over rows with target != ignore_index of logsumexp(row) - row[target]; inv_count[0] = 1 / (number of such rows);
 * dlogits (rows, classes) = softmax(row) - onehot(target) (zero rows where ignored): d loss / d logits = dlogits * inv_count[0].
 * partial: 2 * ceil(rows / 256) doubles of scratch.  A label outside [0, classes) that is not ignore_index makes the row's loss and
 * gradient NaN (torch traps it with a device assert) and adds 1 to *bad_labels (device int, never reset by the library; NULL:
 * not counted) -- what a training loop polls, since a NaN inside a replayed graph raises nothing on the host. */
int rs_cross_entropy_forward(long long rows, int classes, long long ignore_index, const float *logits, const long long *target,
                             float *loss, float *inv_count, float *dlogits, double *partial, int *bad_labels, void *stream);
/* out[i] = x[i] * a[0] * (b ? b[0] : 1): a, b device scalars (the loss gradient times 1 / count and the incoming gradient) */
int rs_scale_by_scalars(long long n, const float *x, const float *a, const float *b, float *out, void *stream);
/* Column sums, stage 1: partial (nblk, n), row slab b of x (rows, n; rows ldx floats apart) summed per column and multiplied by
 * `scale`; rs_reduce_partials (nblk, n) finishes in a fixed order: dout.sum(0), the bias gradient autograd forms for the
 * classifier's output nn.Linear (segmentation/models/repsurf/repsurf_umb_ssg.py:36-41) and the constructor's last Conv1d
 * (segmentation/modules/repsurface_utils.py:298-303). */
int rs_col_sum_partials(long long rows, int n, const float *x, long long ldx, float scale, float *partial, int nblk, void *stream);
/* the same with the row count as device data (see rs_bn_item): rows [min(rows, *rows_dev), rows) are not read */
int rs_col_sum_partials_dev(long long rows, int n, const float *x, long long ldx, float scale, float *partial, int nblk,
                            const int *rows_dev, void *stream);

/* ---- optimizer step -------------------------------------------------------------------------------
 * torch.optim.Adam(lr, betas, eps, weight_decay) as the reference configures it
 * (classification/tool/train_cls_scanobjectnn.py:179-185), for up to RS_ADAM_MAX fp32 tensors per launch:
 *   g += wd * p;  m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
 * hyper (device, 10 DOUBLES): [0..4] = {lr, beta1, beta2, eps, weight_decay}, written by the caller; [5..9] belong to the
 * kernel (zero them once): the bias corrections of the NEXT step, left by a step's last workgroup so that the next launch
 * does not recompute two fp64 powers in every thread.  step (device int) = number of updates done so far,
 * t = *step + 1; with advance != 0 the launch stores t back when its last workgroup retires (`done` = device int, 0
 * between launches) -- pass advance = 0 on all but the last launch of one optimizer step. */
#define RS_ADAM_MAX 80      /* the table travels in the kernel arguments: 80 x 44 bytes, under the 4 KB limit */
typedef struct rs_adam_table {
  float *p[RS_ADAM_MAX];         /* parameters, updated in place */
  const float *g[RS_ADAM_MAX];   /* gradients */
  float *m[RS_ADAM_MAX];         /* exp_avg */
  float *v[RS_ADAM_MAX];         /* exp_avg_sq */
  int n[RS_ADAM_MAX];            /* elements */
  int count;
} rs_adam_table;
int rs_adam_step(const rs_adam_table *t, double *hyper, int *step, int *done, int advance, void *stream);

/* Measurement aid (round 6): one-thread launch that stores the device's constant-rate wall clock (s_memrealtime) in *dst when it runs;
 * rs_timestamp_khz() = ticks per millisecond.  bench.py brackets a launch inside the step's replayed hipGraph with two of them (HIP
 * events cannot be recorded inside a replayed graph on this runtime).  Not used by the product path. */
int rs_timestamp(long long *dst, void *stream);
int rs_timestamp_khz(void);

/* ---- fused 10-channel MLP of UmbrellaSurfaceConstructor ---------------------------------------
 * self.mlps = Conv2d(10,10,bias=False)-BN-ReLU-Conv2d(10,10)-BN-ReLU-Conv2d(10,10) + sum/avg over the fan
 * (classification/modules/repsurface_utils.py:266-274, 296-305) as six register-resident passes over the
 * (rows, 10) geometric features; nothing but BatchNorm vectors is stored between passes
 * (repsurf_amd/csrc/umbrella_mlp.hip).  Each pass is one launch of `nblk` workgroups:
 *   0: stat_partial (nblk,2,10) = {sum y0, sum y0^2}                      -> rs_bn_finalize -> bn0
 *   1: same for y1 (needs bn0)                                             -> rs_bn_finalize -> bn1
 *   2: out (rows/group, 10) = out_scale * sum over the group of y2        (needs bn0, bn1)
 *   3: dw_partial (nblk,110) = {dW2 (10x10), db2 (10)}, stat_partial = {sum dz1, sum dz1*yhat1}
 *   4: dw_partial = {dW1, 0}, stat_partial = {sum dz0, sum dz0*yhat0}     (needs c1 = p,q,r of BN1 backward)
 *   5: dw_partial = {dW0, 0}                                               (needs c0)
 * bn0/bn1: (4,10) = scale, shift, mean, invstd as written by rs_bn_finalize; c0/c1: (>=3,10) = p,q,r as
 * written by rs_bn_backward_finalize.  dout: (rows/group, 10), already scaled for 'avg'.
 * layers = 2: the segmentation constructor's Conv1d(10,10)-BN-ReLU-Conv1d(10,10) + sum over the fan
 * (segmentation/modules/repsurface_utils.py:298-303,323-327), y0 = W0 x + b0: passes 0 (stats of y0), 2 (out = sum of y1),
 * 4 (dw_partial = {dW1, db1}, stat_partial = {sum dz0, sum dz0*yhat0}), 5 (dw_partial = {dW0, 0}; needs c0). */
typedef struct rs_umbrella_mlp {
  const float *x; long long rows; int group;
  const float *w0, *w1, *b1, *w2, *b2;
  const float *bn0, *bn1;
  const float *c0, *c1;
  const float *dout;
  const float *b0;          /* bias of the first conv (two-layer variant; NULL = none) */
  int layers;               /* 3 (0 = 3) or 2 */
} rs_umbrella_mlp;
int rs_umbrella_mlp_pass(int pass, const rs_umbrella_mlp *m, float out_scale, float *out,
                         double *stat_partial, float *dw_partial, int nblk, void *stream);

/* ---- the same MLP on the fp32 matrix pipe (repsurf_amd/csrc/umbrella_mfma.hip, round 4) ---------------------
 * A wave owns 16 points at a time and walks their `group` fan rows as 16-row v_mfma_f32_16x16x4_f32 tiles; weights and
 * BatchNorm vectors live in registers, weight gradients are MFMA accumulators (rows = the reduction index).  The passes whose
 * sums are linear in the input are replaced by the MOMENTS of x: rs_umbrella_moments writes moments (11, 16) fp64,
 * S[m][n] = sum_rows x_m x_n with index 10 = the constant 1 (S[10][k] = sum x_k, S[10][10] = rows) -- a function of the
 * geometry alone (the caller computes it in its geometry stage); `partial` (nblk, RS_UMB_MOM_ROW) floats is scratch.
 * BatchNorm 0 (of y0 = W0 x + b0) follows from the moments; so does the part of dW0 that is not sum dz0 x^T.
 * rs_umbrella_mfma_pass(pass, m, nblk, stream), one launch of nblk workgroups each (RS_UMB_FIN: 10 workgroups):
 *   three layers (classification/modules/repsurface_utils.py:266-274,296-305):
 *     RS_UMB_F1  stat (nblk, 2, 16) fp64 = {sum y1, sum y1^2}; publishes bn0 (+ running statistics 0)       [nblk_f1 = nblk]
 *     RS_UMB_F2  out (rows / group, 10) = out_scale * sum over the fan of y2; publishes bn1 (+ running statistics 1)
 *     RS_UMB_B1  part_b1 (nblk, RS_UMB_B1_ROW): dW2, db2, BatchNorm-1 backward sums, the sums dW1 is built from [nblk_b1 = nblk]
 *     RS_UMB_B2  part_b2 (nblk, RS_UMB_B2_ROW): sum dz0 x^T, BatchNorm-0 backward sums                        [nblk_b2 = nblk]
 *     RS_UMB_FIN grads (360): dW0 [0,100) dgamma0 [100,110) dbeta0 [110,120) dW1 [120,220) dbias1 [220,230) (two layers only)
 *                dgamma1 [230,240) dbeta1 [240,250) dW2 [250,350) dbias2 [350,360) (three layers only)
 *   two layers (segmentation/modules/repsurface_utils.py:298-303,323-327; y0 = W0 x + b0, the output is the sum of y1):
 *     RS_UMB_F2 (publishes bn0), RS_UMB_B2 (also dW1 = sum dy1^T a0, db1), RS_UMB_FIN.
 * bn0 / bn1: (4, 10) = scale, shift, mean, invstd (the layout rs_bn_finalize writes); dout (rows / group, 10), already
 * scaled for 'avg'.  Sums: fp32 inside a workgroup, fp64 across workgroups, fixed order (deterministic). */
#define RS_UMB_C 10
#define RS_UMB_MOM_ROW 176
#define RS_UMB_B1_ROW 544
#define RS_UMB_B2_ROW 368
#define RS_UMB_GRADS 360
enum { RS_UMB_F1 = 1, RS_UMB_F2 = 2, RS_UMB_B1 = 3, RS_UMB_B2 = 4, RS_UMB_FIN = 5 };
typedef struct rs_umbrella_mfma {
  const float *x; long long rows; int group; int layers;
  const float *w0, *b0, *w1, *b1, *w2, *b2;
  const float *gamma0, *beta0, *gamma1, *beta1;
  float eps0, eps1, mom0, mom1;
  float *bn0, *bn1;
  float *run_mean0, *run_var0, *run_mean1, *run_var1;      /* NULL: no running statistics */
  const double *moments;
  const float *dout;
  double *stat; int nblk_f1;
  float *part_b1; int nblk_b1;
  float *part_b2; int nblk_b2;
  float out_scale; float *out;
  float *grads;
  const int *rows_dev;      /* optional: the row count as device data, min(rows, *rows_dev) rows are walked (rs_bn_item); a multiple of group */
} rs_umbrella_mfma;
int rs_umbrella_moments(const float *x, long long rows, float *partial, int nblk, double *moments, void *stream);
int rs_umbrella_mfma_pass(int pass, const rs_umbrella_mfma *m, int nblk, void *stream);

/* ---- sectorized FPS on the device: pointops.sectorized_fps
 * (segmentation/modules/pointops/functions/pointops.py:52-108) without its host loop and read-backs.
 * rs_sectorize: per cloud, angle = atan2(x, y), S + 1 linspace boundaries over [min, max + 1e-4], stable partition of the
 * rows into S angular sectors (S = sec_base[i+1] - sec_base[i] = 1 for clouds below min_points, num_sectors otherwise;
 * sec_base (b + 1) is the running sector count, the host knows the cloud sizes).  Outputs: indices (n_tot) = original row
 * of every sector row, sector_xyz (n_tot, 3), sector_offset / new_sector_offset (sec_base[b]) running ends (picks per
 * sector = new_size / S, remainder in the last, :82-84), *n_max_dev = largest sector (zero it before the call).
 * rs_furthestsampling_sectors: rs_furthestsampling_offset over those sectors; the tie rule's block size comes from
 * *n_max_dev, n_bound >= every sector (the largest cloud).  rs_take_int: out[i] = table[idx[i]] (:105). */
int rs_sectorize(int b, const float *xyz, const int *offset, const int *new_offset, const int *sec_base, int num_sectors,
                 int min_points, int *indices, float *sector_xyz, int *sector_offset, int *new_sector_offset,
                 int *n_max_dev, void *stream);
int rs_furthestsampling_sectors(int b, int n_bound, const int *n_max_dev, const float *xyz, const int *offset,
                                const int *new_offset, float *temp, int *idx, void *stream);
int rs_take_int(int n, const int *table, const int *idx, int *out, void *stream);

/* ---- whole-scene kNN through a uniform grid (repsurf_amd/csrc/scene_knn.hip): pointops.knnquery for ONE large cloud, the
 * search of segmentation/util/utils.py:235-245 (pc_median_filter_gpu) / tool/test_s3dis.py:203-232 at N ~ 1e5..1e6.
 * lo / hi (3 HOST floats each): the rows' bounding box, cell: cell edge, g (3 HOST ints): cells per axis (<= 2^26 cells).
 * rs_scene_cells: cell index per row + histogram (counts: g0*g1*g2 + 1 ints, zeroed by the caller); the caller turns the
 * histogram into first positions (rs_exclusive_scan).  rs_scene_scatter: rows into cell-sorted float4 (x, y, z, row bits);
 * `cursor` = a copy of the first positions, advanced in place.  rs_scene_knn: per query the nsample (<= 32) nearest rows of
 * its 27 cells, ascending by (squared distance, row) -- same arithmetic and tie rule as rs_knnquery_offset; flag[q] = 0: the
 * list is complete (nsample-th distance below the cell edge) and idx / dist2 row q is written; 1: run rs_knnquery_offset for
 * this query. */
int rs_scene_cells(int n, const float *xyz, const float *lo, const float *hi, float cell, const int *g, int *cell_of,
                   int *counts, void *stream);
int rs_scene_scatter(int n, const float *xyz, const int *cell_of, int *cursor, float *sorted, void *stream);
int rs_scene_knn(int m, int nsample, const float *queries, const float *lo, const float *hi, float cell, const int *g,
                 const int *starts, const float *sorted, int *idx, float *dist2, int *flag, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* REPSURF_HIP_H */
