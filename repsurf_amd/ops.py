"""Torch-facing operators over the C ABI of librepsurf_hip.so.

PyTorch is plumbing here: it owns device memory, the current HIP stream and autograd graph
bookkeeping; every computation below is a hand-written HIP kernel reached through
repsurf_amd._lib.call().  All tensors must live on a HIP device — there is no CPU path.

Layouts are channels-last (see include/repsurf_hip.h): xyz (B,N,3), features (B,N,C),
indices int32.
"""
import os

import torch
from torch.autograd import Function

from . import _lib
from . import ragged as _ragged


def _stream():
    return _lib.current_stream()


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.RepSurfHipError(
                "repsurf_amd operators run on the GPU only (got a CPU tensor); there is no CPU fallback")


def _f32c(t):
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _i32c(t):
    if t.dtype != torch.int32:
        t = t.to(torch.int32)
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return None if t is None else t.data_ptr()


# ----------------------------------------------------------------------------- sampling
def furthestsampling(xyz, m, start=None):
    """xyz (B,N,3) -> idx (B,m) int32.  start: (B,) first picks (None = index 0).
    Semantics: farthest_point_sample(cuda=False), classification/modules/pointnet2_utils.py:47-75."""
    _need_gpu(xyz, start)
    xyz = _f32c(xyz)
    b, n, _ = xyz.shape
    idx = torch.empty((b, m), dtype=torch.int32, device=xyz.device)
    st = None if start is None else _i32c(start)
    temp = None
    if n > 16384:
        temp = torch.empty((b, n), dtype=torch.float32, device=xyz.device)
    _lib.call("rs_furthestsampling", b, n, m, _p(xyz), _p(st), _p(temp), _p(idx), _stream())
    return idx


# ----------------------------------------------------------------------------- packed batches (segmentation)
# Clouds concatenated along rows; `offset` (B,) int32 holds the running row ends (segmentation collate
# format).  The reference reads offsets back with per-cloud `.item()` calls in every stage
# (segmentation/modules/repsurface_utils.py:17-22, pointops.py:40-42); here the host copy is taken once per
# offset tensor and travels with it (`_rs_host`), and derived offsets are cached by value, so a model forward
# costs at most one device->host read and is graph-capturable once the caches are warm.
_OFFSET_CACHE = {}


def host_offsets(offset):
    """Running ends of a packed batch as a tuple of ints (one device->host read per tensor object and content:
    the copy is keyed on the tensor's storage address and autograd version, so an offset tensor refilled in place --
    a reused collate buffer -- is read again; a buffer rewritten through a raw pointer or a graph replay must be a
    fresh tensor object, its host values cannot be known without a read)."""
    tag = (offset.data_ptr(), offset._version)
    cached = getattr(offset, "_rs_host", None)
    if cached is not None and cached[0] == tag:
        return cached[1]
    host = tuple(int(v) for v in offset.tolist())
    try:
        offset._rs_host = (tag, host)
    except AttributeError:
        pass
    return host


def offsets_tensor(host, device):
    """Device int32 tensor of the running ends `host` (cached by value)."""
    key = (tuple(host), str(device))
    t = _OFFSET_CACHE.get(key)
    if t is None:
        if len(_OFFSET_CACHE) > 4096:
            _OFFSET_CACHE.clear()
        t = torch.tensor(list(host), dtype=torch.int32, device=device)
        t._rs_host = ((t.data_ptr(), t._version), tuple(host))
        _OFFSET_CACHE[key] = t
    return t


def strided_offset(offset, stride):
    """new_offset of the segmentation sample_and_group (repsurface_utils.py:17-22): running sum of
    (cloud length // stride) -- NOT offset // stride."""
    host = host_offsets(offset)
    out, acc, last = [], 0, 0
    for end in host:
        acc += (end - last) // stride
        out.append(acc)
        last = end
    return offsets_tensor(out, offset.device)


def furthestsampling_offset(xyz, offset, new_offset):
    """Packed batches: xyz (Ntot,3), offset/new_offset (B,) int32 running ends -> idx (Mtot,) int32 global rows;
    first pick of each cloud = its first row (segmentation/.../sampling_cuda_kernel.cu:39)."""
    _need_gpu(xyz, offset, new_offset)
    xyz, offset, new_offset = _f32c(xyz), _i32c(offset), _i32c(new_offset)
    host, new_host = host_offsets(offset), host_offsets(new_offset)
    b = len(host)
    n_max = max((e - s for s, e in zip((0,) + host[:-1], host)), default=0)
    m_tot = new_host[-1] if b else 0
    idx = torch.empty((m_tot,), dtype=torch.int32, device=xyz.device)
    temp = torch.empty((xyz.shape[0],), dtype=torch.float32, device=xyz.device) if n_max > 16384 else None
    _lib.call("rs_furthestsampling_offset", b, n_max, _p(xyz), _p(offset), _p(new_offset), _p(temp), _p(idx), _stream())
    return idx


def sectorized_fps(xyz, offset, new_offset, num_sectors, min_points=10000):
    """pointops.sectorized_fps (segmentation/modules/pointops/functions/pointops.py:52-108) on the device: clouds of at
    least `min_points` rows are cut into `num_sectors` sectors of atan2(x, y), each sector gets new_size // num_sectors
    picks (the last also the remainder) from an independent FPS.  Three launches (sectorize, FPS over the sectors,
    index remap) and NO host read-back: sector sizes stay on the device, output sizes follow from the offsets' host
    copies.  -> idx (Mtot,) int32 global rows."""
    _need_gpu(xyz, offset, new_offset)
    xyz, offset, new_offset = _f32c(xyz), _i32c(offset), _i32c(new_offset)
    host, new_host = host_offsets(offset), host_offsets(new_offset)
    sizes = [e - s for s, e in zip((0,) + host[:-1], host)]
    per = [1 if n < min_points else int(num_sectors) for n in sizes]
    if num_sectors <= 1 or all(p == 1 for p in per):
        # every cloud keeps a single sector, whose row list is the cloud itself in order: plain FPS
        return furthestsampling_offset(xyz, offset, new_offset)
    base, acc = [0], 0
    for p in per:
        acc += p
        base.append(acc)
    dev = xyz.device
    sec_base = offsets_tensor(base, dev)
    n_tot, s_tot, m_tot = xyz.shape[0], acc, (new_host[-1] if new_host else 0)
    indices = torch.empty((n_tot,), dtype=torch.int32, device=dev)
    sector_xyz = torch.empty((n_tot, 3), dtype=torch.float32, device=dev)
    ends = torch.empty((2, s_tot), dtype=torch.int32, device=dev)
    n_max_dev = torch.zeros((1,), dtype=torch.int32, device=dev)
    _lib.call("rs_sectorize", len(host), _p(xyz), _p(offset), _p(new_offset), _p(sec_base), int(num_sectors), int(min_points),
              _p(indices), _p(sector_xyz), _p(ends[0]), _p(ends[1]), _p(n_max_dev), _stream())
    n_bound = max(sizes)
    picks = torch.empty((m_tot,), dtype=torch.int32, device=dev)
    temp = torch.empty((n_tot,), dtype=torch.float32, device=dev) if n_bound > 14000 else None
    _lib.call("rs_furthestsampling_sectors", s_tot, n_bound, _p(n_max_dev), _p(sector_xyz), _p(ends[0]), _p(ends[1]), _p(temp),
              _p(picks), _stream())
    out = torch.empty_like(picks)
    _lib.call("rs_take_int", m_tot, _p(indices), _p(picks), _p(out), _stream())
    return out


SCENE_GRID_MIN_ROWS = 32768      # below this the tiled scan (rs_knnquery_offset) is as fast and needs no set-up


def knn_scene(nsample, xyz, new_xyz=None, return_stats=False):
    """k nearest rows inside ONE large cloud (whole-scene inference: segmentation/util/utils.py:235-245,
    tool/test_s3dis.py:203-232; pointops.knnquery with a single-entry offset): xyz (N,3), new_xyz (M,3) | None = xyz ->
    idx (M,nsample) int32, dist2 (M,nsample) -- the same lists, bit for bit, as `knnquery_offset`, through a uniform grid:
    rows are counting-sorted into cells of edge ~ (nsample / density)^(1/3), a query scans its 27 cells and its list is
    accepted when the nsample-th distance is below the cell edge (then no row outside the block can be closer); the other
    queries (and those outside the rows' bounding box) go through the exact tiled scan.  Reads the bounding box and the
    flagged queries back to the host (a scene-level helper, not a training-step operator)."""
    import ctypes
    _need_gpu(xyz, new_xyz)
    xyz = _f32c(xyz)
    q = xyz if new_xyz is None else _f32c(new_xyz)
    n, m, dev = xyz.shape[0], q.shape[0], xyz.device
    one = offsets_tensor([n], dev)
    if n < SCENE_GRID_MIN_ROWS or nsample > 32:
        idx, d2 = knnquery_offset(nsample, xyz, q, one, offsets_tensor([m], dev))
        return (idx, d2, {"grid": False}) if return_stats else (idx, d2)
    lo_t, hi_t = xyz.min(0).values, xyz.max(0).values
    lo, hi = [float(v) for v in lo_t.tolist()], [float(v) for v in hi_t.tolist()]
    ext = [max(h - l, 1e-6) for l, h in zip(lo, hi)]
    vol = ext[0] * ext[1] * ext[2]
    cell = (nsample * vol / n) ** (1.0 / 3.0)                   # ~nsample rows per cell at the mean density
    g = [int(e / cell) + 1 for e in ext]
    while g[0] * g[1] * g[2] > (1 << 24):                        # keep the cell table <= 64 MB
        cell *= 1.26
        g = [int(e / cell) + 1 for e in ext]
    ncell = g[0] * g[1] * g[2]
    f3, i3 = ctypes.c_float * 3, ctypes.c_int * 3
    clo, chi, cg = f3(*lo), f3(*hi), i3(*g)
    counts = torch.zeros((ncell + 1,), dtype=torch.int32, device=dev)
    cell_of = torch.empty((n,), dtype=torch.int32, device=dev)
    _lib.call("rs_scene_cells", n, _p(xyz), clo, chi, cell, cg, _p(cell_of), _p(counts), _stream())
    starts = torch.empty((ncell + 2,), dtype=torch.int32, device=dev)
    _lib.call("rs_exclusive_scan", ncell + 1, _p(counts), _p(starts), _stream())
    cursor = starts[:ncell + 1].clone()
    cells = torch.empty((n, 4), dtype=torch.float32, device=dev)
    _lib.call("rs_scene_scatter", n, _p(xyz), _p(cell_of), _p(cursor), _p(cells), _stream())
    idx = torch.empty((m, nsample), dtype=torch.int32, device=dev)
    d2 = torch.empty((m, nsample), dtype=torch.float32, device=dev)
    flag = torch.empty((m,), dtype=torch.int32, device=dev)
    _lib.call("rs_scene_knn", m, nsample, _p(q), clo, chi, cell, cg, _p(starts), _p(cells), _p(idx), _p(d2), _p(flag), _stream())
    bad = torch.nonzero(flag).squeeze(1)
    if bad.numel():
        bi, bd = knnquery_offset(nsample, xyz, q[bad].contiguous(), one, offsets_tensor([int(bad.numel())], dev))
        idx[bad] = bi
        d2[bad] = bd
    if return_stats:
        return idx, d2, {"grid": True, "cells": ncell, "cell_edge": cell, "rescanned": int(bad.numel())}
    return idx, d2


def umbrella_fan_offset(xyz, new_xyz, knn_idx, new_offset, inv_sign=None, rotate=True):
    """Segmentation umbrella fan: knn_idx (M,k) global rows of the k nearest neighbours (query included) ->
    (M, k, 10) = [polar, normal, const, centroid] per fan triangle
    (segmentation/modules/repsurface_utils.py:305-321; rotate = sort='fix')."""
    _need_gpu(xyz, new_xyz, knn_idx, new_offset, inv_sign)
    xyz, new_xyz, knn_idx, new_offset = _f32c(xyz), _f32c(new_xyz), _i32c(knn_idx), _i32c(new_offset)
    m, k = knn_idx.shape
    feat = torch.empty((m, k, 10), dtype=torch.float32, device=xyz.device)
    sg = None if inv_sign is None else _f32c(inv_sign.reshape(-1))
    _lib.call("rs_umbrella_fan_offset", m, k, new_offset.numel(), int(bool(rotate)), _p(xyz), _p(new_xyz),
              _p(knn_idx), _p(new_offset), _p(sg), _p(feat), _stream())
    return feat


def interp_weights(dist2):
    """dist2 (n,3) squared 3-NN distances -> normalised inverse-distance weights (n,3)
    (segmentation/modules/repsurface_utils.py:262-265)."""
    _need_gpu(dist2)
    dist2 = _f32c(dist2)
    w = torch.empty_like(dist2)
    _lib.call("rs_interp_weights", dist2.shape[0], _p(dist2), _p(w), _stream())
    return w


def _zero_pair(rows, c0, c1, lead, dev):
    """two zero-filled (lead..., c) scatter targets out of ONE allocation / ONE fill (None where c == 0)"""
    buf = torch.zeros((rows * (c0 + c1),), dtype=torch.float32, device=dev)
    a = buf[:rows * c0].view(*lead, c0) if c0 else None
    b = buf[rows * c0:].view(*lead, c1) if c1 else None
    return a, b


class _GatherRows(Function):
    @staticmethod
    def forward(ctx, points, idx):
        _need_gpu(points, idx)
        points, idx = _f32c(points), _i32c(idx)
        b, n, c = points.shape
        per = idx.numel() // b
        out = torch.empty(tuple(idx.shape) + (c,), dtype=torch.float32, device=points.device)
        _lib.call("rs_gather_rows", b, n, per, c, _p(points), _p(idx), _p(out), _stream())
        ctx.save_for_backward(idx)
        ctx.dims = (b, n, per, c)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        b, n, per, c = ctx.dims
        grad_out = _f32c(grad_out)
        grad = torch.zeros((b, n, c), dtype=torch.float32, device=grad_out.device)
        rows_dev = _ragged.dev(per) if b == 1 else None      # (a packed batch under a captured capacity: rows beyond the count are not scattered)
        if rows_dev is not None:
            _lib.call("rs_gather_rows_backward_dev", b, n, per, c, _p(grad_out), _p(idx), _p(grad), rows_dev, _stream())
        else:
            _lib.call("rs_gather_rows_backward", b, n, per, c, _p(grad_out), _p(idx), _p(grad), _stream())
        return grad, None


def gather_rows(points, idx):
    """points (B,N,C), idx (B,S) or (B,S,K) -> (B,S,C) / (B,S,K,C); differentiable w.r.t. points.
    index_points of classification/modules/pointnet2_utils.py:28-44 without the transposes."""
    return _GatherRows.apply(points, idx)


# ----------------------------------------------------------------------------- neighbour search
def ballquery(radius, nsample, xyz, new_xyz, return_count=False):
    """-> (B,S,nsample) int32 [, (B,S) distinct-neighbour counts]; query_ball_point(cuda=False) semantics
    (pointnet2_utils.py:78-99).  The threshold is float32(radius**2) with the square taken in double, as
    torch's tensor-vs-Python-scalar comparison does."""
    _need_gpu(xyz, new_xyz)
    xyz, new_xyz = _f32c(xyz), _f32c(new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = torch.empty((b, m, nsample), dtype=torch.int32, device=xyz.device)
    cnt = torch.empty((b, m), dtype=torch.int32, device=xyz.device) if return_count else None
    r2 = torch.tensor(float(radius) ** 2, dtype=torch.float32).item()
    _lib.call("rs_ballquery", b, n, m, r2, nsample, _p(new_xyz), _p(xyz), _p(idx), _p(cnt), _stream())
    return (idx, cnt) if return_count else idx


class BallGrid:
    """The per-cloud cell list of a ball query as a reusable image (include/repsurf_hip.h: rs_ballquery_grid_build): one build
    serves every query on the same coordinates and radius."""

    def __init__(self, radius, xyz):
        _need_gpu(xyz)
        self.xyz = _f32c(xyz)
        self.b, self.n, _ = self.xyz.shape
        self.radius = float(radius)
        self.r2 = torch.tensor(self.radius ** 2, dtype=torch.float32).item()
        nbytes = _lib.load().rs_ballquery_grid_bytes(self.b, self.n)
        self.image = torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=xyz.device)
        _lib.call("rs_ballquery_grid_build", self.b, self.n, self.r2, _p(self.xyz), _p(self.image), _stream())

    def query(self, nsample, new_xyz, return_count=False):
        """-> (B,S,nsample) int32 [, (B,S) counts]: the rows rs_ballquery returns for the same clouds, centres and radius, bit for bit"""
        _need_gpu(new_xyz)
        new_xyz = _f32c(new_xyz)
        m = new_xyz.shape[1]
        idx = torch.empty((self.b, m, nsample), dtype=torch.int32, device=new_xyz.device)
        cnt = torch.empty((self.b, m), dtype=torch.int32, device=new_xyz.device) if return_count else None
        _lib.call("rs_ballquery_grid_query", self.b, self.n, m, self.r2, nsample, _p(new_xyz), _p(self.image), _p(idx), _p(cnt), _stream())
        return (idx, cnt) if return_count else idx


def ballquery_grid_ok(n, nsample):
    """shapes the reusable cell-list image covers (otherwise: ops.ballquery)"""
    return 64 <= n <= 4096 and 1 <= nsample <= 64


def knnquery(nsample, xyz, new_xyz=None, return_dist=False):
    """-> (B,S,nsample) int32 [, squared distances]; query_knn_point(cuda=False) semantics (:102-111)."""
    if new_xyz is None:
        new_xyz = xyz
    _need_gpu(xyz, new_xyz)
    xyz, new_xyz = _f32c(xyz), _f32c(new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = torch.empty((b, m, nsample), dtype=torch.int32, device=xyz.device)
    d2 = torch.empty((b, m, nsample), dtype=torch.float32, device=xyz.device) if return_dist else None
    _lib.call("rs_knnquery", b, n, m, nsample, _p(xyz), _p(new_xyz), _p(idx), _p(d2), _stream())
    return (idx, d2) if return_dist else idx


KNN_GRID = os.environ.get("REPSURF_KNN_GRID", "1") != "0"
UMBRELLA_GRID = os.environ.get("REPSURF_UMBRELLA_GRID", "1") != "0"        # classification: rs_umbrella_features_grid ...
# ... from this many points per cloud (k = 9, stand-alone: 32 x 1024: grid 126 us, scan 92; 64 x 2048: 145 / 383; 16 x 4096: 133 / 340;
# 8 x 8192: 158 / 546).  Round 5: 1 024 instead of 2 048 -- inside the 32 x 1024 step the grid form is the cheaper NEIGHBOUR of the
# network it runs beside (fewer CU-microseconds, although longer alone): 1.2943 -> 1.2814 ms per step, four interleaved pairs on one box
# (round 4 had them level, 1.402 / 1.405); identical lists and features (tests/test_geometry_gpu.py), whole GPU suite green either way.
UMBRELLA_GRID_MIN_ROWS = int(os.environ.get("REPSURF_UMBRELLA_GRID_MIN_ROWS", "1024"))
# rows per cell = fill * nsample.  The ball of the nsample nearest rows has the volume of nsample / density; one wave per query
# (lists of 17 .. 64 entries) wants cells about as large as that ball -- 64 candidates per trip, the 27 cells around the query are
# enough for most queries (16 384 queries over 16 x 4096 rows: fill 0.45 / 1.0 / 2.0 -> 76 / 71 / 68 us) --, one thread per query
# (short lists) wants small cells: every candidate costs the same whether it is near or far (umbrella fans, 65 536 queries: fill
# 0.083 / 0.25 / 0.8 -> 122 / 128 / 145 us).
KNN_GRID_FILL = tuple(float(v) for v in os.environ.get("REPSURF_KNN_GRID_FILL", "0.125,1.0").split(","))
# average rows per cloud from which the grid is used, for lists of <= 3 / <= 16 / <= 64 entries (below, the scan is as fast and
# needs no set-up launch)
KNN_GRID_MIN_ROWS = tuple(int(v) for v in os.environ.get("REPSURF_KNN_GRID_MIN_ROWS", "512,512,128").split(","))


def _largest_cloud(offset):
    """rows of the largest cloud of a packed batch if the host already holds the offsets (no device read), else 0"""
    cached = getattr(offset, "_rs_host", None)
    if cached is None or cached[0] != (offset.data_ptr(), offset._version) or not cached[1]:
        return 0
    ends = cached[1]
    return max(e - s for s, e in zip((0,) + ends[:-1], ends))


def knnquery_offset(nsample, xyz, new_xyz, offset, new_offset, grid=None):
    """Packed batches: xyz (N,3), new_xyz (M,3) -> idx (M,nsample) int32, dist2 (M,nsample).
    nsample <= 64 goes through per-cloud uniform grids (rs_knn_grid_build / rs_knn_grid_query: the cells around the query, ring by
    ring, until the list is provably complete) -- the same lists, bit for bit, as the scan of the whole cloud per query
    (rs_knnquery_offset; grid=False or REPSURF_KNN_GRID=0 selects it)."""
    _need_gpu(xyz, new_xyz, offset, new_offset)
    xyz, new_xyz, offset, new_offset = _f32c(xyz), _f32c(new_xyz), _i32c(offset), _i32c(new_offset)
    m = new_xyz.shape[0]
    dev = xyz.device
    idx = torch.empty((m, nsample), dtype=torch.int32, device=dev)
    d2 = torch.empty((m, nsample), dtype=torch.float32, device=dev)
    b = offset.numel()
    if grid is None:      # by measurement (tools/knn_grid_bench.py): the grid pays above a cloud size that grows with the list length
        grid = KNN_GRID and xyz.shape[0] >= b * (KNN_GRID_MIN_ROWS[0] if nsample <= 3 else KNN_GRID_MIN_ROWS[1] if nsample <= 16 else KNN_GRID_MIN_ROWS[2])
    if grid and 0 < nsample <= 64 and m > 0 and xyz.shape[0] > 0:
        # cloud sizes, when the host has them (offset tensors made by offsets_tensor / strided_offset, or read once by host_offsets):
        # they size the launch and the LDS staging; unknown = the safe bounds
        max_q, max_n = _largest_cloud(new_offset), _largest_cloud(offset)
        rows = torch.empty((xyz.shape[0], 4), dtype=torch.float32, device=dev)
        starts = torch.empty((b, _lib.KNN_GRID_CELLS + 1), dtype=torch.int32, device=dev)
        cells = torch.empty((b, 16), dtype=torch.float32, device=dev)
        _lib.call("rs_knn_grid_build", b, _p(xyz), _p(offset), max(1.5, KNN_GRID_FILL[0 if nsample <= 16 else 1] * nsample), _p(rows), _p(starts), _p(cells), _stream())
        _lib.call("rs_knn_grid_query", m, nsample, b, max_q, max_n, _p(new_xyz), _p(new_offset), _p(rows), _p(starts), _p(cells), _p(idx), _p(d2), _stream())
        return idx, d2
    _lib.call("rs_knnquery_offset", m, nsample, _p(xyz), _p(new_xyz), _p(offset), _p(new_offset),
              offset.numel(), _p(idx), _p(d2), _stream())
    return idx, d2


def umbrella_features(xyz, k=9, inv_sign=None, return_knn=False):
    """xyz (B,N,3) -> (B,N,k-1,10) = [centroid, polar, normal, const] per fan triangle
    (everything UmbrellaSurfaceConstructor does before self.mlps, repsurface_utils.py:276-293)."""
    _need_gpu(xyz, inv_sign)
    xyz = _f32c(xyz)
    b, n, _ = xyz.shape
    feat = torch.empty((b, n, k - 1, 10), dtype=torch.float32, device=xyz.device)
    kidx = torch.empty((b, n, k), dtype=torch.int32, device=xyz.device) if return_knn else None
    sg = None if inv_sign is None else _f32c(inv_sign.reshape(-1))
    if UMBRELLA_GRID and n >= UMBRELLA_GRID_MIN_ROWS and b * n < 2 ** 31:
        # the search through per-cloud uniform grids (csrc/grid_knn.hip): the cells around a point instead of its whole cloud
        dev = xyz.device
        off = offsets_tensor([(i + 1) * n for i in range(b)], dev)
        rows = torch.empty((b * n, 4), dtype=torch.float32, device=dev)
        starts = torch.empty((b, _lib.KNN_GRID_CELLS + 1), dtype=torch.int32, device=dev)
        cells = torch.empty((b, 16), dtype=torch.float32, device=dev)
        _lib.call("rs_umbrella_features_grid", b, n, k, _p(xyz), _p(off), _p(sg), _p(kidx), _p(feat), _p(rows), _p(starts), _p(cells), _stream())
    else:
        _lib.call("rs_umbrella_features", b, n, k, _p(xyz), _p(sg), _p(kidx), _p(feat), _stream())
    return (feat, kidx) if return_knn else feat


# ----------------------------------------------------------------------------- inverse of a gather index
GATHER_BACKWARD = os.environ.get("REPSURF_GATHER_BACKWARD", "1") != "0"


def inverse_index(src, per, edge_ends, point_ends):
    """The edges that read each source row of a packed batch (rs_inverse_index): src (rows, per) | (rows * per,) int32 global source
    rows, edge_ends / point_ends (B,) running ends of the query rows / source rows per cloud -> (csr_off (P + 1), csr_edges (E)) or
    None when the host does not hold the cloud sizes or a cloud has more than 16 384 source rows (the backward then scatters).
    Geometry only: lets the backward of the gather WRITE every gradient element as a sum in ascending edge order (no atomics)."""
    if not GATHER_BACKWARD:
        return None
    largest = _largest_cloud(point_ends)
    if largest <= 0 or largest > 16384:
        return None
    _need_gpu(src, edge_ends, point_ends)
    src, edge_ends, point_ends = _i32c(src), _i32c(edge_ends), _i32c(point_ends)
    points = point_ends._rs_host[1][-1]
    e = src.numel()
    dev = src.device
    csr_off = torch.empty((points + 1,), dtype=torch.int32, device=dev)
    csr_edges = torch.empty((max(e, 1),), dtype=torch.int32, device=dev)
    _lib.call("rs_inverse_index", edge_ends.numel(), per, largest, _p(src), _p(edge_ends), _p(point_ends), _p(csr_off), _p(csr_edges),
              _p(_inverse_overflow(dev)), _stream())
    return csr_off, csr_edges


_inv_overflow = {}


def _inverse_overflow(device):
    """Persistent per-device counter rs_inverse_index adds to when a cloud holds more rows than the host-side offsets said (a stale
    `_rs_host` cache: the lists of that cloud are then not written).  Allocated eagerly -- inside a capture it would live in the graph's
    pool and every replay would reset it -- and read by `inverse_index_overflow_count`."""
    key = _lib.device_key(device)
    if key not in _inv_overflow:
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("repsurf_amd.ops.inverse_index: the first call on a device must run eagerly (one warm-up pass) before capture")
        _inv_overflow[key] = torch.zeros((1,), dtype=torch.int32, device=device)
    return _inv_overflow[key]


def inverse_index_overflow_count(device=None):
    """Clouds (since the process started) whose inverse index could not be built because the device offsets exceeded the host copy the
    launch was sized from; the backward of such a cloud read unwritten lists.  0 in a healthy run: check it where the loss is read."""
    if device is None:
        return sum(int(t.item()) for t in _inv_overflow.values())
    t = _inv_overflow.get(_lib.device_key(device))
    return 0 if t is None else int(t.item())


# ----------------------------------------------------------------------------- grouping
class _GroupFeatures(Function):
    @staticmethod
    def forward(ctx, center, new_center, normal, feature, idx, polar, aligned, csr=None):
        _need_gpu(center, new_center, normal, feature, idx)
        center, new_center, normal, idx = _f32c(center), _f32c(new_center), _f32c(normal), _i32c(idx)
        feature = None if feature is None else _f32c(feature)
        b, n, _ = center.shape
        _, m, ns = idx.shape
        cn = normal.shape[2]
        cf = 0 if feature is None else feature.shape[2]
        cpos = 6 if polar else 3
        pad = (-cpos) % 4 if aligned else 0                       # position block padded to a float4 boundary
        ctot = cpos + pad + cn + cf
        ldo = -(-ctot // 4) * 4 if aligned else ctot              # rows a multiple of 16 bytes apart
        out = torch.empty((b * m * ns, ldo), dtype=torch.float32, device=center.device)
        _lib.call("rs_group_features", b, n, m, ns, cn, cf, int(polar), _p(center), _p(new_center),
                  _p(normal), _p(feature), _p(idx), _p(out), pad, ldo, _stream())
        ctx.save_for_backward(idx)
        ctx.csr = csr if (csr is not None and b == 1 and csr[0].numel() == n + 1 and csr[1].numel() >= m * ns) else None
        ctx.dims = (b, n, m, ns, cn, cf, int(polar), pad, ldo)
        ctx.need = (ctx.needs_input_grad[2], feature is not None and ctx.needs_input_grad[3])
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        b, n, m, ns, cn, cf, polar, pad, ldo = ctx.dims
        grad_out = _f32c(grad_out)
        dev = grad_out.device
        if ctx.csr is not None:      # gather form (ops.inverse_index of the grouping index, built with the geometry): no fill, no atomics
            c0, c1 = (cn if ctx.need[0] else 0), (cf if ctx.need[1] else 0)
            if c0 + c1 == 0:
                return (None,) * 8
            buf = torch.empty((n * (c0 + c1),), dtype=torch.float32, device=dev)
            gn = buf[:n * c0].view(1, n, c0) if c0 else None
            gf = buf[n * c0:].view(1, n, c1) if c1 else None
            _lib.call("rs_group_features_backward_csr", n, cn, cf, (6 if polar else 3) + pad, ldo, _p(grad_out), _p(ctx.csr[0]), _p(ctx.csr[1]),
                      _p(gn), _p(gf), _stream())
            return None, None, gn, gf, None, None, None, None
        gn, gf = _zero_pair(b * n, cn if ctx.need[0] else 0, cf if ctx.need[1] else 0, (b, n), dev)
        if gn is not None or gf is not None:
            groups_dev = _ragged.dev(m) if b == 1 else None      # (a packed batch under a captured capacity: groups beyond the count are not scattered)
            if groups_dev is not None:
                _lib.call("rs_group_features_backward_dev", b, n, m, ns, cn, cf, polar, _p(grad_out), _p(idx),
                          _p(gn), _p(gf), pad, ldo, groups_dev, _stream())
            else:
                _lib.call("rs_group_features_backward", b, n, m, ns, cn, cf, polar, _p(grad_out), _p(idx),
                          _p(gn), _p(gf), pad, ldo, _stream())
        return None, None, gn, gf, None, None, None, None


def group_features(center, new_center, normal, feature, idx, polar=True, aligned=False, csr=None):
    """Grouped shared-MLP input of sample_and_group (repsurface_utils.py:36-57):
    -> (B*S*ns, 3+3*polar+Cn+Cf) rows [offset, polar(offset), normal[idx], feature[idx]];
    differentiable w.r.t. normal and feature.
    aligned=True: the position block is zero-padded to 4 channels and rows to a multiple of 4 floats --
    (rows, ld) with [offset(3|6), 0.., normal, feature, unused..]; `aligned_layout` gives the offsets."""
    return _GroupFeatures.apply(center, new_center, normal, feature, idx, polar, aligned, csr)


def aligned_layout(polar, cn, cf):
    """(feature-branch offset, feature-branch width, row stride) of group_features(aligned=True)."""
    cpos = 6 if polar else 3
    off = cpos + (-cpos) % 4
    return off, cn + cf, -(-(off + cn + cf) // 4) * 4


class CompactGroups:
    """Grouped shared-MLP operand with the padding copies of ball-query rows removed.
    x (capacity, C): rows [offsets[g], offsets[g+1]) belong to group g, offsets[groups] rows are valid (a
    device-side count: no host sync); mult[row] = number of dense rows it stands for."""
    __slots__ = ("x", "mult", "grp", "slot", "src", "offsets", "groups", "nsample", "rows_full")

    def __init__(self, x, mult, grp, slot, src, offsets, groups, nsample):
        self.x, self.mult, self.grp, self.slot, self.src, self.offsets = x, mult, grp, slot, src, offsets
        self.groups, self.nsample = groups, nsample
        self.rows_full = groups * nsample        # what BatchNorm statistics are taken over

    @property
    def rows_dev_ptr(self):
        return self.offsets.data_ptr() + 4 * self.groups


class CompactIndex:
    """The bookkeeping of compacted groups (rs_compact_index): offsets (groups + 1), mult / grp / slot / src (capacity).  It
    depends on the ball query's (idx, cnt) only, so the geometry stage of a pipelined step builds it ahead of time.
    csr (round 4, optional): the inverse of `src` for the gather form of the backward (rs_compact_csr): `pts` (2 P + 1 ints:
    csr_off (P + 1) then centre_of (P), P = source points) and `csr_rows` (capacity; the fourth row of `meta`)."""
    __slots__ = ("offsets", "mult", "grp", "slot", "src", "meta", "pts", "csr_rows", "csr_ready")

    def __init__(self, offsets, mult, grp, slot, src, meta=None, pts=None, csr_rows=None):
        """meta: the (3 | 4, capacity) int32 allocation grp / slot / src (/ csr_rows) are rows of, when they were allocated that way
        (copied and cloned as ONE tensor by the pipelined step's state sets); None when they are separate tensors."""
        self.offsets, self.mult, self.grp, self.slot, self.src, self.meta = offsets, mult, grp, slot, src, meta
        self.pts, self.csr_rows = pts, csr_rows
        self.csr_ready = False          # set by whoever ran rs_compact_csr on these tensors

    @staticmethod
    def from_meta(offsets, mult, meta, pts=None):
        return CompactIndex(offsets, mult, meta[0], meta[1], meta[2], meta, pts, meta[3] if (pts is not None and meta.shape[0] > 3) else None)

    @staticmethod
    def empty(groups, nsample, dev, points=0):
        """points > 0: with room for the inverse of `src` over that many source points"""
        cap = groups * nsample
        meta = torch.empty((4 if points else 3, cap), dtype=torch.int32, device=dev)
        pts = torch.empty((2 * points + 1,), dtype=torch.int32, device=dev) if points else None
        return CompactIndex.from_meta(torch.empty((groups + 1,), dtype=torch.int32, device=dev),
                                      torch.empty((cap,), dtype=torch.float32, device=dev), meta, pts)

    def csr(self, points):
        """(csr_off, centre_of, csr_rows) views, or None"""
        if self.pts is None or self.csr_rows is None or not self.csr_ready or self.pts.numel() != 2 * points + 1:
            return None
        return self.pts[:points + 1], self.pts[points + 1:], self.csr_rows


COMPACT_CSR = os.environ.get("REPSURF_COMPACT_CSR", "1") != "0"        # the compacted grouping's backward as a gather (rs_compact_csr)


def compact_index(idx, cnt, n, out=None, stream=None, csr=False, fps_idx=None):
    """idx (B, M, ns), cnt (B, M) of a ball query over clouds of n points -> CompactIndex.
    csr=True: also the inverse of `src` (rs_compact_csr; fps_idx (B, M): the centres' own rows) -- the grouping's backward then gathers."""
    _need_gpu(idx, cnt)
    idx, cnt = _i32c(idx), _i32c(cnt)
    b, m, ns = idx.shape
    ci = out if out is not None else CompactIndex.empty(b * m, ns, idx.device, points=b * n if csr else 0)
    st = _stream() if stream is None else stream
    _lib.call("rs_compact_index", b, n, m, ns, _p(idx), _p(cnt), _p(ci.offsets), _p(ci.grp), _p(ci.slot), _p(ci.src), _p(ci.mult), st)
    if csr and ci.pts is not None:
        fps = None if fps_idx is None else _i32c(fps_idx)
        _lib.call("rs_compact_csr", b, n, m, _p(ci.src), _p(ci.offsets), _p(fps), ci.pts.data_ptr(), ci.pts.data_ptr() + 4 * (b * n + 1),
                  _p(ci.csr_rows), st)
        ci.csr_ready = True
    return ci


def _row_stride(t, rows, c):
    """t viewed as (rows, c) with unit inner stride: its row stride, or None when it is not that regular"""
    if t.numel() != rows * c or t.stride(-1) != 1:
        return None
    ld = t.stride(-2)
    if ld < c:
        return None
    lead, step = t.shape[:-2], ld * t.shape[-2]
    for d in range(len(lead) - 1, -1, -1):          # leading dimensions must continue the same row pitch
        if lead[d] != 1 and t.stride(d) != step:
            return None
        step *= lead[d]
    return ld


class _GroupFeaturesCompact(Function):
    @staticmethod
    def forward(ctx, center, new_center, normal, feature, idx, cnt, polar, index, fps_idx):
        _need_gpu(center, new_center, normal, feature, idx, cnt)
        center, new_center, normal = _f32c(center), _f32c(new_center), _f32c(normal)
        idx, cnt = _i32c(idx), _i32c(cnt)
        feature = None if feature is None else _f32c(feature)
        b, n, _ = center.shape
        _, m, ns = idx.shape
        cn = normal.shape[2]
        cf = 0 if feature is None else feature.shape[2]
        ctot = (6 if polar else 3) + cn + cf
        dev = center.device
        groups, cap = b * m, b * m * ns
        have = index is not None
        if not have:
            index = CompactIndex.empty(groups, ns, dev)
            _lib.call("rs_exclusive_scan", groups, _p(cnt), _p(index.offsets), _stream())
        offsets, mult, grp, slot, src = index.offsets, index.mult, index.grp, index.slot, index.src
        out = torch.empty((cap, ctot), dtype=torch.float32, device=dev)
        fps_idx = None if (fps_idx is None or cn == 0) else _i32c(fps_idx)
        new_normal = None if fps_idx is None else torch.empty((b, m, cn), dtype=torch.float32, device=dev)
        _lib.call("rs_group_features_compact", b, n, m, ns, cn, cf, int(polar), _p(center), _p(new_center),
                  _p(normal), _p(feature), _p(idx), _p(cnt), _p(offsets), _p(out), _p(mult), _p(grp), _p(slot),
                  _p(src), int(have), _p(fps_idx), _p(new_normal), _stream())
        ctx.save_for_backward(src, offsets, fps_idx)
        ctx.csr = index.csr(b * n) if (have and COMPACT_CSR) else None
        ctx.dims = (b, n, m, cn, cf, int(polar), cap, groups)
        ctx.need = (ctx.needs_input_grad[2], feature is not None and ctx.needs_input_grad[3])
        ctx.mark_non_differentiable(mult, grp, slot, src, offsets)
        ctx.set_materialize_grads(False)      # no zero-filled "gradients" for the index/bookkeeping outputs or an unused new_normal
        if new_normal is None:
            new_normal = torch.empty((0,), dtype=torch.float32, device=dev)
            ctx.mark_non_differentiable(new_normal)
        return out, new_normal, mult, grp, slot, src, offsets

    @staticmethod
    def backward(ctx, grad_out, grad_new_normal, *unused):
        src, offsets, fps_idx = ctx.saved_tensors
        b, n, m, cn, cf, polar, cap, groups = ctx.dims
        if grad_out is None and grad_new_normal is None:
            return (None,) * 9
        dev = src.device
        centre = fps_idx is not None and grad_new_normal is not None and ctx.need[0]
        if grad_out is not None and ctx.csr is not None and (not centre or fps_idx is not None):
            # gather form: per source point, the sum over the compacted rows that name it (+ its own centre row's gradient): every
            # element written once -- no zero fill, no atomics, a fixed order (rs_compact_csr ran in the geometry stage)
            c0, c1 = (cn if ctx.need[0] else 0), (cf if ctx.need[1] else 0)
            if c0 + c1 == 0:
                return (None,) * 9
            buf = torch.empty((b * n * (c0 + c1),), dtype=torch.float32, device=dev)
            gn = buf[:b * n * c0].view(b, n, c0) if c0 else None
            gf = buf[b * n * c0:].view(b, n, c1) if c1 else None
            grad_out = _f32c(grad_out)
            ldg = 0
            if centre:
                ldg = _row_stride(grad_new_normal, groups, cn)
                if ldg is None:
                    grad_new_normal, ldg = _f32c(grad_new_normal), cn
            csr_off, centre_of, csr_rows = ctx.csr
            _lib.call("rs_group_features_compact_backward_csr", b, n, cn, cf, polar, _p(grad_out), _p(csr_off), _p(csr_rows), _p(centre_of),
                      _p(gn), _p(gf), _p(grad_new_normal) if centre else None, ldg, _p(fps_idx) if centre else None, m, _stream())
            return None, None, gn, gf, None, None, None, None, None
        gn, gf = _zero_pair(b * n, cn if ctx.need[0] else 0, cf if (ctx.need[1] and grad_out is not None) else 0, (b, n), dev)
        if grad_out is None:       # only the centres' own rows were used downstream
            if centre:
                g = _f32c(grad_new_normal)
                _lib.call("rs_gather_rows_backward", b, n, m, cn, _p(g), _p(fps_idx), _p(gn), _stream())
            return None, None, gn, gf, None, None, None, None, None
        grad_out = _f32c(grad_out)
        ldg = 0
        if centre:
            ldg = _row_stride(grad_new_normal, groups, cn)
            if ldg is None:
                grad_new_normal, ldg = _f32c(grad_new_normal), cn
        if gn is not None or gf is not None:
            _lib.call("rs_group_features_compact_backward", cap, offsets.data_ptr() + 4 * groups, cn, cf, polar,
                      _p(grad_out), _p(src), _p(gn), _p(gf), b, n, m, _p(fps_idx) if centre else None,
                      _p(grad_new_normal) if centre else None, ldg, _stream())
        return None, None, gn, gf, None, None, None, None, None


def group_features_compact(center, new_center, normal, feature, idx, cnt, polar=True, index=None, fps_idx=None):
    """Compacted grouped operand (see CompactGroups); differentiable w.r.t. normal and feature.
    index: a CompactIndex built ahead of time from the same (idx, cnt) (else built here).
    fps_idx (B, M): also return index_points(normal, fps_idx) -- the centres' own normal rows -- from the same launches
    (forward: inside the gather kernel; backward: inside the scatter kernel, into the same gradient buffer, instead of a
    fill + scatter + add of their own): -> (CompactGroups, new_normal (B, M, cn))."""
    out, new_normal, mult, grp, slot, src, offsets = _GroupFeaturesCompact.apply(center, new_center, normal, feature, idx, cnt, polar,
                                                                                 index, fps_idx)
    groups = CompactGroups(out, mult, grp, slot, src, offsets, idx.shape[0] * idx.shape[1], idx.shape[2])
    return groups if fps_idx is None else (groups, new_normal)


class _GroupAllFeatures(Function):
    @staticmethod
    def forward(ctx, center, normal, feature, polar):
        _need_gpu(center, normal, feature)
        center, normal = _f32c(center), _f32c(normal)
        feature = None if feature is None else _f32c(feature)
        b, n, _ = center.shape
        cn, cf = normal.shape[2], 0 if feature is None else feature.shape[2]
        cpos = 6 if polar else 3
        out = torch.empty((b * n, cpos + cn + cf), dtype=torch.float32, device=center.device)
        _lib.call("rs_group_all_features", b, n, cn, cf, int(polar), _p(center), _p(normal), _p(feature), _p(out), _stream())
        ctx.dims = (b, n, cpos, cn, cf)
        return out

    @staticmethod
    def backward(ctx, grad):
        b, n, cpos, cn, cf = ctx.dims
        # column slices of the incoming gradient, as views: the consumers (rs_pool_max_backward's `ldd`, the compacted
        # grouping's `ldg`) read rows at a pitch, so nothing is copied
        gn = grad[:, cpos:cpos + cn].view(b, n, cn) if ctx.needs_input_grad[1] else None
        gf = grad[:, cpos + cn:].view(b, n, cf) if (cf and ctx.needs_input_grad[2]) else None
        return None, gn, gf, None


def group_all_features(center, normal, feature, polar=True):
    """sample_and_group_all (repsurface_utils.py:62-88) -> (B*N, 3+3*polar+Cn+Cf), one launch; differentiable w.r.t. normal
    and feature (their gradients are column slices of the row gradient, returned as views)."""
    return _GroupAllFeatures.apply(center, normal, feature, polar)


# ----------------------------------------------------------------------------- three-NN interpolation
def three_nn(unknown, known):
    """-> (sqrt-free) squared distances (B,n,3), idx (B,n,3) int32 (pointops.nearestneighbor)."""
    _need_gpu(unknown, known)
    unknown, known = _f32c(unknown), _f32c(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    d2 = torch.empty((b, n, 3), dtype=torch.float32, device=unknown.device)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=unknown.device)
    _lib.call("rs_three_nn", b, n, m, _p(unknown), _p(known), _p(d2), _p(idx), _stream())
    return d2, idx


class _ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, points, idx, weight):
        _need_gpu(points, idx, weight)
        points, idx, weight = _f32c(points), _i32c(idx), _f32c(weight)
        b, m, c = points.shape
        n = idx.shape[1]
        out = torch.empty((b, n, c), dtype=torch.float32, device=points.device)
        _lib.call("rs_three_interpolate", b, c, m, n, _p(points), _p(idx), _p(weight), _p(out), _stream())
        ctx.save_for_backward(idx, weight)
        ctx.dims = (b, c, n, m)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        b, c, n, m = ctx.dims
        grad_out = _f32c(grad_out)
        grad = torch.zeros((b, m, c), dtype=torch.float32, device=grad_out.device)
        _lib.call("rs_three_interpolate_backward", b, c, n, m, _p(grad_out), _p(idx), _p(weight), _p(grad), _stream())
        return grad, None, None


def three_interpolate(points, idx, weight):
    """points (B,m,C), idx/weight (B,n,3) -> (B,n,C); differentiable w.r.t. points."""
    return _ThreeInterpolate.apply(points, idx, weight)


class _ThreeInterpolateAddRelu(Function):
    """relu(three_interpolate(points, idx, weight) + add): one launch forward, one backward (the masked gradient is scattered to
    `points` and written out once for `add`)."""

    @staticmethod
    def forward(ctx, points, idx, weight, add, csr=None):
        _need_gpu(points, idx, weight, add)
        points, idx, weight = _f32c(points), _i32c(idx), _f32c(weight)
        add = None if add is None else _f32c(add)
        b, m, c = points.shape
        n = idx.shape[1]
        out = torch.empty((b, n, c), dtype=torch.float32, device=points.device)
        _lib.call("rs_three_interpolate_fused", b, c, m, n, _p(points), _p(idx), _p(weight), _p(add), 1, _p(out), _stream())
        ctx.save_for_backward(idx, weight, out)
        ctx.dims = (b, c, n, m)
        ctx.has_add = add is not None
        # gather form of the backward (ops.inverse_index of idx, per = 3): packed layout (b = 1), no skip branch here
        ctx.csr = csr if (csr is not None and b == 1 and add is None and csr[0].numel() == m + 1 and m * c < 2 ** 31) else None
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, out = ctx.saved_tensors
        b, c, n, m = ctx.dims
        grad_out = _f32c(grad_out)
        if ctx.csr is not None:
            grad = torch.empty((b, m, c), dtype=torch.float32, device=grad_out.device)
            _lib.call("rs_three_interpolate_backward_csr", m, c, None, _p(grad_out), _p(out), _p(weight), _p(ctx.csr[0]), _p(ctx.csr[1]), _p(grad),
                      None, None, None, None, 0, _stream())
            return grad, None, None, None, None
        grad = torch.zeros((b, m, c), dtype=torch.float32, device=grad_out.device)
        gadd = torch.empty((b, n, c), dtype=torch.float32, device=grad_out.device) if (ctx.has_add and ctx.needs_input_grad[3]) else None
        rows_dev = _ragged.dev(n) if b == 1 else None      # (a packed batch under a captured capacity)
        if rows_dev is not None:
            _lib.call("rs_three_interpolate_fused_backward_dev", b, c, n, m, _p(grad_out), _p(out), _p(idx), _p(weight), _p(grad), _p(gadd), rows_dev, _stream())
        else:
            _lib.call("rs_three_interpolate_fused_backward", b, c, n, m, _p(grad_out), _p(out), _p(idx), _p(weight), _p(grad), _p(gadd), _stream())
        return grad, None, None, gadd, None


def three_interpolate_add_relu(points, idx, weight, add=None, csr=None):
    """relu(three_interpolate(points, idx, weight) [+ add]): points (B,m,C), idx / weight (B,n,3), add (B,n,C) | None -> (B,n,C);
    differentiable w.r.t. points and add (segmentation/modules/repsurface_utils.py:266-270 in one launch each way)."""
    return _ThreeInterpolateAddRelu.apply(points, idx, weight, add, csr)
