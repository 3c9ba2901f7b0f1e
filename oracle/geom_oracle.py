"""numpy/ctypes front-end of oracle/geom_oracle.c (TEST INFRASTRUCTURE ONLY — see the C header).

Every function takes/returns numpy arrays in the channels-last layouts of include/repsurf_hip.h.
`build()` compiles the C file with gcc (-ffp-contract=off) into oracle/_build/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgeom_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "geom_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off",
                               "-fno-fast-math", "-fopenmp", "-o", _SO, src, "-lm"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def fps(xyz, m, start=None):
    xyz = _f(xyz)
    b, n, _ = xyz.shape
    idx = np.empty((b, m), np.int32)
    st = None if start is None else _i(start)
    lib().oracle_fps(b, n, m, _p(xyz), _p(st), _p(idx))
    return idx


def radius2_of(radius):
    """float32(radius ** 2 in double): what `sqrdists > radius ** 2` compares against
    (classification/modules/pointnet2_utils.py:90)."""
    return np.float32(float(radius) ** 2)


def ballquery(radius, nsample, xyz, new_xyz):
    xyz, new_xyz = _f(xyz), _f(new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = np.empty((b, m, nsample), np.int32)
    lib().oracle_ballquery(b, n, m, ctypes.c_float(radius2_of(radius)), nsample, _p(new_xyz), _p(xyz), _p(idx))
    return idx


def knn(k, xyz, new_xyz, return_dist=False):
    xyz, new_xyz = _f(xyz), _f(new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = np.empty((b, m, k), np.int32)
    d2 = np.empty((b, m, k), np.float32) if return_dist else None
    lib().oracle_knn(b, n, m, k, _p(xyz), _p(new_xyz), _p(idx), _p(d2))
    return (idx, d2) if return_dist else idx


def umbrella(xyz, k=9, inv_sign=None):
    """-> feat (b,n,k-1,10), knn_idx (b,n,k), near_tie (b,n) bool"""
    xyz = _f(xyz)
    b, n, _ = xyz.shape
    feat = np.empty((b, n, k - 1, 10), np.float32)
    kidx = np.empty((b, n, k), np.int32)
    tie = np.zeros((b, n), np.uint8)
    sg = None if inv_sign is None else _f(inv_sign)
    lib().oracle_umbrella(b, n, k, _p(xyz), _p(sg), _p(kidx), _p(feat), _p(tie))
    return feat, kidx, tie.astype(bool)


def group_features(center, new_center, normal, feature, idx, polar=True):
    center, new_center, normal = _f(center), _f(new_center), _f(normal)
    idx = _i(idx)
    b, n, _ = center.shape
    _, m, ns = idx.shape
    cn = normal.shape[2]
    cf = 0 if feature is None else feature.shape[2]
    ft = None if feature is None else _f(feature)
    out = np.empty((b * m * ns, (6 if polar else 3) + cn + cf), np.float32)
    lib().oracle_group_features(b, n, m, ns, cn, cf, int(polar), _p(center), _p(new_center), _p(normal),
                                _p(ft), _p(idx), _p(out))
    return out


def group_all_features(center, normal, feature, polar=True):
    center, normal = _f(center), _f(normal)
    b, n, _ = center.shape
    cn = normal.shape[2]
    cf = 0 if feature is None else feature.shape[2]
    ft = None if feature is None else _f(feature)
    out = np.empty((b * n, (6 if polar else 3) + cn + cf), np.float32)
    lib().oracle_group_all_features(b, n, cn, cf, int(polar), _p(center), _p(normal), _p(ft), _p(out))
    return out


def three_nn(unknown, known):
    unknown, known = _f(unknown), _f(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    d2 = np.empty((b, n, 3), np.float32)
    idx = np.empty((b, n, 3), np.int32)
    lib().oracle_three_nn(b, n, m, _p(unknown), _p(known), _p(d2), _p(idx))
    return d2, idx


def three_interpolate(points, idx, weight):
    points, weight, idx = _f(points), _f(weight), _i(idx)
    b, m, c = points.shape
    n = idx.shape[1]
    out = np.empty((b, n, c), np.float32)
    lib().oracle_three_interpolate(b, c, m, n, _p(points), _p(idx), _p(weight), _p(out))
    return out


def fps_offset(xyz, offset, new_offset):
    xyz, offset, new_offset = _f(xyz), _i(offset), _i(new_offset)
    idx = np.empty((int(new_offset[-1]),), np.int32)
    lib().oracle_fps_offset(len(offset), _p(xyz), _p(offset), _p(new_offset), _p(idx))
    return idx


def knn_offset(k, xyz, new_xyz, offset, new_offset):
    xyz, new_xyz, offset, new_offset = _f(xyz), _f(new_xyz), _i(offset), _i(new_offset)
    m = new_xyz.shape[0]
    idx = np.empty((m, k), np.int32)
    d2 = np.empty((m, k), np.float32)
    lib().oracle_knn_offset(m, k, len(offset), _p(xyz), _p(new_xyz), _p(offset), _p(new_offset), _p(idx), _p(d2))
    return idx, d2


def umbrella_fan_offset(xyz, new_xyz, knn_idx, new_offset, inv_sign=None, rotate=True):
    """-> feat (m,k,10) = [polar, normal, const, centroid], near_tie (m,) bool"""
    xyz, new_xyz, knn_idx, new_offset = _f(xyz), _f(new_xyz), _i(knn_idx), _i(new_offset)
    m, k = knn_idx.shape
    feat = np.empty((m, k, 10), np.float32)
    tie = np.zeros((m,), np.uint8)
    sg = None if inv_sign is None else _f(inv_sign)
    lib().oracle_umbrella_fan_offset(m, k, len(new_offset), int(rotate), _p(xyz), _p(new_xyz), _p(knn_idx),
                                     _p(new_offset), _p(sg), _p(feat), _p(tie))
    return feat, tie.astype(bool)


def interp_weights(dist2):
    dist2 = _f(dist2)
    w = np.empty_like(dist2)
    lib().oracle_interp_weights(ctypes.c_longlong(dist2.shape[0]), _p(dist2), _p(w))
    return w
