"""The reference's OWN pointops kernels, executed on the CPU (TEST INFRASTRUCTURE ONLY).

oracle/Makefile.ref compiles /root/reference/{classification,segmentation}/modules/pointops/src/*/*_cuda_kernel.cu,
unmodified, as host C++ (oracle/ref_shim/) into oracle/_ref/libref_pointops_{cls,seg}.so.  This file is the Python
side the reference's `*_cuda.cpp` wrappers + `pointops_api.cpp` would be: `module(kind)` returns an object that
answers to the `pointops_cuda.<name>(ints…, tensors…)` calls of the reference's
`modules/pointops/functions/pointops.py` by handing the tensors' data pointers to the matching `*_launcher`,
argument for argument (classification/modules/pointops/src/pointops_api.cpp:13-31,
segmentation/modules/pointops/src/pointops_api.cpp:12-22).  Installed as `sys.modules["pointops_cuda"]` it lets the
reference's Python run end to end on CPU tensors with its own kernels underneath
(tests/golden/make_golden_seg.py); the oracle restatements are pinned against it in tests/test_oracle_ref.py.

Only tests/, tests/golden/make_*.py, __graft_entry__.build() and bench.py's cpu_baseline leg may touch this.
"""
import ctypes
import os
import subprocess
import types

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_DIR = os.path.join(_HERE, "_ref")
REFERENCE = "/root/reference"

# python name (pointops_api.cpp m.def) -> (launcher symbol, launcher takes a trailing cudaStream_t)
_CLS = {
    "ballquery_cuda": ("ballquery_cuda_launcher_fast", True),
    "knnquery_cuda": ("knnquery_cuda_launcher", True),
    "knnquery_heap_cuda": ("knnquery_heap_cuda_launcher", True),
    "grouping_forward_cuda": ("grouping_forward_cuda_launcher_fast", False),
    "grouping_backward_cuda": ("grouping_backward_cuda_launcher", False),
    "grouping_int_forward_cuda": ("grouping_int_forward_cuda_launcher_fast", False),
    "gathering_forward_cuda": ("gathering_forward_cuda_launcher", False),
    "gathering_backward_cuda": ("gathering_backward_cuda_launcher", False),
    "furthestsampling_cuda": ("furthestsampling_cuda_launcher", False),
    "nearestneighbor_cuda": ("nearestneighbor_cuda_launcher_fast", False),
    "interpolation_forward_cuda": ("interpolation_forward_cuda_launcher_fast", False),
    "interpolation_backward_cuda": ("interpolation_backward_cuda_launcher", False),
}
_SEG = {n: (n + "_launcher", False) for n in (
    "knnquery_cuda", "furthestsampling_cuda", "grouping_forward_cuda", "grouping_backward_cuda",
    "interpolation_forward_cuda", "interpolation_backward_cuda", "subtraction_forward_cuda",
    "subtraction_backward_cuda", "aggregation_forward_cuda", "aggregation_backward_cuda")}
_TABLE = {"cls": _CLS, "seg": _SEG}
_libs = {}


def so_path(kind):
    return os.path.join(_DIR, "libref_pointops_%s.so" % kind)


def available(kind="seg"):
    return os.path.exists(so_path(kind))


def build():
    """Compile oracle/_ref from the reference sources (build container only: needs /root/reference)."""
    if not os.path.isdir(REFERENCE):
        return False
    subprocess.check_call(["make", "-s", "-f", "oracle/Makefile.ref"], cwd=_ROOT)
    return True


def _lib(kind):
    if kind not in _libs:
        if not available(kind):
            raise FileNotFoundError(so_path(kind) + " (run `make -f oracle/Makefile.ref` where /root/reference exists)")
        _libs[kind] = ctypes.CDLL(so_path(kind))
    return _libs[kind]


def _arg(a):
    if isinstance(a, (bool, int, np.integer)):
        return ctypes.c_int(int(a))
    if isinstance(a, (float, np.floating)):
        return ctypes.c_float(float(a))
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return ctypes.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr") and a.dim() == 0:     # e.g. n_max = offset[0] (pointops.py:43): pybind11 casts it to int
        return ctypes.c_int(int(a.item()))
    if hasattr(a, "data_ptr"):                      # torch CPU tensor
        assert a.device.type == "cpu" and a.is_contiguous()
        return ctypes.c_void_p(a.data_ptr())
    raise TypeError(type(a))


def module(kind):
    """A stand-in for the compiled `pointops_cuda` extension of the `kind` ("cls" | "seg") sub-project."""
    lib = _lib(kind)
    mod = types.ModuleType("pointops_cuda")
    for name, (sym, stream) in _TABLE[kind].items():
        fn = getattr(lib, sym)
        fn.restype = None

        def call(*args, _fn=fn, _stream=stream):
            cargs = [_arg(a) for a in args]
            if _stream:
                cargs.append(ctypes.c_void_p(0))
            _fn(*cargs)
        mod.__dict__[name] = call
    return mod


# ---- numpy conveniences used by tests/test_oracle_ref.py (layouts of the reference kernels)
def seg_knn(k, xyz, new_xyz, offset, new_offset):
    """segmentation/modules/pointops/functions/pointops.py:114-130 minus the sqrt: (idx (m,k) i32, dist2 (m,k) f32)."""
    xyz, new_xyz = np.ascontiguousarray(xyz, np.float32), np.ascontiguousarray(new_xyz, np.float32)
    offset, new_offset = np.ascontiguousarray(offset, np.int32), np.ascontiguousarray(new_offset, np.int32)
    m = new_xyz.shape[0]
    idx, d2 = np.zeros((m, k), np.int32), np.zeros((m, k), np.float32)
    module("seg").knnquery_cuda(m, k, xyz, new_xyz, offset, new_offset, idx, d2)
    return idx, d2


def seg_fps(xyz, offset, new_offset):
    """segmentation/modules/pointops/functions/pointops.py:31-49: tmp = 1e10, n = largest cloud."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    offset, new_offset = np.ascontiguousarray(offset, np.int32), np.ascontiguousarray(new_offset, np.int32)
    n, b, n_max = xyz.shape[0], offset.shape[0], int(offset[0])
    for i in range(1, b):
        n_max = max(int(offset[i] - offset[i - 1]), n_max)
    idx = np.zeros((int(new_offset[b - 1]),), np.int32)
    tmp = np.full((n,), 1e10, np.float32)
    module("seg").furthestsampling_cuda(b, n_max, xyz, offset, new_offset, tmp, idx)
    return idx


def cls_three_nn(unknown, known):
    """classification/modules/pointops/functions/pointops.py:86-109 minus the sqrt: (dist2 (b,n,3), idx (b,n,3))."""
    unknown, known = np.ascontiguousarray(unknown, np.float32), np.ascontiguousarray(known, np.float32)
    b, n, _ = unknown.shape
    m = known.shape[1]
    d2, idx = np.zeros((b, n, 3), np.float32), np.zeros((b, n, 3), np.int32)
    module("cls").nearestneighbor_cuda(b, n, m, unknown, known, d2, idx)
    return d2, idx


def cls_three_interpolate(points_bcm, idx, weight):
    """classification/.../pointops.py:112-147 forward: points (b,c,m), idx/weight (b,n,3) -> (b,c,n)."""
    points_bcm, weight = np.ascontiguousarray(points_bcm, np.float32), np.ascontiguousarray(weight, np.float32)
    idx = np.ascontiguousarray(idx, np.int32)
    b, c, m = points_bcm.shape
    n = idx.shape[1]
    out = np.zeros((b, c, n), np.float32)
    module("cls").interpolation_forward_cuda(b, c, m, n, points_bcm, idx, weight, out)
    return out


def cls_three_interpolate_backward(grad_out_bcn, idx, weight, m):
    grad_out_bcn, weight = np.ascontiguousarray(grad_out_bcn, np.float32), np.ascontiguousarray(weight, np.float32)
    idx = np.ascontiguousarray(idx, np.int32)
    b, c, n = grad_out_bcn.shape
    grad = np.zeros((b, c, m), np.float32)
    module("cls").interpolation_backward_cuda(b, c, n, m, grad_out_bcn, idx, weight, grad)
    return grad


def cls_ballquery(radius, nsample, xyz, new_xyz):
    xyz, new_xyz = np.ascontiguousarray(xyz, np.float32), np.ascontiguousarray(new_xyz, np.float32)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)
    module("cls").ballquery_cuda(b, n, m, float(radius), nsample, new_xyz, xyz, idx)
    return idx


def cls_knn(k, xyz, new_xyz):
    xyz, new_xyz = np.ascontiguousarray(xyz, np.float32), np.ascontiguousarray(new_xyz, np.float32)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx, d2 = np.zeros((b, m, k), np.int32), np.zeros((b, m, k), np.float32)
    module("cls").knnquery_cuda(b, n, m, k, xyz, new_xyz, idx, d2)
    return idx, d2


def cls_fps(xyz, m):
    """classification/.../pointops.py:35-54: starts at index 0 (sampling_cuda_kernel.cu:72-74), temp = 1e10."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    b, n, _ = xyz.shape
    idx = np.zeros((b, m), np.int32)
    tmp = np.full((b, n), 1e10, np.float32)
    module("cls").furthestsampling_cuda(b, n, m, xyz, tmp, idx)
    return idx
