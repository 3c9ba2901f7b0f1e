"""A `pointops_cuda` for the reference's UNMODIFIED `modules/pointops/functions/pointops.py` files: the extension-module
names those files call (classification/modules/pointops/src/pointops_api.cpp:13-31,
segmentation/modules/pointops/src/pointops_api.cpp:12-22), answered by librepsurf_hip.so through its C ABI.

    import repsurf_amd.pointops_cuda as pc
    pc.install("classification")      # or "segmentation": sys.modules["pointops_cuda"] = the matching namespace
    from modules.pointops.functions import pointops      # the reference's own file, on an MI355X

Contract of the reference's extension kept: the caller allocates outputs and scratch, functions write in place and return
None, index tensors are int32, everything runs on the current stream.  Differences a maintainer should know:
  * classification operators are channels-first (b, c, n); the library is channels-last, so gathering / grouping /
    interpolation transpose around the kernels (the mirror modules of this package call the channels-last entry points
    directly and never transpose);
  * ballquery / knnquery (classification) use the CPU path's expanded distance formula (DESIGN.md §1 `cuda=` note): the
    same neighbour lists as the CUDA kernels except where the two formulas round across the radius / swap two neighbours
    (0 rows of 1 536 in tests/golden/cls_pointops.npz);
  * the Point-Transformer operators (subtraction_*, aggregation_*) are out of scope and raise.
"""
import sys
import types

import torch

from . import _lib


def _p(t):
    return None if t is None else t.data_ptr()


def _s():
    return _lib.current_stream()


def _cl(t):          # (b, c, n) -> contiguous (b, n, c)
    return t.transpose(1, 2).contiguous()


# ------------------------------------------------------------------------------------------------ classification
def _cls():
    ns = types.SimpleNamespace()

    def furthestsampling_cuda(b, n, m, xyz, temp, idx):
        _lib.call("rs_furthestsampling", b, n, m, _p(xyz), None, _p(temp), _p(idx), _s())       # first pick = row 0

    def gathering_forward_cuda(b, c, n, m, features, idx, out):
        rows = torch.empty((b, m, c), dtype=torch.float32, device=features.device)
        _lib.call("rs_gather_rows", b, n, m, c, _p(_cl(features)), _p(idx), _p(rows), _s())
        out.copy_(rows.transpose(1, 2))

    def gathering_backward_cuda(b, c, n, m, grad_out, idx, grad_features):
        acc = torch.zeros((b, n, c), dtype=torch.float32, device=grad_out.device)
        _lib.call("rs_gather_rows_backward", b, n, m, c, _p(_cl(grad_out)), _p(idx), _p(acc), _s())
        grad_features.add_(acc.transpose(1, 2))

    def ballquery_cuda(b, n, m, radius, nsample, new_xyz, xyz, idx):
        r2 = torch.tensor(float(radius) ** 2, dtype=torch.float32).item()
        _lib.call("rs_ballquery", b, n, m, r2, nsample, _p(new_xyz), _p(xyz), _p(idx), None, _s())

    def knnquery_cuda(b, n, m, nsample, xyz, new_xyz, idx, dist2):
        _lib.call("rs_knnquery", b, n, m, nsample, _p(xyz), _p(new_xyz), _p(idx), _p(dist2), _s())

    def grouping_forward_cuda(b, c, n, m, nsample, features, idx, out):
        rows = torch.empty((b, m, nsample, c), dtype=torch.float32, device=features.device)
        _lib.call("rs_group_rows", b, n, m, nsample, c, _p(_cl(features)), _p(idx), _p(rows), _s())
        out.copy_(rows.permute(0, 3, 1, 2))

    def grouping_backward_cuda(b, c, n, m, nsample, grad_out, idx, grad_features):
        acc = torch.zeros((b, n, c), dtype=torch.float32, device=grad_out.device)
        _lib.call("rs_group_rows_backward", b, n, m, nsample, c, _p(grad_out.permute(0, 2, 3, 1).contiguous()), _p(idx), _p(acc), _s())
        grad_features.add_(acc.transpose(1, 2))

    def grouping_int_forward_cuda(b, c, n, m, nsample, features, idx, out):
        # grouping_int_cuda_kernel.cu:33-49: the same gather on an int64 payload.  rs_group_rows moves 4-byte words without
        # arithmetic, so an int64 channel travels as two of them: (b, c, n) int64 -> channels-last (b, n, 2c) words -> gather
        words = features.to(torch.int64).permute(0, 2, 1).contiguous().view(torch.float32)          # (b, n, 2c)
        rows = torch.empty((b, m, nsample, 2 * c), dtype=torch.float32, device=features.device)
        _lib.call("rs_group_rows", b, n, m, nsample, 2 * c, _p(words), _p(idx), _p(rows), _s())
        out.copy_(rows.view(torch.int64).permute(0, 3, 1, 2))

    def nearestneighbor_cuda(b, n, m, unknown, known, dist2, idx):
        _lib.call("rs_three_nn", b, n, m, _p(unknown), _p(known), _p(dist2), _p(idx), _s())

    def interpolation_forward_cuda(b, c, m, n, features, idx, weight, out):
        rows = torch.empty((b, n, c), dtype=torch.float32, device=features.device)
        _lib.call("rs_three_interpolate", b, c, m, n, _p(_cl(features)), _p(idx), _p(weight), _p(rows), _s())
        out.copy_(rows.transpose(1, 2))

    def interpolation_backward_cuda(b, c, n, m, grad_out, idx, weight, grad_features):
        acc = torch.zeros((b, m, c), dtype=torch.float32, device=grad_out.device)
        _lib.call("rs_three_interpolate_backward", b, c, n, m, _p(_cl(grad_out)), _p(idx), _p(weight), _p(acc), _s())
        grad_features.add_(acc.transpose(1, 2))

    for f in (furthestsampling_cuda, gathering_forward_cuda, gathering_backward_cuda, ballquery_cuda, knnquery_cuda,
              grouping_forward_cuda, grouping_backward_cuda, grouping_int_forward_cuda, nearestneighbor_cuda,
              interpolation_forward_cuda, interpolation_backward_cuda):
        setattr(ns, f.__name__, f)
    # knnquery_heap_cuda_kernel.cu:55-90 keeps a max-heap and ends with heap_sort: the same ascending lists as knnquery_cuda
    # (its heap order is internal), so both operators bind the one kernel; nsample up to the reference's 200 / 100 is served
    # by csrc/knn_wide.hip
    ns.knnquery_heap_cuda = knnquery_cuda
    return ns


# ------------------------------------------------------------------------------------------------- segmentation
def _seg():
    ns = types.SimpleNamespace()

    def furthestsampling_cuda(b, n_max, xyz, offset, new_offset, tmp, idx):
        _lib.call("rs_furthestsampling_offset", b, int(n_max), _p(xyz), _p(offset), _p(new_offset), _p(tmp), _p(idx), _s())

    def knnquery_cuda(m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2):
        _lib.call("rs_knnquery_offset", m, nsample, _p(xyz), _p(new_xyz), _p(offset), _p(new_offset), offset.numel(),
                  _p(idx), _p(dist2), _s())

    def grouping_forward_cuda(m, nsample, c, input, idx, output):
        _lib.call("rs_group_rows", 1, input.shape[0], m, nsample, c, _p(input), _p(idx), _p(output), _s())

    def grouping_backward_cuda(m, nsample, c, grad_output, idx, grad_input):
        _lib.call("rs_group_rows_backward", 1, grad_input.shape[0], m, nsample, c, _p(grad_output.contiguous()), _p(idx),
                  _p(grad_input), _s())

    def interpolation_forward_cuda(n, c, k, input, idx, weight, output):
        if k != 3:
            raise _lib.RepSurfHipError("interpolation kernels are built for k = 3 neighbours")
        _lib.call("rs_three_interpolate", 1, c, input.shape[0], n, _p(input), _p(idx), _p(weight), _p(output), _s())

    def interpolation_backward_cuda(n, c, k, grad_output, idx, weight, grad_input):
        if k != 3:
            raise _lib.RepSurfHipError("interpolation kernels are built for k = 3 neighbours")
        _lib.call("rs_three_interpolate_backward", 1, c, n, grad_input.shape[0], _p(grad_output.contiguous()), _p(idx),
                  _p(weight), _p(grad_input), _s())

    def _out_of_scope(*a, **k):
        raise _lib.RepSurfHipError("Point-Transformer operators (subtraction / aggregation) are out of scope (DESIGN.md §9)")

    for f in (furthestsampling_cuda, knnquery_cuda, grouping_forward_cuda, grouping_backward_cuda,
              interpolation_forward_cuda, interpolation_backward_cuda):
        setattr(ns, f.__name__, f)
    for name in ("subtraction_forward_cuda", "subtraction_backward_cuda", "aggregation_forward_cuda", "aggregation_backward_cuda"):
        setattr(ns, name, _out_of_scope)
    return ns


def namespace(kind):
    """kind: "classification" | "segmentation" -> an object with the reference extension's function names."""
    if kind not in ("classification", "segmentation"):
        raise ValueError(kind)
    return _cls() if kind == "classification" else _seg()


def install(kind):
    """Register the namespace as the importable module `pointops_cuda` (what `import pointops_cuda` at
    pointops.py:7-8 finds) and return it."""
    ns = namespace(kind)
    mod = types.ModuleType("pointops_cuda")
    mod.__dict__.update(ns.__dict__)
    sys.modules["pointops_cuda"] = mod
    return mod
