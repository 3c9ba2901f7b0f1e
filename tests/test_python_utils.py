"""The tensor-op utilities the mirror modules re-export (`modules.recons_utils`, `modules.polar_utils` of both
sub-projects: ordinary torch code, off the hot path since the fused kernels replaced their call sites) against the
reference's own functions, imported by path.  CPU only; needs /root/reference (build container)."""
import os

import numpy as np
import pytest
import torch

from tests.util import ROOT, load_by_path

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs /root/reference (build container)")


def _pair(sub, name):
    ours = load_by_path(f"ours_{sub}_{name}", os.path.join(ROOT, "repsurf_amd", sub, "modules", name + ".py"))
    ref = load_by_path(f"ref_{sub}_{name}", os.path.join(REF, sub, "modules", name + ".py"))
    return ours, ref


def _tri(seed, shape):
    g = torch.Generator().manual_seed(seed)
    t = torch.rand(*shape, 3, 3, generator=g) * 2 - 1
    t[..., 0, :] = 0                                     # fan triangles share the origin
    return t


@pytest.mark.parametrize("sub", ["classification", "segmentation"])
def test_polar_utils_match_reference(sub):
    ours, ref = _pair(sub, "polar_utils")
    g = torch.Generator().manual_seed(0)
    xyz = torch.rand(4, 50, 7, 3, generator=g) * 2 - 1
    xyz[0, 0, 0] = 0                                     # rho == 0 -> theta 0
    for normalize in (True, False):
        assert torch.allclose(ours.xyz2sphere(xyz, normalize), ref.xyz2sphere(xyz, normalize), atol=1e-6, equal_nan=True)
        assert torch.allclose(ours.xyz2cylind(xyz, normalize), ref.xyz2cylind(xyz, normalize), atol=1e-6, equal_nan=True)


def test_classification_recons_utils_match_reference():
    ours, ref = _pair("classification", "recons_utils")
    tri = _tri(1, (3, 40, 8))
    tri[1, 5, 2, 2] = tri[1, 5, 2, 1]                    # a degenerate triangle -> NaN normal
    for is_group in (True, False):
        a, b = ours.cal_normal(tri, False, is_group), ref.cal_normal(tri, False, is_group)
        assert torch.allclose(a, b, atol=1e-6, equal_nan=True)
    torch.manual_seed(3)
    a = ours.cal_normal(tri, True, True)
    torch.manual_seed(3)
    b = ref.cal_normal(tri, True, True)
    assert torch.allclose(a, b, atol=1e-6, equal_nan=True)                    # same CPU-generator draw for the flip
    cen_a, cen_b = ours.cal_center(tri), ref.cal_center(tri)
    assert torch.allclose(cen_a, cen_b, atol=1e-7)
    assert torch.allclose(ours.cal_const(b.nan_to_num(), cen_b), ref.cal_const(b.nan_to_num(), cen_b), atol=1e-6)
    pos = ref.cal_const(b, cen_b)
    for x, y in zip(ours.check_nan_umb(b, cen_b, pos), ref.check_nan_umb(b, cen_b, pos)):
        assert torch.allclose(x, y, atol=1e-7, equal_nan=True)
    n3, c3 = b[:, :, 0], cen_b[:, :, 0]
    n3 = n3.clone()
    n3[0, 7] = float("nan")
    for x, y in zip(ours.check_nan(n3, c3), ref.check_nan(n3, c3)):
        assert torch.allclose(x, y, atol=1e-7, equal_nan=True)


def test_segmentation_recons_utils_match_reference():
    ours, ref = _pair("segmentation", "recons_utils")
    tri = _tri(2, (90, 9))                               # packed rows (N, G, 3, 3)
    tri[7, 3, 2] = tri[7, 3, 1]
    offset = torch.tensor([40, 90], dtype=torch.int32)
    for is_group in (True, False):
        assert torch.allclose(ours.cal_normal(tri, offset, False, is_group), ref.cal_normal(tri, offset, False, is_group),
                              atol=1e-6, equal_nan=True)
    np.random.seed(4)
    a = ours.cal_normal(tri, offset, True, True)
    np.random.seed(4)
    b = ref.cal_normal(tri, offset, True, True)          # same numpy-generator draw (recons_utils.py:29)
    assert torch.allclose(a, b, atol=1e-6, equal_nan=True)
    cen = ref.cal_center(tri)
    assert torch.allclose(ours.cal_center(tri), cen, atol=1e-7)
    pos = ref.cal_const(b, cen)
    assert torch.allclose(ours.cal_const(b, cen), pos, atol=1e-6, equal_nan=True)
    for x, y in zip(ours.check_nan_umb(b, cen, pos), ref.check_nan_umb(b, cen, pos)):
        assert torch.allclose(x, y, atol=1e-7, equal_nan=True)


def test_pipelined_step_identifies_offsets_by_position_and_refuses_other_boundaries():
    """repsurf_amd.graph: the packed batch's offsets are the LAST element of a list input (not 'any int32 tensor'); other
    int32 tensors are data; a batch with other cloud boundaries is refused."""
    import pytest
    from repsurf_amd import graph
    coord, lab32 = torch.zeros(10, 3), torch.arange(10, dtype=torch.int32)
    off = torch.tensor([4, 10], dtype=torch.int32)
    x = [coord, lab32, off]
    c = graph._clone_inputs(x)
    assert c[2] is off and c[1] is not lab32 and c[0] is not coord
    graph._copy_inputs(c, [coord + 1, lab32 + 1, torch.tensor([5, 10], dtype=torch.int32)])
    assert torch.equal(c[1], lab32 + 1) and torch.equal(c[0], coord + 1) and torch.equal(c[2], off)
    graph._check_offsets(c, [coord, lab32, torch.tensor([4, 10], dtype=torch.int32)])
    with pytest.raises(ValueError):
        graph._check_offsets(c, [coord, lab32, torch.tensor([5, 10], dtype=torch.int32)])
    with pytest.raises(TypeError):
        graph._clone_inputs([coord, off, coord])
    graph._check_offsets(coord, coord)            # plain tensor inputs: nothing to check


def test_bench_real_batch_and_step_count():
    """bench.py helpers (CPU side): `--data real` tiles the 4 scanned objects of tests/golden/geom_real.npz to the batch -- every copy
    a rotated, jittered version of its source (no two clouds equal), deterministic in the seed -- and the timed region replays
    max(--steps, ceil(min-seconds / step)) steps."""
    import argparse
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    pts, lab = bench.real_batch(7, 8, 1024, torch.device("cpu"))
    pts2, _ = bench.real_batch(7, 8, 1024, torch.device("cpu"))
    assert pts.shape == (8, 3, 1024) and lab.shape == (8,) and torch.equal(pts, pts2)
    flat = pts.reshape(8, -1)
    assert all(not torch.equal(flat[i], flat[j]) for i in range(8) for j in range(i))
    src = np.load(os.path.join(root, "tests", "golden", "geom_real.npz"))["xyz"]
    # a rotation about the y axis + 0.002 jitter: the heights (y) of copy 5 are those of scan 5 % 4 = 1 up to the jitter
    assert np.abs(pts[5, 1].numpy() - src[1, :, 1]).max() < 0.02

    class Dist:
        @staticmethod
        def max_over_ranks(v, device=None):
            return v
    calls = []
    args = argparse.Namespace(steps=20, min_seconds=0.5)
    n = bench.timed_step_count(args, lambda: calls.append(1), lambda: None, 1, None, Dist)
    assert n >= 20 and len(calls) == 10            # a fast step: thousands of replays to fill 0.5 s, probed with 10
    args = argparse.Namespace(steps=20, min_seconds=0.0)
    assert bench.timed_step_count(args, lambda: None, lambda: None, 1, None, Dist) == 20
