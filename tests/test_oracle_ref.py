"""The oracle restatements (oracle/geom_oracle.c) against the REFERENCE'S OWN pointops kernels (oracle/_ref: the
reference's *_cuda_kernel.cu files compiled unmodified as host code, oracle/Makefile.ref).  CPU only.

This is what un-pins "parity unpinned" for the CUDA-only kernels: packed kNN (heap), packed FPS (tree reduction),
three_nn / three_interpolate (+ backward), and — as cross-checks of the two semantics the classification API
carries — the CUDA ball query / kNN / FPS against the CPU-path oracle the classification kernels follow.

Tie contract: on rows whose candidate distances are pairwise distinct the oracle equals the reference
bit for bit.  Exactly equal distances: the reference's kNN leaves them in heap order (a function of the scan history,
not reproducible by a parallel algorithm) — there dist² stays bit-exact and every returned row has exactly the
returned distance; the oracle lists ties in ascending row order."""
import os

import numpy as np
import pytest

from oracle import geom_oracle as G
from oracle import ref_pointops as R
from tests.util import GOLDEN, cloud

pytestmark = pytest.mark.skipif(not (R.available("seg") and R.available("cls")),
                                reason="oracle/_ref not built (needs /root/reference: make -f oracle/Makefile.ref)")


def packed_case(seed, sizes, kind="uniform"):
    pts = np.concatenate([cloud(seed + i, 1, max(n, 2), kind)[0][:n] for i, n in enumerate(sizes)]).astype(np.float32)
    return np.ascontiguousarray(pts), np.cumsum(sizes).astype(np.int32)


def direct_d2(q, p):
    d = q.astype(np.float32) - p.astype(np.float32)
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


def check_knn_rows(ref_idx, ref_d2, got_idx, got_d2, xyz, new_xyz):
    """bit-exact dist²; idx exact on tie-free rows; on tie rows every idx has its returned distance."""
    assert np.array_equal(ref_d2, got_d2)
    k = ref_d2.shape[-1]
    tie = (np.diff(ref_d2, axis=-1) == 0).any(-1) if k > 1 else np.zeros(ref_d2.shape[:-1], bool)
    assert np.array_equal(ref_idx[~tie], got_idx[~tie])
    for idx in (ref_idx, got_idx):
        rows = np.nonzero(tie)
        d = direct_d2(new_xyz[rows][:, None, :], xyz[idx[rows]])
        real = ref_d2[rows] < 1e9                       # 1e10 padding rows keep idx = start
        assert np.array_equal(d[real], ref_d2[rows][real])
    return int(tie.sum())


KNN_CASES = [  # (seed, sizes, kind, k, stride of the queries)
    (0, [300, 512, 217], "uniform", 9, 1), (1, [300, 512, 217], "uniform", 32, 4), (2, [64, 1, 40], "uniform", 3, 1),
    (3, [5, 700], "uniform", 9, 1), (4, [20, 33], "uniform", 32, 1), (5, [1024], "uniform", 16, 2),
    (6, [343, 200], "grid", 9, 1), (7, [400, 100], "dup", 9, 1), (8, [256, 256], "clustered", 32, 4),
    (9, [600], "dup", 3, 3), (10, [2048, 1536], "uniform", 32, 4), (11, [2048, 1536], "uniform", 3, 1),
]


@pytest.mark.parametrize("seed,sizes,kind,k,stride", KNN_CASES)
def test_packed_knn_restatement_equals_reference_kernel(seed, sizes, kind, k, stride):
    """oracle_knn_offset == segmentation/modules/pointops/src/knnquery/knnquery_cuda_kernel.cu:65-108"""
    xyz, off = packed_case(seed, sizes, kind)
    if stride == 1:
        q, qoff = xyz, off
    else:
        keep = np.concatenate([np.arange(a, b)[::stride] for a, b in zip(np.r_[0, off[:-1]], off)])
        q = np.ascontiguousarray(xyz[keep])
        qoff = np.cumsum([len(np.arange(a, b)[::stride]) for a, b in zip(np.r_[0, off[:-1]], off)]).astype(np.int32)
    ri, rd = R.seg_knn(k, xyz, q, off, qoff)
    oi, od = G.knn_offset(k, xyz, q, off, qoff)
    ties = check_knn_rows(ri, rd, oi, od, xyz, q)
    if kind == "uniform":
        assert ties == 0 or min(sizes) < k               # clouds smaller than k pad with 1e10 (equal "distances")


FPS_CASES = [(0, [300, 512, 217], 4), (1, [1500], 4), (2, [64, 1, 40], 2), (3, [1024, 1024], 8), (4, [2048, 1536], 4),
             (5, [33, 700, 5], 3), (6, [4096], 16)]


@pytest.mark.parametrize("seed,sizes,stride", FPS_CASES)
def test_packed_fps_restatement_equals_reference_kernel(seed, sizes, stride):
    """oracle_fps_offset == segmentation/modules/pointops/src/sampling/sampling_cuda_kernel.cu:14-129, the 10-level
    shared-memory tree reduction executed phase by phase (cooperative fibers)."""
    xyz, off = packed_case(seed, sizes)
    noff = np.cumsum([max(n // stride, 1) for n in sizes]).astype(np.int32)
    assert np.array_equal(R.seg_fps(xyz, off, noff), G.fps_offset(xyz, off, noff))


def test_packed_fps_tie_rule_of_the_reference_kernel():
    """Exact distance ties (lattices, duplicated points): the kernel's strided per-thread scan + tree reduction picks
    the candidate with the lowest (row - start) mod block_size, then the lowest row (sampling_cuda_kernel.cu:44-58,
    __update :7-12) where block_size = opt_n_threads(largest cloud); the oracle applies the same key."""
    for seed, sizes, kind in ((0, [343], "grid"), (1, [500, 300], "dup"), (2, [1331, 100], "grid"), (3, [3000], "dup")):
        xyz, off = packed_case(seed, sizes, kind)
        noff = np.cumsum([n // 4 for n in sizes]).astype(np.int32)
        assert np.array_equal(R.seg_fps(xyz, off, noff), G.fps_offset(xyz, off, noff)), (seed, sizes, kind)


def test_fixture_kernels_match_oracle():
    """The raw kernel outputs stored in seg_geom.npz (reference Functions over oracle/_ref) == the oracle."""
    g = np.load(os.path.join(GOLDEN, "seg_geom.npz"))
    coord, off = g["coord"], g["offset"]
    assert np.array_equal(G.fps_offset(coord, off, g["fps_new_offset"]), g["fps_idx"])
    for k in (9, 32):
        oi, od = G.knn_offset(k, coord, coord, off, off)
        assert np.array_equal(oi, g[f"knn{k}_idx"])
        ref = g[f"knn{k}_dist"]
        assert (np.abs(np.sqrt(od) - ref) <= np.spacing(ref)).all()      # torch.sqrt (MKL VML) vs sqrtf: last bit


@pytest.mark.parametrize("seed,b,n,m,kind", [(0, 2, 256, 100, "uniform"), (1, 1, 1000, 250, "uniform"),
                                             (2, 3, 64, 64, "clustered"), (3, 2, 300, 5, "uniform"),
                                             (4, 2, 200, 216, "grid"), (5, 1, 128, 2, "uniform")])
def test_three_nn_and_interpolate_equal_reference_kernels(seed, b, n, m, kind):
    """oracle_three_nn / oracle_three_interpolate == classification/modules/pointops/src/interpolation/
    interpolation_cuda_kernel.cu:134-195 (+ the backward kernel :90-114, order-independent here: one thread)."""
    unknown = cloud(seed, b, n, "uniform")
    known = cloud(seed + 50, b, m, kind)
    rd, ri = R.cls_three_nn(unknown, known)
    od, oi = G.three_nn(unknown, known)
    if kind == "grid":
        assert np.array_equal(rd, od)
        tie = (np.diff(rd, axis=-1) == 0).any(-1)
        assert np.array_equal(ri[~tie], oi[~tie])
    else:
        assert np.array_equal(rd, od) and np.array_equal(ri, oi)
    rng = np.random.RandomState(seed)
    c = 7
    feats = rng.randn(b, m, c).astype(np.float32)                        # ours: channels-last
    w = rng.rand(b, n, 3).astype(np.float32)
    w /= w.sum(-1, keepdims=True)
    ref = R.cls_three_interpolate(np.ascontiguousarray(feats.transpose(0, 2, 1)), ri, w)     # (b,c,n)
    got = G.three_interpolate(feats, ri, w)                                                    # (b,n,c)
    assert np.array_equal(ref.transpose(0, 2, 1), got)
    gout = rng.randn(b, n, c).astype(np.float32)
    gref = R.cls_three_interpolate_backward(np.ascontiguousarray(gout.transpose(0, 2, 1)), ri, w, m)   # (b,c,m)
    acc = np.zeros((b, m, c), np.float64)
    for bi in range(b):
        for j in range(3):
            np.add.at(acc[bi], ri[bi, :, j], gout[bi].astype(np.float64) * w[bi, :, j:j + 1])
    assert np.abs(gref.transpose(0, 2, 1) - acc).max() <= 1e-5 * max(1.0, np.abs(acc).max())


@pytest.mark.parametrize("seed", range(4))
def test_cuda_ballquery_agrees_with_cpu_path_oracle(seed):
    """The classification kernels follow the CPU path (expanded distances, `not d > r²`); the reference's CUDA kernel
    (ballquery_cuda_kernel.cu:47-80) uses direct differences and `d < r*r`.  SURVEY §8a row 4 predicts identical
    neighbour lists away from |d² - r²| ≲ 5e-7: count the rows that differ."""
    xyz = cloud(seed, 4, 1024)
    centres = xyz[:, :256]
    diff = 0
    for r, ns in ((0.2, 32), (0.4, 64), (0.1, 24)):
        a = R.cls_ballquery(r, ns, xyz, centres)
        b = G.ballquery(r, ns, xyz, centres)
        diff += int((a != b).any(-1).sum())
    assert diff <= 2, diff


@pytest.mark.parametrize("seed", range(3))
def test_cuda_knn_and_fps_agree_with_cpu_path_oracle(seed):
    """kNN: same neighbour SET as the CPU-path oracle (order may differ where the two distance formulas round
    differently: SURVEY §8a row 5).  FPS: the CUDA kernel starts at row 0 (sampling_cuda_kernel.cu:72-74); with
    start = 0 the CPU-path oracle picks the same rows on tie-free clouds."""
    xyz = cloud(seed, 2, 1024)
    ri, _ = R.cls_knn(9, xyz, xyz)
    oi = G.knn(9, xyz, xyz)
    rows = (np.sort(ri, -1) != np.sort(oi, -1)).any(-1).sum()
    assert rows <= 2, rows
    assert np.array_equal(R.cls_fps(xyz, 256), G.fps(xyz, 256, np.zeros(2, np.int32)))
