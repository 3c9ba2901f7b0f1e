"""The native boundary, EXECUTED: the reference's own operator files (`modules/pointops/functions/pointops.py` of both
sub-projects, loaded unmodified by path) with `import pointops_cuda` resolving to repsurf_amd.pointops_cuda, i.e. to
librepsurf_hip.so through its C ABI.  Expected values: the fixtures the SAME files produced on CPU over the reference's
own kernels (oracle/_ref): tests/golden/cls_pointops.npz, seg_geom.npz, seg_sector.npz."""
import os

import numpy as np
import pytest
import torch

from tests.util import GOLDEN, load_by_path, parity_report, staged_reference_file

CLS = staged_reference_file("classification", "modules/pointops/functions/pointops.py")
SEG = staged_reference_file("segmentation", "modules/pointops/functions/pointops.py")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(CLS is None or SEG is None, reason="reference pointops.py files not staged")]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_reference_classification_operators_over_the_hip_library():
    import repsurf_amd.pointops_cuda as pc
    pc.install("classification")
    P = load_by_path("ref_cls_pointops", CLS)
    fx = np.load(os.path.join(GOLDEN, "cls_pointops.npz"))
    xyz, feats, new_xyz = dev(fx["xyz"]), dev(fx["feats"]), dev(fx["new_xyz"])
    fps = P.furthestsampling(xyz, fx["fps"].shape[1])
    assert np.array_equal(fps.cpu().numpy(), fx["fps"])
    f1 = feats.clone().requires_grad_()
    gath = P.gathering(f1, fps)
    assert np.array_equal(gath.detach().cpu().numpy(), fx["gathering"])
    (gath * dev(fx["gathering_w"])).sum().backward()
    assert np.abs(f1.grad.cpu().numpy() - fx["gathering_grad"]).max() <= 1e-6
    rows = sum(int((P.ballquery(r, ns, xyz, new_xyz).cpu().numpy() != fx[f"ball_{ns}"]).any(-1).sum()) for r, ns in ((0.2, 16), (0.4, 32)))
    knn = P.knnquery(9, xyz, new_xyz)
    knn = (knn[0] if isinstance(knn, tuple) else knn).cpu().numpy()
    knn_rows = int((np.sort(knn, -1) != np.sort(fx["knn9"], -1)).any(-1).sum())
    parity_report("reference_cls_pointops_over_hip", ballquery_rows_differing=rows, knn_rows_with_other_set=knn_rows)
    assert rows <= 1 and knn_rows <= 1
    f2 = feats.clone().requires_grad_()
    grp = P.grouping(f2, dev(fx["ball_16"]))
    assert np.array_equal(grp.detach().cpu().numpy(), fx["grouping"])
    (grp * dev(fx["grouping_w"])).sum().backward()
    assert np.abs(f2.grad.cpu().numpy() - fx["grouping_grad"]).max() <= 2e-5
    # grouping_int (int64 payload, grouping_int_cuda_kernel.cu:33-49) and the operator's full kNN width (knnquery_cuda_kernel.cu:21-22)
    lab = torch.arange(feats.shape[0] * 2 * feats.shape[2], device="cuda").view(feats.shape[0], 2, feats.shape[2]) * 7 - 3 + (1 << 40)
    gi = P.grouping_int(lab, dev(fx["ball_16"]))
    want = torch.gather(lab, 2, dev(fx["ball_16"]).long().reshape(lab.shape[0], 1, -1).expand(-1, 2, -1)).view(gi.shape)
    assert gi.dtype == torch.int64 and torch.equal(gi, want)
    wide = P.knnquery(200, xyz, new_xyz)
    wide = (wide[0] if isinstance(wide, tuple) else wide)
    assert wide.shape[-1] == 200 and torch.equal(wide[..., :9].cpu(), torch.from_numpy(knn))
    dist, nidx = P.nearestneighbor(xyz, new_xyz)
    assert np.array_equal(nidx.cpu().numpy(), fx["nn_idx"])
    assert (np.abs(dist.cpu().numpy() - fx["nn_dist"]) <= np.spacing(fx["nn_dist"])).all()
    kf = dev(fx["interp_feats"]).requires_grad_()
    itp = P.interpolation(kf, nidx, dev(fx["nn_weight"]))
    assert np.abs(itp.detach().cpu().numpy() - fx["interp"]).max() <= 1e-6
    (itp * dev(fx["interp_w"])).sum().backward()
    assert np.abs(kf.grad.cpu().numpy() - fx["interp_grad"]).max() <= 2e-5


def test_reference_segmentation_operators_over_the_hip_library():
    import repsurf_amd.pointops_cuda as pc
    pc.install("segmentation")
    P = load_by_path("ref_seg_pointops", SEG)
    g = np.load(os.path.join(GOLDEN, "seg_geom.npz"))
    coord, off = dev(g["coord"]), dev(g["offset"])
    assert np.array_equal(P.furthestsampling(coord, off, dev(g["fps_new_offset"])).cpu().numpy(), g["fps_idx"])
    for k in (9, 32):
        idx, dist = P.knnquery(k, coord, coord, off, off)
        assert np.array_equal(idx.cpu().numpy(), g[f"knn{k}_idx"])
        assert (np.abs(dist.cpu().numpy() - g[f"knn{k}_dist"]) <= np.spacing(g[f"knn{k}_dist"])).all()
    # the reference's OWN sectorized_fps host loop (pointops.py:52-108) over this library's FPS kernel
    s = np.load(os.path.join(GOLDEN, "seg_sector.npz"))
    for ns in (1, 2, 4):
        got = P.sectorized_fps(dev(s["coord"]), dev(s["offset"]), dev(s["new_offset"]), ns, int(s["min_points"])).cpu().numpy()
        assert np.array_equal(got, s[f"idx_s{ns}"]), ns
    # differentiable operators: grouping and the 3-NN interpolation Function against plain tensor code
    rng = np.random.RandomState(0)
    feat = dev(rng.randn(coord.shape[0], 6).astype(np.float32)).requires_grad_()
    idx, _ = P.knnquery(9, coord, coord, off, off)
    grp = P.grouping(feat, idx)
    assert torch.equal(grp.detach(), feat.detach()[idx.long()])
    w = dev(rng.randn(*grp.shape).astype(np.float32))
    (grp * w).sum().backward()
    ref = torch.zeros_like(feat).index_add_(0, idx.reshape(-1).long(), w.reshape(-1, 6))
    assert (feat.grad - ref).abs().max().item() <= 1e-5
    nc, no = dev(g["sg_x_center"]), dev(g["sg_x_offset"])
    cf = dev(rng.randn(nc.shape[0], 5).astype(np.float32)).requires_grad_()
    out = P.interpolation2(nc, coord, cf, no, off)
    plain = P.interpolation(nc, coord, cf.detach(), no, off)
    assert (out.detach() - plain).abs().max().item() <= 1e-6
    out.sum().backward()
    assert torch.isfinite(cf.grad).all() and abs(cf.grad.sum().item() - coord.shape[0] * 5) <= 1e-2 * coord.shape[0]


@pytest.mark.parametrize("sizes", [[60000, 30000], [20000, 120000]])
def test_sectorized_fps_of_large_clouds_equals_the_reference_host_loop(sizes):
    """Clouds of the reference's S3DIS size (up to 80 000 points, 4 sectors from 10 000 points: segmentation/modules/pointops/functions/
    pointops.py:52-108).  Device path (repsurf_amd.ops.sectorized_fps): the host bounds a sector by its whole cloud, so the launch pairs
    fps_lds_kernel (coordinates in registers, running distance in LDS, <= 24 576 positions) with fps_global_kernel and the largest
    sector -- known on the device only -- decides which of the two works.  Against the reference's OWN host loop over this library's
    FPS entry point, which is handed each launch's exact largest sector: same rows, bit for bit, whichever kernels either side took
    ([60000, 30000]: sectors of ~15 000 / 7 500 rows; [20000, 120000]: a 30 000-row sector pushes the device path to the global kernel)."""
    import repsurf_amd.pointops_cuda as pc
    from repsurf_amd import ops
    pc.install("segmentation")
    P = load_by_path("ref_seg_pointops", SEG)
    r = np.random.RandomState(len(sizes) + sizes[0])
    n = sum(sizes)
    coord = (r.rand(n, 3) * 2 - 1).astype(np.float32)
    offset = np.cumsum(sizes).astype(np.int32)
    new_offset = np.cumsum([s // 4 for s in sizes]).astype(np.int32)
    want = P.sectorized_fps(dev(coord), dev(offset), dev(new_offset), 4, 10000).cpu().numpy()
    got = ops.sectorized_fps(dev(coord), dev(offset), dev(new_offset), 4, 10000).cpu().numpy()
    assert np.array_equal(got, want)
