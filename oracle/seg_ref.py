"""CPU restatement of the RepSurf-U segmentation step (forward + cross-entropy + backward).
TEST INFRASTRUCTURE ONLY (checker for repsurf_amd.segmentation).

Follows the reference's segmentation path function by function:
  Model.forward                 segmentation/models/repsurf/repsurf_umb_ssg.py:43-66
  UmbrellaSurfaceConstructor    segmentation/modules/repsurface_utils.py:287-329
  sample_and_group              segmentation/modules/repsurface_utils.py:15-51
  SurfaceAbstractionCD          segmentation/modules/repsurface_utils.py:176-230
  SurfaceFeaturePropagationCD   segmentation/modules/repsurface_utils.py:233-284
Geometry comes from oracle/geom_oracle.c (packed-batch FPS and kNN restate the reference's CUDA kernels and are pinned
against those kernels executed as host code, oracle/_ref + tests/test_oracle_ref.py; everything downstream of them is
pinned against the reference's own torch code run over its own kernels on CPU, tests/golden/make_golden_seg.py ->
tests/golden/seg_model.npz).
Dense part: PyTorch fp32 CPU ops, BatchNorm statistics in fp64 (see oracle/torch_ref.py:_bn_train).

Parameters come from a state_dict with the reference's key names.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import geom_oracle as G
from .torch_ref import _bn_train

SA = [dict(stride=4, nsample=32), dict(stride=4, nsample=32), dict(stride=4, nsample=32), dict(stride=4, nsample=32)]


def strided_offset(offset, stride):
    """new_offset of sample_and_group (:17-22): running sum of (cloud length // stride)."""
    offset = np.asarray(offset, np.int64)
    lens = np.diff(np.concatenate([[0], offset]))
    return np.cumsum(lens // stride).astype(np.int32)


def _lin(x, p, key):
    w = p[key + ".weight"]
    return F.linear(x, w.view(w.shape[0], w.shape[1]), p.get(key + ".bias"))


def _rows(t, idx):
    """t (N,C) torch, idx (...) numpy int -> (..., C), differentiable"""
    return t[torch.from_numpy(np.ascontiguousarray(idx).reshape(-1).astype(np.int64))].view(*idx.shape, t.shape[1])


def umbrella_rows(coord, offset, inv_sign, k=9, rotate=True):
    """-> feat (N, k, 10) numpy, knn idx (N,k), near_tie (N,)"""
    idx, _ = G.knn_offset(k, coord, coord, offset, offset)
    feat, tie = G.umbrella_fan_offset(coord, coord, idx, offset, inv_sign, rotate)
    return feat, idx, tie


def sample_and_group(stride, nsample, center, normal, feature, offset, return_polar=False):
    """center (N,3) numpy; normal (N,10) / feature (N,C) torch -> new_center numpy, new_normal torch,
    rows (M*nsample, 3(+3)+10+C) torch, new_offset, (fps idx, knn idx)   (:15-51)"""
    new_offset = strided_offset(offset, stride)
    fidx = G.fps_offset(center, offset, new_offset)
    new_center = center[fidx]
    new_normal = _rows(normal, fidx)
    gidx, _ = G.knn_offset(nsample, center, new_center, offset, new_offset)
    m = new_center.shape[0]
    g = torch.from_numpy(center[gidx.reshape(-1)].reshape(m, nsample, 3) - new_center[:, None, :]).to(normal.dtype)
    parts = [g]
    if return_polar:
        parts.append(xyz2sphere(g))
    parts.append(_rows(normal, gidx))
    if feature is not None:
        parts.append(_rows(feature, gidx))
    rows = torch.cat(parts, dim=-1).reshape(m * nsample, -1)
    return new_center, new_normal, rows, new_offset, (fidx, gidx)


def xyz2sphere(xyz):
    """segmentation/modules/polar_utils.py:10-31 (torch ops, same as the reference)."""
    rho = torch.sqrt(torch.sum(torch.pow(xyz, 2), dim=-1, keepdim=True))
    rho = torch.clamp(rho, min=0)
    theta = torch.acos(xyz[..., 2, None] / rho)
    phi = torch.atan2(xyz[..., 1, None], xyz[..., 0, None])
    theta = torch.where(rho == 0, torch.zeros_like(theta), theta)
    return torch.cat([rho, theta / np.pi, phi / (2 * np.pi) + .5], dim=-1)


def step(state, coord, feat, offset, label=None, inv_sign=None, k=9, return_polar=False, want_grads=True,
         dtype=torch.float32):
    """One training step on CPU.  coord (N,3), feat (N,Cin-3) float32 numpy, offset (B,) int32 running ends,
    inv_sign (B,) +-1 or None, label (N,) int.  Returns logits, loss, stage outputs and {name: grad}.
    dtype=torch.float64: the TRUTH leg of the three-way parity tests -- the same indices, fan features and interpolation
    weights (fp32 outputs of the geometry oracle, widened exactly), every dense operation of the network in float64."""
    p = {k_: v.detach().clone().to(dtype).requires_grad_(v.dtype.is_floating_point and want_grads)
         for k_, v in state.items() if "running" not in k_ and "num_batches" not in k_}
    coord = np.ascontiguousarray(coord, np.float32)
    offset = np.ascontiguousarray(offset, np.int32)
    out = {}
    pos_ch = 6 if return_polar else 3
    # --- umbrella surface constructor (:305-329)
    ufeat, _, tie = umbrella_rows(coord, offset, inv_sign, k)
    out["umb_feat"], out["near_tie"] = ufeat, tie
    n = coord.shape[0]
    h = torch.from_numpy(ufeat.reshape(n * k, 10)).to(dtype)
    h = F.relu(_bn_train(_lin(h, p, "surface_constructor.mlps.0"), p["surface_constructor.mlps.1.weight"],
                         p["surface_constructor.mlps.1.bias"]))
    normal = _lin(h, p, "surface_constructor.mlps.3").view(n, k, -1).sum(dim=1)          # (N,10)
    out["normal"] = normal
    feature = torch.cat([torch.from_numpy(coord), torch.from_numpy(np.ascontiguousarray(feat, np.float32))], 1).to(dtype)
    levels = [(coord, feature, offset)]
    center = coord
    for si, st in enumerate(SA, 1):
        pre = f"sa{si}"
        center, normal, rows, offset, (fidx, gidx) = sample_and_group(st["stride"], st["nsample"], center, normal,
                                                                       feature, offset, return_polar)
        out[pre + "_fps"], out[pre + "_knn"] = fidx, gidx
        loc = _bn_train(_lin(rows[:, :pos_ch], p, pre + ".mlp_l0"), p[pre + ".bn_l0.weight"], p[pre + ".bn_l0.bias"])
        ft = _bn_train(_lin(rows[:, pos_ch:], p, pre + ".mlp_f0"), p[pre + ".bn_f0.weight"], p[pre + ".bn_f0.bias"])
        h = F.relu(loc + ft)
        i = 0
        while f"{pre}.mlp_convs.{i}.weight" in p:
            h = F.relu(_bn_train(_lin(h, p, f"{pre}.mlp_convs.{i}"), p[f"{pre}.mlp_bns.{i}.weight"],
                                 p[f"{pre}.mlp_bns.{i}.bias"]))
            i += 1
        feature = h.view(center.shape[0], st["nsample"], -1).max(dim=1)[0]
        out[pre + "_feat"] = feature
        levels.append((center, feature, offset))

    def fp(pre, lvl1, lvl2, skip):
        xyz1, pts1, off1 = lvl1
        xyz2, pts2, off2 = lvl2
        idx, d2 = G.knn_offset(3, xyz2, xyz1, off2, off1)                               # (:261)
        w = torch.from_numpy(G.interp_weights(d2)).to(dtype)
        pts2 = _bn_train(_lin(pts2, p, pre + ".mlp_f0"), p[pre + ".norm_f0.weight"], p[pre + ".norm_f0.bias"])
        interp = torch.zeros(xyz1.shape[0], pts2.shape[1], dtype=dtype)
        for i in range(3):
            interp = interp + _rows(pts2, idx[:, i]) * w[:, i].unsqueeze(-1)
        if skip:
            s = _bn_train(_lin(pts1, p, pre + ".mlp_s0"), p[pre + ".norm_s0.weight"], p[pre + ".norm_s0.bias"])
            h = F.relu(interp + s)
        else:
            h = F.relu(interp)
        i = 0
        while f"{pre}.mlp_convs.{i}.weight" in p:
            h = F.relu(_bn_train(_lin(h, p, f"{pre}.mlp_convs.{i}"), p[f"{pre}.mlp_bns.{i}.weight"],
                                 p[f"{pre}.mlp_bns.{i}.bias"]))
            i += 1
        return h

    f3 = fp("fp4", levels[3], levels[4], True)
    f2 = fp("fp3", levels[2], (levels[3][0], f3, levels[3][2]), True)
    f1 = fp("fp2", levels[1], (levels[2][0], f2, levels[2][2]), True)
    f0 = fp("fp1", (levels[0][0], None, levels[0][2]), (levels[1][0], f1, levels[1][2]), False)
    out["fp1_feat"] = f0
    h = F.relu(_bn_train(_lin(f0, p, "classifier.0"), p["classifier.1.weight"], p["classifier.1.bias"]))
    logits = _lin(h, p, "classifier.4")                                                  # dropout disabled
    out["logits"] = logits
    if label is not None:
        loss = F.cross_entropy(logits, torch.as_tensor(label, dtype=torch.long))
        out["loss"] = loss
        if want_grads:
            loss.backward()
            out["grads"] = {k_: v.grad for k_, v in p.items() if v.grad is not None}
    return out
