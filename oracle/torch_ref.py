"""CPU restatement of the RepSurf-U classifier step (forward + SmoothClsLoss + backward).
TEST INFRASTRUCTURE ONLY (checker for repsurf_amd, and bench.py's `cpu_baseline` leg).

Follows the reference's classification path function by function:
  Model.forward                 classification/models/repsurf/repsurf_ssg_umb.py:43-57
  UmbrellaSurfaceConstructor    classification/modules/repsurface_utils.py:276-307
  SurfaceAbstractionCD          classification/modules/repsurface_utils.py:218-249
  sample_and_group(_all)        classification/modules/repsurface_utils.py:15-88
  SmoothClsLoss                 classification/util/utils.py:55-69
Geometry (FPS, ball query, kNN, umbrella features, grouping) comes from oracle/geom_oracle.c
— the exact-arithmetic restatement pinned bit-for-bit against the reference — and the dense
part (1x1 convs as row GEMMs, BatchNorm with batch statistics, ReLU, max) runs as PyTorch fp32
CPU ops, the same arithmetic library the reference itself runs on.  Pinned against the
reference's outputs in tests/golden/model_b4.npz (tests/test_oracle_golden.py).

Parameters are taken from a state_dict with the reference's key names, so the same weights can
be given to the GPU model and to this oracle.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import geom_oracle as G

STAGES = {
    "repsurf_ssg_umb": [dict(npoint=512, radius=0.2, nsample=32), dict(npoint=128, radius=0.4, nsample=64), dict()],
    "repsurf_ssg_umb_2x": [dict(npoint=512, radius=0.1, nsample=24), dict(npoint=128, radius=0.2, nsample=24),
                           dict(npoint=32, radius=0.4, nsample=24), dict()],
}


def _bn_train(y, w, b, eps=1e-5):
    """Training-mode BatchNorm over rows.  nn.BatchNorm2d on the reference's (B,C,nsample,npoint)
    layout reduces with a cascade sum that is accurate to ~2e-6; F.batch_norm on a 2-D (rows, C)
    tensor uses a plain running sum that drifts by 1e-4 at 5e5 rows (probed), so the statistics
    are restated here in float64 and applied as y*alpha + beta like the CPU kernel does."""
    mean = y.mean(0, dtype=torch.float64)                       # fp64 accumulation, no fp64 copy of y
    var = ((y * y).mean(0, dtype=torch.float64) - mean * mean).clamp_min(0)
    alpha = (w.double() / torch.sqrt(var + eps)).to(y.dtype)
    beta = (b.double() - mean * (w.double() / torch.sqrt(var + eps))).to(y.dtype)
    return y * alpha + beta


def _conv(x, p, key):
    w = p[key + ".weight"]
    return F.linear(x, w.view(w.shape[0], w.shape[1]), p.get(key + ".bias"))


def _gather(t, idx):
    """t (B,N,C) torch, idx (B,...) numpy int -> (B,...,C) differentiable"""
    b = t.shape[0]
    flat = torch.from_numpy(idx.reshape(b, -1).astype(np.int64))
    out = torch.gather(t, 1, flat.unsqueeze(-1).expand(-1, -1, t.shape[2]))
    return out.view(*idx.shape, t.shape[2])


def _bn_train_fp32(y, w, b, eps=1e-5):
    """What the reference itself executes (nn.BatchNorm2d -> the fp32 CPU batch_norm kernel): the TIMED leg of
    bench.py's cpu_baseline uses it, so that the port is not slower than the thing it stands in for (the fp64
    statistics above cost 0.9 s per 524 288 x 64 layer on 8 threads, the fp32 kernel 0.1 s)."""
    return F.batch_norm(y, None, None, w, b, True, 0.1, eps)


def step(state, xyz, label=None, inv_sign=None, fps_starts=None, arch="repsurf_ssg_umb", k=9,
         want_grads=True, timed=False):
    """One training step on CPU.  state: {name: tensor} (reference key names); xyz (B,N,3) float32
    numpy; inv_sign (B,) +-1 or None; fps_starts: list of (B,) int arrays, one per sampling stage.
    Returns dict with logits, loss, per-stage outputs and {name: grad}.
    timed=True: BatchNorm through torch's fp32 kernel (the reference's own arithmetic and cost) -- for timing only,
    parity checks use the default."""
    _bn_train = _bn_train_fp32 if timed else globals()["_bn_train"]
    p = {k_: v.detach().clone().float().requires_grad_(v.dtype.is_floating_point and want_grads)
         for k_, v in state.items() if "running" not in k_ and "num_batches" not in k_}
    xyz = np.ascontiguousarray(xyz, np.float32)
    b, n, _ = xyz.shape
    out = {}
    # --- umbrella surface constructor (repsurface_utils.py:276-307)
    feat, _, near_tie = G.umbrella(xyz, k, inv_sign)
    out["umb_feat"], out["near_tie"] = feat, near_tie
    g = k - 1
    h = torch.from_numpy(feat.reshape(b * n * g, 10))
    h = F.relu(_bn_train(_conv(h, p, "surface_constructor.mlps.0"), p["surface_constructor.mlps.1.weight"], p["surface_constructor.mlps.1.bias"]))
    h = F.relu(_bn_train(_conv(h, p, "surface_constructor.mlps.3"), p["surface_constructor.mlps.4.weight"], p["surface_constructor.mlps.4.bias"]))
    normal = _conv(h, p, "surface_constructor.mlps.6").view(b, n, g, 10).sum(dim=2)      # (B,N,10)
    out["normal"] = normal
    center = torch.from_numpy(xyz)
    feature = None
    stages = STAGES[arch]
    for si, st in enumerate(stages, 1):
        pre = f"sa{si}"
        c_np = center.numpy()
        if st:   # sample_and_group (repsurface_utils.py:15-59)
            start = None if fps_starts is None else fps_starts[si - 1]
            fidx = G.fps(c_np, st["npoint"], start)
            new_center = _gather(center, fidx)
            new_normal = _gather(normal, fidx)
            bidx = G.ballquery(st["radius"], st["nsample"], c_np, new_center.numpy())
            out[pre + "_fps"], out[pre + "_ball"] = fidx, bidx
            pos = G.group_features(c_np, new_center.numpy(), np.zeros((b, c_np.shape[1], 0), np.float32), None, bidx, polar=True)
            parts = [torch.from_numpy(pos), _gather(normal, bidx).reshape(pos.shape[0], -1)]
            if feature is not None:
                parts.append(_gather(feature, bidx).reshape(pos.shape[0], -1))
            ns, s = st["nsample"], st["npoint"]
        else:    # sample_and_group_all (repsurface_utils.py:62-88)
            npts = c_np.shape[1]
            pos = G.group_all_features(c_np, np.zeros((b, npts, 0), np.float32), None, polar=True)
            parts = [torch.from_numpy(pos), normal.reshape(b * npts, -1), feature.reshape(b * npts, -1)]
            new_center = torch.zeros(b, 1, 3)
            new_normal = new_center
            ns, s = npts, 1
        x = torch.cat(parts, dim=1)
        loc = _bn_train(_conv(x[:, :6], p, pre + ".mlp_l0"), p[pre + ".bn_l0.weight"], p[pre + ".bn_l0.bias"])
        ft = _bn_train(_conv(x[:, 6:], p, pre + ".mlp_f0"), p[pre + ".bn_f0.weight"], p[pre + ".bn_f0.bias"])
        h = F.relu(loc + ft)
        i = 0
        while f"{pre}.mlp_convs.{i}.weight" in p:
            h = F.relu(_bn_train(_conv(h, p, f"{pre}.mlp_convs.{i}"), p[f"{pre}.mlp_bns.{i}.weight"], p[f"{pre}.mlp_bns.{i}.bias"]))
            i += 1
        feature = h.view(b * s, ns, -1).max(dim=1)[0].view(b, s, -1)
        center, normal = new_center, new_normal
        out[pre + "_feat"] = feature
    h = feature.reshape(b, -1)
    h = F.relu(_bn_train(F.linear(h, p["classfier.0.weight"], p["classfier.0.bias"]), p["classfier.1.weight"], p["classfier.1.bias"]))
    h = F.relu(_bn_train(F.linear(h, p["classfier.4.weight"], p["classfier.4.bias"]), p["classfier.5.weight"], p["classfier.5.bias"]))
    logits = F.log_softmax(F.linear(h, p["classfier.8.weight"], p["classfier.8.bias"]), -1)     # dropout disabled
    out["logits"] = logits
    if label is not None:
        lab = torch.as_tensor(label, dtype=torch.long)
        eps, ncls = 0.1, logits.shape[1]
        soft = torch.full_like(logits, eps / (ncls - 1)).scatter_(1, lab.view(-1, 1), 1 - eps)
        loss = -(soft * logits).sum(dim=1).mean()
        out["loss"] = loss
        if want_grads:
            loss.backward()
            out["grads"] = {k_: v.grad for k_, v in p.items() if v.grad is not None}
    return out
