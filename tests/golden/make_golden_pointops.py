#!/usr/bin/env python3
"""Generate tests/golden/cls_pointops.npz: the reference's classification L1 operators
(classification/modules/pointops/functions/pointops.py:35-354), i.e. its own autograd Functions, run on CPU tensors
over oracle/_ref (the reference's *_cuda_kernel.cu compiled unmodified as host code) — forward outputs and the three
operator backwards (gathering, grouping, interpolation).

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_pointops.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/classification"

from oracle import ref_pointops  # noqa: E402


def install():
    ref_pointops.build()
    sys.modules["pointops_cuda"] = ref_pointops.module("cls")

    def ctor(dtype):
        class _T:
            def __new__(cls, *a):
                return torch.empty(*a, dtype=dtype)
        return _T
    torch.cuda.IntTensor, torch.cuda.FloatTensor, torch.cuda.LongTensor = ctor(torch.int32), ctor(torch.float32), ctor(torch.int64)
    sys.path.insert(0, REF)


def main():
    install()
    from modules.pointops.functions import pointops as P
    g = torch.Generator().manual_seed(7)
    b, n, m, c = 3, 512, 128, 5
    xyz = (torch.rand(b, n, 3, generator=g) * 2 - 1).contiguous()
    feats = torch.randn(b, c, n, generator=g).contiguous()
    out = {"xyz": xyz.numpy(), "feats": feats.numpy()}
    fps = P.furthestsampling(xyz, m)
    out["fps"] = fps.numpy()
    f1 = feats.clone().requires_grad_()
    gath = P.gathering(f1, fps)
    w = torch.randn(gath.shape, generator=g)
    (gath * w).sum().backward()
    out["gathering"], out["gathering_w"], out["gathering_grad"] = gath.detach().numpy(), w.numpy(), f1.grad.numpy()
    new_xyz = P.gathering(xyz.transpose(1, 2).contiguous(), fps).transpose(1, 2).contiguous()
    out["new_xyz"] = new_xyz.numpy()
    for r, ns in ((0.2, 16), (0.4, 32)):
        out[f"ball_{ns}"] = P.ballquery(r, ns, xyz, new_xyz).numpy()
    out["knn9"] = P.knnquery(9, xyz, new_xyz)[0].numpy() if isinstance(P.knnquery(9, xyz, new_xyz), tuple) else P.knnquery(9, xyz, new_xyz).numpy()
    out["knn9_heap"] = P.knnquery_heap(9, xyz, new_xyz)[0].numpy() if isinstance(P.knnquery_heap(9, xyz, new_xyz), tuple) else P.knnquery_heap(9, xyz, new_xyz).numpy()
    idx = torch.from_numpy(out["ball_16"])
    f2 = feats.clone().requires_grad_()
    grp = P.grouping(f2, idx)
    w2 = torch.randn(grp.shape, generator=g)
    (grp * w2).sum().backward()
    out["grouping"], out["grouping_w"], out["grouping_grad"] = grp.detach().numpy(), w2.numpy(), f2.grad.numpy()
    dist, nidx = P.nearestneighbor(xyz, new_xyz)           # unknown = all points, known = the sampled ones
    out["nn_dist"], out["nn_idx"] = dist.numpy(), nidx.numpy()
    dr = 1.0 / (dist + 1e-8)
    weight = (dr / dr.sum(2, keepdim=True)).contiguous()
    out["nn_weight"] = weight.numpy()
    known_feats = torch.randn(b, c, m, generator=g).contiguous().requires_grad_()
    itp = P.interpolation(known_feats, nidx, weight)
    w3 = torch.randn(itp.shape, generator=g)
    (itp * w3).sum().backward()
    out["interp_feats"], out["interp"], out["interp_w"] = known_feats.detach().numpy(), itp.detach().numpy(), w3.numpy()
    out["interp_grad"] = known_feats.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "cls_pointops.npz"), **out)
    print("wrote cls_pointops.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
