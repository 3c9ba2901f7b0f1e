"""Diagnostic: where do the HIP umbrella features differ from the oracle / the reference fixtures by more than 1e-5?
(run on the GPU box; prints per case: outlier count, max error, how many outliers are azimuth near-ties / kNN distance ties)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import geom_oracle as G
from repsurf_amd import ops
from tests.util import GOLDEN, cloud

dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for kind, b, n in [("uniform", 2, 1024), ("dup", 2, 512), ("grid", 1, 729), ("clustered", 2, 1024), ("uniform", 1, 3000), ("uniform", 3, 77)]:
    for k in (9, 5):
        xyz = cloud(21 + n, b, n, kind)
        nn = xyz.shape[1]
        sign = np.where(np.random.RandomState(nn).rand(b) < 0.5, -1.0, 1.0).astype(np.float32)
        feat = ops.umbrella_features(dev(xyz), k, dev(sign)).cpu().numpy()
        of, oi, tie = G.umbrella(xyz, k, sign)
        err = np.nan_to_num(np.abs(feat - of))
        tol = 1e-5 + 1e-5 * np.nan_to_num(np.abs(of))
        bad = (err > tol).reshape(b, nn, -1).any(-1)
        perpt = err.reshape(b, nn, -1).max(-1)
        print(kind, b, n, k, "outlier points", int(bad.sum()), "of", b * nn, "max err", perpt.max(), "outliers that are near_tie", int((bad & tie).sum()),
              "near_tie total", int(tie.sum()), "max err on non-tie", perpt[~tie].max() if (~tie).any() else 0)
        if bad.any():
            bi = np.argwhere(bad)[:3]
            for (bb, pp) in bi:
                print("   pt", bb, pp, "err", perpt[bb, pp], "feat", feat.reshape(b, nn, -1)[bb, pp][:12], "ref", of.reshape(b, nn, -1)[bb, pp][:12])
for tag in ["seed0", "seed1", "seed2", "seed3", "real"]:
    g = np.load(os.path.join(GOLDEN, f"geom_{tag}.npz"))
    feat = ops.umbrella_features(dev(g["xyz"][:1]), 9, dev(g["umb_inv_sign"])).cpu().numpy()
    ref = g["umb_feat"]
    _, _, tie = G.umbrella(g["xyz"][:1], 9, g["umb_inv_sign"])
    err = np.nan_to_num(np.abs(feat - ref)).reshape(ref.shape[1], -1).max(-1)
    flagged = tie[0] | g["knn9_tie_rows"][0]
    print("fixture", tag, "points > 1e-5:", int((err > 1e-5).sum()), "flagged", int(flagged.sum()), "unflagged > 1e-5:", int(((err > 1e-5) & ~flagged).sum()),
          "max clean err", err[~flagged].max(), "max err", err.max())
