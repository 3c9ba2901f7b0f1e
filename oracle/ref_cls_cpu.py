#!/usr/bin/env python3
"""TEST / MEASUREMENT INFRASTRUCTURE ONLY: times the REFERENCE's own classification CPU path:
`pointops_cuda` stubbed, the reference's classification/models/repsurf/repsurf_ssg_umb.py imported UNMODIFIED with
cuda_ops=False, model.train(), zero_grad -> forward -> SmoothClsLoss -> backward on B x N synthetic clouds, 1 warm-up +
`--steps` timed iterations, one JSON line on stdout.

The files come from /root/reference/classification (build container) or from the copy oracle/Makefile.ref stages under the
git-ignored oracle/_ref/dropin/classification (the GPU box has no /root/reference; nothing is copied into the history).
Only bench.py's `cpu_baseline` leg and tools/ run this script -- in a process of its own, because the reference's top-level
package names (`modules`, `models`, `util`) are also the names of this package's mirrors."""
import argparse
import json
import os
import sys
import time
import types

HERE = os.path.dirname(os.path.abspath(__file__))


def reference_root():
    for base in ("/root/reference/classification", os.path.join(HERE, "_ref", "dropin", "classification")):
        if os.path.exists(os.path.join(base, "modules", "repsurface_utils.py")) and \
                os.path.exists(os.path.join(base, "models", "repsurf", "repsurf_ssg_umb.py")):
            return base
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--model", default="repsurf_ssg_umb")
    a = ap.parse_args()
    root = reference_root()
    if root is None:
        print(json.dumps({"error": "reference classification CPU path not staged (make -f oracle/Makefile.ref)"}))
        return 1
    os.environ.setdefault("OMP_NUM_THREADS", str(a.threads))
    import importlib
    import numpy as np
    import torch
    torch.set_num_threads(a.threads)
    sys.modules["pointops_cuda"] = types.ModuleType("pointops_cuda")      # the compiled extension: never called with cuda_ops=False
    sys.path.insert(0, root)
    Model = importlib.import_module(f"models.repsurf.{a.model}").Model
    from util.utils import SmoothClsLoss
    args = argparse.Namespace(num_point=a.points, return_dist=True, return_center=True, return_polar=True, group_size=8,
                              umb_pool="sum", cuda_ops=False, num_class=15)
    torch.manual_seed(0)
    model = Model(args).train()
    crit = SmoothClsLoss()
    g = torch.Generator().manual_seed(123)
    pts = (torch.rand(a.batch, a.points, 3, generator=g) * 2 - 1).permute(0, 2, 1).contiguous()
    label = torch.randint(0, 15, (a.batch,), generator=g)
    times = []
    for _ in range(1 + a.steps):
        t0 = time.perf_counter()
        model.zero_grad()
        loss = crit(model(pts), label)
        loss.backward()
        times.append(time.perf_counter() - t0)
    dt = float(np.mean(times[1:]))
    print(json.dumps({"clouds_per_s": round(a.batch / dt, 3), "s_per_step": round(dt, 4), "steps_s": [round(t, 4) for t in times],
                      "threads": torch.get_num_threads(), "batch": a.batch, "points": a.points, "loss": float(loss.detach()),
                      "source": "reference tree" if root.startswith("/root/reference") else "oracle/_ref/dropin (staged, unmodified)",
                      "torch": torch.__version__}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
