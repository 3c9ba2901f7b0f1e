#!/usr/bin/env python3
"""TEST / MEASUREMENT INFRASTRUCTURE ONLY: times the REFERENCE's own segmentation step on the CPU.

The reference's segmentation path has no CPU implementation of its own: `pointops_cuda` is a compiled CUDA extension and the
Python allocates with torch.cuda.IntTensor / FloatTensor (segmentation/modules/pointops/functions/pointops.py:42-44,125-127).
Here -- exactly as tests/golden/make_golden_seg.py does for the fixtures -- `pointops_cuda` is oracle/_ref (the reference's
OWN *_cuda_kernel.cu files compiled unmodified as host code, oracle/Makefile.ref) and the two torch.cuda constructors are CPU
constructors, so every line of the reference's segmentation/models/repsurf/repsurf_umb_ssg.py + modules/ runs UNMODIFIED:
zero_grad -> forward -> cross-entropy -> backward on `--clouds` x `--points` synthetic clouds, 1 warm-up + `--steps` timed
iterations, one JSON line.  (The kernels run single-threaded block after block -- it is the reference's GPU code on a CPU, not a
tuned CPU path; the dense layers use torch's CPU kernels with `--threads` threads.)

Files: /root/reference/segmentation (build container) or the copy oracle/Makefile.ref stages under the git-ignored
oracle/_ref/dropin/segmentation (the GPU box).  Only bench.py's `cpu_baseline` leg runs this, in a process of its own."""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def reference_root():
    for base in ("/root/reference/segmentation", os.path.join(HERE, "_ref", "dropin", "segmentation")):
        if os.path.exists(os.path.join(base, "modules", "repsurface_utils.py")) and \
                os.path.exists(os.path.join(base, "models", "repsurf", "repsurf_umb_ssg.py")) and \
                os.path.exists(os.path.join(base, "modules", "pointops", "functions", "pointops.py")):
            return base
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", type=int, default=2)
    ap.add_argument("--points", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    a = ap.parse_args()
    root = reference_root()
    sys.path.insert(0, ROOT)
    from oracle import ref_pointops
    if root is None or not ref_pointops.available("seg"):
        print(json.dumps({"error": "reference segmentation path / oracle/_ref not staged (make -f oracle/Makefile.ref)"}))
        return 1
    os.environ.setdefault("OMP_NUM_THREADS", str(a.threads))
    import numpy as np
    import torch
    torch.set_num_threads(a.threads)
    sys.modules["pointops_cuda"] = ref_pointops.module("seg")

    def ctor(dtype):
        class _T:
            def __new__(cls, *args):
                if len(args) == 1 and isinstance(args[0], (list, tuple)):
                    return torch.tensor(args[0], dtype=dtype)
                return torch.empty(*args, dtype=dtype)
        return _T
    torch.cuda.IntTensor, torch.cuda.FloatTensor = ctor(torch.int32), ctor(torch.float32)
    sys.path.insert(0, root)
    from models.repsurf.repsurf_umb_ssg import Model
    margs = argparse.Namespace(return_polar=False, in_channel=6, group_size=8, num_class=13)
    torch.manual_seed(0)
    model = Model(margs).train()
    r = np.random.RandomState(125)
    n = a.clouds * a.points
    coord = torch.from_numpy((r.rand(n, 3) * 2 - 1).astype(np.float32))
    rgb = torch.from_numpy(r.rand(n, 3).astype(np.float32))
    label = torch.from_numpy(r.randint(0, 13, n).astype(np.int64))
    offset = torch.from_numpy((np.arange(1, a.clouds + 1) * a.points).astype(np.int32))
    crit = torch.nn.CrossEntropyLoss(ignore_index=255)
    times = []
    for _ in range(1 + a.steps):
        t0 = time.perf_counter()
        model.zero_grad()
        loss = crit(model([coord, rgb.clone(), offset]), label)
        loss.backward()
        times.append(time.perf_counter() - t0)
    dt = float(np.mean(times[1:]))
    print(json.dumps({"clouds_per_s": round(a.clouds / dt, 4), "s_per_step": round(dt, 4), "steps_s": [round(t, 4) for t in times],
                      "threads": torch.get_num_threads(), "clouds": a.clouds, "points": a.points, "loss": float(loss.detach()),
                      "source": "reference tree" if root.startswith("/root/reference") else "oracle/_ref/dropin (staged, unmodified)",
                      "torch": torch.__version__}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
