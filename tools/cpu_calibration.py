#!/usr/bin/env python3
"""Time the REFERENCE's own CPU path (BASELINE.md §2 / SURVEY §8(d): stub pointops_cuda, import
/root/reference/classification/models/repsurf/repsurf_ssg_umb.py, cuda_ops=False, model.train(), B=32x1024,
zero_grad -> forward -> SmoothClsLoss -> backward) next to the oracle port bench.py times on the GPU box (which has no
/root/reference), on THIS host, same thread count -> profiles/cpu_port_vs_reference.json.
Build container only.   python tools/cpu_calibration.py [--batch 32] [--steps 3]"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cpu_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "?"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "cpu_port_vs_reference.json"))
    a = ap.parse_args()
    threads = os.cpu_count()
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(123)
    xyz = (torch.rand(a.batch, a.points, 3, generator=g) * 2 - 1)
    label = torch.randint(0, 15, (a.batch,), generator=g)

    # ---- the reference
    sys.modules["pointops_cuda"] = types.ModuleType("pointops_cuda")
    sys.path.insert(0, "/root/reference/classification")
    from models.repsurf.repsurf_ssg_umb import Model
    from util.utils import SmoothClsLoss
    args = argparse.Namespace(num_point=a.points, return_dist=True, return_center=True, return_polar=True, group_size=8,
                              umb_pool="sum", cuda_ops=False, num_class=15)
    torch.manual_seed(0)
    model = Model(args).train()
    crit = SmoothClsLoss()
    pts = xyz.permute(0, 2, 1).contiguous()
    ref_t = []
    for i in range(1 + a.steps):
        t0 = time.perf_counter()
        model.zero_grad()
        crit(model(pts), label).backward()
        ref_t.append(time.perf_counter() - t0)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k in [k for k in sys.modules if k.split(".")[0] in ("models", "modules", "util")]:
        del sys.modules[k]
    sys.path.remove("/root/reference/classification")

    # ---- the port (what bench.py's cpu_baseline leg runs)
    from oracle import geom_oracle, torch_ref
    geom_oracle.build()
    port_t = []
    starts = [np.zeros(a.batch, np.int32)] * 3
    for i in range(1 + a.steps):
        t0 = time.perf_counter()
        torch_ref.step(state, xyz.numpy(), label.numpy(), None, starts, timed=True)
        port_t.append(time.perf_counter() - t0)
    ref_s, port_s = float(np.mean(ref_t[1:])), float(np.mean(port_t[1:]))
    out = {"host_cpu": cpu_name(), "threads": threads, "batch": a.batch, "points": a.points, "timed_steps": a.steps,
           "torch": torch.__version__,
           "reference_s_per_step": round(ref_s, 4), "reference_clouds_per_s": round(a.batch / ref_s, 3),
           "port_s_per_step": round(port_s, 4), "port_clouds_per_s": round(a.batch / port_s, 3),
           "port_over_reference_speed": round(ref_s / port_s, 3),
           "reference_steps_s": [round(t, 4) for t in ref_t], "port_steps_s": [round(t, 4) for t in port_t]}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
