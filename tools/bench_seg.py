#!/usr/bin/env python3
"""Segmentation workload (BASELINE configs[3]): RepSurf-U S3DIS network, B=16 x 4096 points x (xyz+rgb),
forward + cross-entropy + backward + Adam on one MI355X.  Not the driver's bench line (bench.py is); this
measures the widened path with the same conventions: inputs resident in HBM, HIP-event kernel breakdown,
CPU oracle timed beside it on a bounded sample.

    python tools/bench_seg.py [--steps K --warmup W --clouds 16 --points 4096 --breakdown out.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "repsurf_amd", "segmentation")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--clouds", type=int, default=16)
    ap.add_argument("--points", type=int, default=4096)
    ap.add_argument("--breakdown", default="")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu-clouds", type=int, default=2)
    args = ap.parse_args()
    from repsurf_amd import _lib
    from models.repsurf.repsurf_umb_ssg import Model
    dev = torch.device("cuda", 0)
    margs = argparse.Namespace(return_polar=False, in_channel=6, group_size=8, num_class=13)
    torch.manual_seed(0)
    model = Model(margs).to(dev).train()
    cpu_state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
    r = np.random.RandomState(0)
    n = args.clouds * args.points
    coord_h = (r.rand(n, 3) * 2 - 1).astype(np.float32)
    rgb_h = r.rand(n, 3).astype(np.float32)
    label_h = r.randint(0, 13, n).astype(np.int64)
    off_h = (np.arange(1, args.clouds + 1) * args.points).astype(np.int32)
    coord, rgb, label, offset = (torch.from_numpy(a).to(dev) for a in (coord_h, rgb_h, label_h, off_h))
    np.random.seed(1)

    def step():
        for p in model.parameters():
            p.grad = None
        loss = torch.nn.functional.cross_entropy(model([coord, rgb, offset]), label)
        loss.backward()
        opt.step()
        return loss

    mode = "eager launches"
    if not args.no_graph:
        from repsurf_amd.graph import GraphedStep
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True, capturable=True)
        eager_step = step
        step = GraphedStep(lambda x: model(x), torch.nn.functional.cross_entropy, opt, [coord, rgb, offset], label, warmup=2)
        mode = "hipgraph replay"
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    if not args.no_graph:          # per-launch HIP events need eager launches: a copy of the model, after the timed region
        import copy
        twin = copy.deepcopy(model)
        topt = torch.optim.Adam(twin.parameters(), lr=1e-3, fused=True)

        def step():
            for p in twin.parameters():
                p.grad = None
            torch.nn.functional.cross_entropy(twin([coord, rgb, offset]), label).backward()
            topt.step()
        step(); step()
    _lib.profile_enable(True)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    table = []
    for name, recs in _lib.profile_collect().items():
        by = {}
        for ms, dims in recs:
            by.setdefault(tuple(d for d in dims if not isinstance(d, str)), []).append(ms)
        for dims, ts in by.items():
            table.append({"kernel": name, "dims": list(dims), "launches_per_step": len(ts) / 3,
                          "avg_us": float(np.mean(ts)) * 1e3, "ms_per_step": float(np.sum(ts)) / 3})
    table.sort(key=lambda t: -t["ms_per_step"])
    if args.breakdown:
        json.dump({"ms_per_step": dt * 1e3, "kernels": table}, open(args.breakdown, "w"), indent=1)
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import seg_ref
        nb = args.cpu_clouds
        m = nb * args.points
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        ts = []
        for i in range(2):
            t1 = time.perf_counter()
            seg_ref.step(cpu_state, coord_h[:m], rgb_h[:m], off_h[:nb], label_h[:m], None)
            ts.append(time.perf_counter() - t1)
        cpu = {"value": round(nb / ts[-1], 3), "unit": "clouds/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"1 step (after 1 warm-up) of {nb} x {args.points}-point clouds, oracle/seg_ref.py"}
    out = {"metric": "point-clouds/sec fwd+bwd, RepSurf-U S3DIS seg, 4096-pt clouds", "value": round(args.clouds / dt, 2),
           "unit": "clouds/s", "points_per_s": round(n / dt), "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dt * 1e3, 3), "dtype": "f32" if os.environ.get("REPSURF_MLP_DTYPE", "fp32") == "fp32" else "bf16 MFMA operands, f32 accumulate/storage", "data": "synthetic uniform clouds + rgb, random-init weights",
           "config": {"workload": f"configs[3]: repsurf_umb_ssg, B={args.clouds}x{args.points}x6, fwd+CE+bwd+Adam, " + mode,
                      "loss": round(float(loss.item()), 5)},
           "hip_kernel_ms_per_step": round(sum(t["ms_per_step"] for t in table), 3),
           "top_kernels": [[t["kernel"], t["dims"], round(t["ms_per_step"], 3)] for t in table[:8]], "cpu_baseline": cpu}
    if cpu:
        out["gpu_over_cpu"] = round(out["value"] / cpu["value"], 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
