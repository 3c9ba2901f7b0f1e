#!/usr/bin/env python3
"""bench.py — point-clouds/sec, forward+backward, RepSurf-U 1024-pt classifier @ B=32 per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic clouds already resident in HBM:
zero_grad -> Model.forward (umbrella constructor, 3 SurfaceAbstractionCD stages, head) ->
SmoothClsLoss -> backward (+ the RCCL gradient all-reduce when N > 1) -> optimizer step
(BASELINE.md's definition stops at backward; the Adam step is kept inside the timed region so no
part of a training step is skipped — `--no-optim` reproduces the BASELINE.md definition exactly).
Batches shard across ranks (one process per GPU, weak scaling: B=32 per rank); the gradients are averaged by ONE
RCCL all-reduce of a flat 5.9 MB buffer over xGMI, recorded inside the step's hipGraph (repsurf_amd.graph.PipelinedStep;
eager launches: DistributedDataParallel with a single bucket).

Rank 0 prints ONE JSON line with the driver's contract fields plus
  "roofline":     the dominant instrumented HIP kernel, timed live with HIP events on the launch stream,
  "cpu_baseline": the REFERENCE's own CPU path (its unmodified model + modules files, cuda_ops=False; oracle/ref_cls_cpu.py in
                  a process of its own) timed on this host on the same workload, rank 0, N=1 only -- kind "reference"; the
                  oracle port (oracle/torch_ref.py + oracle/geom_oracle.c) rides along as cpu_baseline.port, and is the
                  baseline itself (kind "port") where the reference files are not staged.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "repsurf_amd", "classification")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

PEAK_HBM_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PEAK_F32_MFMA_TF = 157.3    # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak
PEAK_BF16_MFMA_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (--dtype bf16 launches only)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="clouds per GPU")
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--model", default="repsurf_ssg_umb")
    ap.add_argument("--workload", default="cls", choices=["cls", "seg"],
                    help="cls: BASELINE configs[1] (the metric; configs[4] with --dtype bf16 --batch 64 --points 2048); "
                         "seg: configs[3], RepSurf-U S3DIS segmentation, 16 clouds x 4096 points x (xyz+rgb) per GPU")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"],
                    help="arithmetic of the shared-MLP row GEMMs: fp32 MFMA (configs[1], the metric) or bf16 MFMA operands and "
                         "bf16 storage of the conv outputs with fp32 accumulation (configs[4]: use --batch 64 --points 2048)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="N=1 graph mode: compute each batch's geometry inside its own step instead of under the previous "
                         "batch's network (repsurf_amd.graph.PipelinedStep)")
    ap.add_argument("--no-optim", action="store_true", help="stop the step at backward (BASELINE.md definition)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-arithmetic", action="store_true",
                    help="skip the second leg: the same step under the other fp32 product arithmetic (fp32 MFMA), 30 steps in a child process")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the bracketing legs of the classification line: dense groups, real scans, eager launches (child processes, 20 steps each)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the per-launch HIP events")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--cpu-batch", type=int, default=32, help="clouds in the CPU-baseline sample (default: the GPU batch)")
    ap.add_argument("--cpu-threads", type=int, default=32)
    ap.add_argument("--breakdown", default="", help="write a per-kernel timing breakdown JSON here")
    ap.add_argument("--launch-log", default="", help="write the ordered (abi call, sizes) list of the timing pass "
                                                     "(tools/traffic_from_pmc.py matches it to a rocprofv3 --pmc run)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--data", default="uniform", choices=["uniform", "real"],
                    help="uniform: synthetic uniform clouds in [-1,1]^3 (the metric's data); real: the 4 scanned objects of "
                         "tests/golden/geom_real.npz (first 1024 rows of the reference's visualization/*.txt clouds) tiled to the batch "
                         "-- surfaces, not volumes: 4-6x more DISTINCT ball-query slots, i.e. more shared-MLP rows (DESIGN 6/7)")
    ap.add_argument("--ragged", action="store_true",
                    help="--workload seg: what the reference's loader emits -- every step a DIFFERENT packed batch (cloud sizes drawn in "
                         "[points/2, points], segmentation/util/data_util.py:15-23) through ONE captured network graph (RaggedSegStep: row counts "
                         "as device data); the eager loop is timed beside it (--no-graph: only the eager loop)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher check: spawn / join the ranks, one all-reduce over the process group (gloo where there is no "
                         "HIP device), rank 0 prints one JSON line; no model, no kernels (tests/test_ddp_gloo.py)")
    ap.add_argument("--min-seconds", type=float, default=0.5,
                    help="the timed region replays max(--steps, ceil(min-seconds / step time)) steps (`steps_timed` on the line): "
                         "a 20-step sample of a 1.6 ms step is 32 ms, one clock wobble wide")
    return ap.parse_args()


def model_args():
    return argparse.Namespace(num_point=1024, return_dist=True, return_center=True, return_polar=True,
                              group_size=8, umb_pool="sum", cuda_ops=True, num_class=15)


def synthetic_batch(seed, b, n, device):
    g = torch.Generator().manual_seed(seed)
    xyz = (torch.rand(b, n, 3, generator=g) * 2 - 1)
    label = torch.randint(0, 15, (b,), generator=g)
    return xyz.permute(0, 2, 1).contiguous().to(device), label.to(device)


def real_batch(seed, b, n, device):
    """The 4 real scans of tests/golden/geom_real.npz (1024 rows each, unit-sphere normalised like the reference's loader)
    tiled to b clouds; every copy gets its own rotation about the up axis and a small jitter so that no two clouds of the
    batch are identical (the reference's train-time augmentation, classification/tool/train_cls_scanobjectnn.py)."""
    fx = np.load(os.path.join(ROOT, "tests", "golden", "geom_real.npz"))["xyz"].astype(np.float32)
    assert n == fx.shape[1], f"--data real holds {fx.shape[1]}-point clouds"
    r = np.random.RandomState(seed)
    out = np.empty((b, n, 3), np.float32)
    for i in range(b):
        a = r.rand() * 2 * np.pi
        rot = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
        out[i] = fx[i % fx.shape[0]] @ rot + (0.002 * r.randn(n, 3)).astype(np.float32)
    label = torch.from_numpy(r.randint(0, 15, (b,)))
    return torch.from_numpy(out).permute(0, 2, 1).contiguous().to(device), label.to(device)


def distinct_slot_fraction(points, stages=((512, 0.2, 32), (128, 0.4, 64))):
    """Fraction of ball-query slots holding a DISTINCT neighbour at the model's two grouped stages: what the compacted
    shared-MLP stacks process relative to the dense (B*S*nsample)-row formulation."""
    from repsurf_amd import ops
    xyz = points.permute(0, 2, 1).contiguous()
    out = []
    for s_, r, ns in stages:
        start = torch.zeros(xyz.shape[0], dtype=torch.int32, device=xyz.device)
        centres = ops.gather_rows(xyz, ops.furthestsampling(xyz, s_, start))
        _, cnt = ops.ballquery(r, ns, xyz, centres, return_count=True)
        out.append(round(float(cnt.float().mean().item()) / ns, 4))
        xyz = centres
    return out


def reference_cpu_baseline(args, threads):
    """the REFERENCE's own classification CPU path (its unmodified model + modules files, cuda_ops=False,
    pointops_cuda stubbed) timed on this host by oracle/ref_cls_cpu.py in a process of its own (its package names collide
    with this package's mirrors).  The files are /root/reference's in the build container and the copy oracle/Makefile.ref
    stages under the git-ignored oracle/_ref/dropin/ on the GPU box.  None when they are not there."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "ref_cls_cpu.py"), "--batch", str(min(args.batch, args.cpu_batch)),
           "--points", str(args.points), "--steps", str(args.cpu_steps), "--threads", str(threads), "--model", args.model]
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env={**os.environ, "OMP_NUM_THREADS": str(threads)})
        rec = json.loads(res.stdout.strip().splitlines()[-1])
        return None if "error" in rec else rec
    except (subprocess.SubprocessError, OSError, ValueError, IndexError):
        return None


def reference_seg_cpu_baseline(points, threads):
    """SURVEY 8(d) for configs[3]: the REFERENCE's own segmentation step timed on this host by oracle/ref_seg_cpu.py (its unmodified
    Python over oracle/_ref, the host build of its own kernels), 2 clouds, 1 warm-up + 1 timed step, in a process of its own.
    None when the files / the host build are not staged."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "ref_seg_cpu.py"), "--clouds", "2", "--points", str(points), "--steps", "1",
           "--threads", str(threads)]
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env={**os.environ, "OMP_NUM_THREADS": str(threads)})
        rec = json.loads(res.stdout.strip().splitlines()[-1])
        return None if "error" in rec else rec
    except (subprocess.SubprocessError, OSError, ValueError, IndexError):
        return None


def cpu_baseline(args, state):
    """The CPU path on the same workload: B x points clouds, zero_grad -> forward -> SmoothClsLoss -> backward, 1 warm-up +
    `--cpu-steps` timed iterations on this host.  /root/reference does not exist on the GPU box, so the
    timed code is the oracle port (oracle/torch_ref.py: the same PyTorch CPU kernels for the dense part, the C geometry
    threaded over clouds); profiles/cpu_port_vs_reference.json (tools/cpu_calibration.py, build container) times the port
    and the reference's own path side by side on one host, and its ratio is quoted in `sample`."""
    from oracle import geom_oracle, torch_ref
    geom_oracle.build()
    threads = min(os.cpu_count() or 1, args.cpu_threads)
    torch.set_num_threads(threads)
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    nb = min(args.batch, args.cpu_batch)
    g = torch.Generator().manual_seed(123)
    xyz = (torch.rand(nb, args.points, 3, generator=g) * 2 - 1).numpy()
    label = torch.randint(0, 15, (nb,), generator=g).numpy()
    starts = [np.zeros(nb, np.int32)] * 3
    times = []
    for i in range(1 + args.cpu_steps):
        t0 = time.perf_counter()
        torch_ref.step(state, xyz, label, None, starts, arch=args.model, timed=True)
        times.append(time.perf_counter() - t0)
    dt = float(np.mean(times[1:]))
    cpu = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    calib = ""
    try:
        c = json.load(open(os.path.join(ROOT, "profiles", "cpu_port_vs_reference.json")))
        calib = (f"; calibration on one host ({c['host_cpu']}, {c['threads']} threads, B={c['batch']}): reference path "
                 f"{c['reference_clouds_per_s']} clouds/s, this port {c['port_clouds_per_s']} clouds/s = "
                 f"{c['port_over_reference_speed']}x the reference's speed (the port is the FASTER of the two, so gpu_over_cpu "
                 f"understates the ratio to the reference)")
    except (OSError, ValueError, KeyError):
        pass
    port = {"value": round(nb / dt, 3), "unit": "clouds/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{args.cpu_steps} timed steps (after 1 warm-up) of B={nb}x{args.points} clouds, fwd+loss+bwd; "
                      f"dense ops torch {torch.__version__} CPU, geometry C/OpenMP, {threads} threads; "
                      f"host CPU: {cpu}" + calib,
            "s_per_step": round(dt, 4), "steps_s": [round(t, 4) for t in times[1:]]}
    ref = reference_cpu_baseline(args, threads)
    if ref is None:
        return port
    # the reference's own code is the baseline; the port's number rides along for comparison
    return {"value": ref["clouds_per_s"], "unit": "clouds/s", "cores": ref["threads"], "kind": "reference",
            "sample": f"the reference's own CPU path ({ref['source']}: classification/models/repsurf/{args.model}.py over its modules/, "
                      f"cuda_ops=False, pointops_cuda stubbed), {args.cpu_steps} timed steps (after 1 warm-up) of B={ref['batch']}x{ref['points']} "
                      f"uniform clouds, zero_grad+fwd+SmoothClsLoss+bwd, dense uniform-cube data (no compaction on either CPU leg), "
                      f"torch {ref['torch']} CPU, {ref['threads']} threads; host CPU: {cpu}",
            "s_per_step": ref["s_per_step"], "steps_s": ref["steps_s"][1:],
            "port": {"value": port["value"], "s_per_step": port["s_per_step"], "kind": "port",
                     "what": "oracle/torch_ref.py + oracle/geom_oracle.c (C/OpenMP geometry), same batch shape and thread count"}}


def geometry_lines(device, points):
    """The two geometry kernels the north-star prices separately, timed live with HIP events on the launch stream
    (10 back-to-back launches after 3 warm-ups): ball query against the HBM roofline by its ALGORITHMIC bytes
    4*(3BN + 3BS + BS*nsample) at the step's sa1 shape for B = 32 / 256 / 2048 clouds per launch, and FPS as
    microseconds per pick (a dependency chain of S-1 reductions: neither HBM- nor MFMA-bound)."""
    from repsurf_amd import ops

    def timed(fn, reps=10):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3          # seconds per launch

    g = torch.Generator().manual_seed(7)
    ball = {}
    n, s_, ns, r = points, 512, 32, 0.2
    for b in (32, 256, 2048):
        xyz = (torch.rand(b, n, 3, generator=g) * 2 - 1).to(device)
        centres = xyz[:, :s_].contiguous()
        t = timed(lambda: ops.ballquery(r, ns, xyz, centres))
        nbytes = 4.0 * (3 * b * n + 3 * b * s_ + b * s_ * ns)
        ball[str(b)] = {"us": round(t * 1e6, 2), "achieved": round(nbytes / t / 1e9, 1), "frac": round(nbytes / t / 1e9 / PEAK_HBM_GBS, 4),
                        "algorithmic_bytes": nbytes}
        # Round 6: the cell list built once per cloud as an image in HBM (rs_ballquery_grid_build) and the query alone
        # against it (rs_ballquery_grid_query) -- what a consumer pays that queries the same coordinates more than once.  Algorithmic bytes
        # of the query: the image it reads (16 B per point sorted as float4 + 2 B index + the cells' starts) + centres + rows written;
        # of the build: the cloud read + the image written.
        if ops.ballquery_grid_ok(n, ns):
            grid = ops.BallGrid(r, xyz)
            image = grid.image.numel()
            tq = timed(lambda: grid.query(ns, centres))
            tb = timed(lambda: ops.BallGrid(r, xyz))
            qbytes = float(image) + 4.0 * (3 * b * s_ + b * s_ * ns)
            bbytes = 4.0 * 3 * b * n + float(image)
            ball[str(b)]["prebuilt_grid"] = {
                "query_us": round(tq * 1e6, 2), "query_algorithmic_bytes": qbytes, "query_achieved": round(qbytes / tq / 1e9, 1),
                "query_frac": round(qbytes / tq / 1e9 / PEAK_HBM_GBS, 4),
                # the SAME rows priced on the fused launch's algorithmic bytes (what the north-star's 40 % is quoted on)
                "query_frac_on_fused_bytes": round(nbytes / tq / 1e9 / PEAK_HBM_GBS, 4),
                "build_us": round(tb * 1e6, 2), "build_algorithmic_bytes": bbytes, "build_frac": round(bbytes / tb / 1e9 / PEAK_HBM_GBS, 4),
                "build_plus_query_us": round((tb + tq) * 1e6, 2)}
            del grid
        del xyz, centres
    fps = {}
    # A pick is a dependent reduction inside ONE workgroup (one cloud = one CU): the launch time does not depend on the
    # number of clouds until the chip is full.  The *_x3 entry is the sa1 chain of THREE batches of 32 clouds in one
    # launch (96 of 256 CUs) with its cost amortised per batch of 32 -- what a loader that is three batches ahead pays;
    # the step itself hides the chain of ONE batch under the previous batch's network (config.launch).
    for name, (b, nn, m) in {"sa1_1024_to_512": (32, points, 512), "sa2_512_to_128": (32, 512, 128),
                             "sample_2048_to_1024": (32, 2048, 1024), "sa1_1024_to_512_x3": (96, points, 512)}.items():
        xyz = (torch.rand(b, nn, 3, generator=g) * 2 - 1).to(device)
        start = torch.zeros(b, dtype=torch.int32, device=device)
        t = timed(lambda: ops.furthestsampling(xyz, m, start))
        fps[name] = {"us": round(t * 1e6, 1), "us_per_pick": round(t * 1e6 / (m - 1), 4), "clouds": b}
        if b > 32:
            fps[name]["us_per_pick_per_batch_of_32"] = round(t * 1e6 / (m - 1) / (b / 32), 4)
    return ({"kernel": "rs_ballquery", "shape": f"N={n} S={s_} nsample={ns} r={r}", "bound": "hbm", "unit": "GB/s",
             "peak": PEAK_HBM_GBS, "clouds_per_launch": ball}, fps)


def algorithmic_cost(name, dims):
    """(unit, amount) of algorithmic work of one launch of an instrumented ABI call."""
    if name == "rs_ballquery":
        b, n, m, ns = dims
        return "bytes", 4.0 * (3 * b * n + 3 * b * m + b * m * ns)
    # compacted launches: dims[0] is the capacity, the rows really processed come as a trailing "rows=<mean>" note
    notes = dict(d.split("=", 1) for d in dims if isinstance(d, str))
    dims = [d for d in dims if not isinstance(d, str)]
    if "rows" in notes:
        dims[0] = float(notes["rows"])
    if name in ("rs_mlp_gemm_rows", "rs_mlp_gemm_rows_bf16"):
        rows, kdim, cols = dims[:3]
        return "flops", 2.0 * rows * kdim * cols
    if name in ("rs_mlp_wgrad", "rs_mlp_wgrad_bf16"):
        rows, ncols, kcols = dims[:3]
        return "flops", 2.0 * rows * ncols * kcols
    return None, 0.0


def algorithmic_bytes(name, dims):
    """HBM bytes one launch must move (each operand / mask tensor read once, the output written once):
    row GEMM: rows * (K * operand tensors + N * (1 + mask tensors)) elements; weight gradient: rows * (N * P tensors +
    K * Q tensors) elements; 4 bytes per element, 2 for the tensors the launch's "sb=" note marks as bf16-stored."""
    notes = dict(d.split("=", 1) for d in dims if isinstance(d, str))
    ints = [d for d in dims if not isinstance(d, str)]
    rows = float(notes["rows"]) if "rows" in notes else float(ints[0])
    # tensors an operand mode reads, as (a, b) presence: ID / RELU1: a; RELU2 / AFF2: a and b; POOLED: b only (the pooled
    # gradient behind a is small); BCAST: the broadcast source is tiny.  "sb=" marks the tensors stored as bf16 (2 bytes).
    reads = {0: (1, 0), 1: (1, 0), 2: (1, 1), 3: (1, 1), 4: (0, 1), 5: (0, 0)}
    sb = notes.get("sb", "00000")
    width = lambda i: 2.0 if sb[i] == "1" else 4.0
    if name.startswith("rs_mlp_gemm_rows") and "op" in notes:
        k, n = ints[1], ints[2]
        epi = notes.get("epi", "0")
        ra, rb = reads[int(notes["op"])]
        total = k * (ra * width(0) + rb * width(1)) + n * width(2)
        if epi.startswith("2"):
            total += n * width(3) + (n * width(4) if epi.endswith("+2") else 0.0)
        return rows * total
    if name.startswith("rs_mlp_wgrad") and "p" in notes:
        n, k = ints[1], ints[2]
        pa, pb = reads[int(notes["p"])]
        qa, qb = reads[int(notes["q"])]
        return rows * (n * (pa * width(0) + pb * width(1)) + k * (qa * width(2) + qb * width(3)))
    return None


def gemm_family_in_graph(net, criterion, points, label, replays=20):
    """The matrix-pipe launches of ONE step (every rs_mlp_gemm_rows / rs_mlp_wgrad call of forward + backward) as they run inside
    a replayed step: a forward + backward of `net` is captured as a hipGraph with the GEMM-family ABI calls recorded
    (repsurf_amd.graph.GraphedStep(record_calls=True)); those calls are then captured on their own into ONE hipGraph and replayed
    `replays` times between two HIP events.  Per-launch events of the eager pass include ~5-10 us of idle device in front of every
    launch (the host is slower than the kernels); a replay has none, which is what the training step's graph sees (rocprofv3 of
    the replayed step agrees: profiles/r04/).  Every pointer of the recorded calls lies in the first graph's private memory pool,
    which lives until this function returns: the replays read and write nothing else.
    -> {"ms_per_step": all launches, "launches": n, "by_class_us": {class key: avg us per launch}}"""
    from repsurf_amd import _lib
    from repsurf_amd.graph import GraphedStep
    holder = GraphedStep(net, criterion, None, points, label, warmup=2, record_calls=True)
    calls = holder.recorded_calls
    if not calls:
        return None
    holder()                      # one replay of the whole step: a capture runs nothing, and the recorded launches read device-side row
    torch.cuda.synchronize()      # counts (compacted groups), indices and activations that only the step's other kernels produce

    def timed_graph(subset):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            _lib.replay_calls(subset, torch.cuda.current_stream().cuda_stream)
        for _ in range(3):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(replays):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / replays

    out = {"ms_per_step": timed_graph(calls), "launches": len(calls), "replays": replays}
    # per launch class (ABI call + sizes + operand / epilogue modes, as in roofline_from_profile): its launches alone, back to back
    classes = {}
    for name, a in calls:
        ints = tuple(x for x, t in zip(a, _lib.SIGNATURES[name]) if t is _lib.c_int or t is _lib.c_ll)
        if name.startswith("rs_mlp_gemm_rows"):
            key = (name,) + ints + (f"op={a[4]._obj.mode}", f"epi={a[7]._obj.mode}{'+2' if a[7]._obj.my2 else ''}")
        else:
            key = (name,) + ints + (f"p={a[4]._obj.mode}", f"q={a[5]._obj.mode}")
        classes.setdefault(key, []).append((name, a))
    # a class on its own: its launches, ten copies per graph (a one-kernel graph would time the graph launch): back to back with
    # itself on a warm cache -- an optimistic bound, reported next to the in-step estimate roofline_from_profile makes
    out["by_class_us"] = {k: timed_graph(v * 10) * 1e3 / (10 * len(v)) for k, v in classes.items()}
    # ... and INSIDE the step's family graph: the recorded launches captured as one graph again, with a device
    # wall-clock stamp (rs_timestamp: a one-thread launch storing s_memrealtime) in front of and behind every launch of the class --
    # HIP events cannot be recorded inside a replayed graph on this runtime (tools/probe_graph_events.py).  A launch's duration is the
    # stamp-to-stamp interval minus ONE stamp-to-stamp gap (calibrated by three stamps back to back at the head of the same graph; the
    # interval holds two such gaps, but a 512-workgroup launch takes longer to start than a one-thread stamp: with one gap taken off the
    # figure agrees with the rocprofv3 kernel trace of the replayed step within ~1 %, with two it is 3 % BELOW it -- the conservative
    # form is the one reported); averaged over the launches of the class and 10 replays.  The class runs between the launches that precede and follow
    # it in the step, on the cache state they leave (the alone-replay above is ten copies back to back on a warm cache: optimistic).
    # For the three classes with the largest alone totals (one of them is the line's dominant kernel).
    top = sorted(classes, key=lambda k: -out["by_class_us"][k] * len(classes[k]))[:3]
    out["in_step_us"] = {}
    khz = _lib.load().rs_timestamp_khz()
    for k in top:
        members = {id(a) for _, a in classes[k]}
        nst = 3 + 2 * len(members)
        stamps = torch.zeros(nst, dtype=torch.int64, device="cuda")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            st = torch.cuda.current_stream().cuda_stream
            for j in range(3):
                _lib.call("rs_timestamp", stamps.data_ptr() + 8 * j, st)
            j = 3
            for name, a in calls:
                if id(a) in members:
                    _lib.call("rs_timestamp", stamps.data_ptr() + 8 * j, st)
                    _lib.replay_calls([(name, a)], st)
                    _lib.call("rs_timestamp", stamps.data_ptr() + 8 * (j + 1), st)
                    j += 2
                else:
                    _lib.replay_calls([(name, a)], st)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        acc, gaps = [], []
        for _ in range(10):
            g.replay()
            torch.cuda.synchronize()
            t = stamps.cpu().numpy().astype(np.float64) * (1e3 / khz)         # microseconds
            gap = 0.5 * (t[2] - t[0])                                          # one stamp -> next stamp: launch gap + the stamp itself
            gaps.append(gap)
            acc.append(float(np.mean([t[i + 1] - t[i] - gap for i in range(3, nst, 2)])))
        out["in_step_us"][k] = float(np.mean(acc))
        out.setdefault("stamp_gap_us", round(float(np.mean(gaps)), 2))
        del g
    del holder
    return out


def roofline_from_profile(prof, timed_steps, dtype, in_graph=None):
    """prof: {abi name: [(ms, dims)]} of an eager pass -> (roofline of the MFMA / byte-priced launch class with the largest
    time per step, table of all classes).  in_graph (gemm_family_in_graph): the same launches timed inside a replayed hipGraph --
    when present, the GEMM-shaped classes are priced on THOSE durations (what the step runs at) and the per-launch HIP-event
    figures of the eager pass ride along as `eager_*`."""
    roofline = None
    table = []
    graph_us, step_us = {}, {}
    if in_graph:
        for key, us in in_graph["by_class_us"].items():
            graph_us[(key[0],) + tuple(str(d) for d in key[1:])] = us
        for key, us in in_graph.get("in_step_us", {}).items():
            step_us[(key[0],) + tuple(str(d) for d in key[1:])] = us
    for name, recs in prof.items():
        by_dims = {}
        rows_of = {}
        for t_ms, dims in recs:
            # the class key: sizes + operand / epilogue modes; "rows=<n>" (row count of a compacted launch) varies a little per step
            static = tuple(d for d in dims if not (isinstance(d, str) and d.startswith("rows=")))
            by_dims.setdefault(static, []).append(t_ms)
            for d in dims:
                if isinstance(d, str) and d.startswith("rows="):
                    rows_of.setdefault(static, []).append(int(d.split("=")[1]))
        for dims, ts in by_dims.items():
            if dims in rows_of:
                dims = dims + (f"rows={float(np.mean(rows_of[dims])):.1f}",)
            unit, amount = algorithmic_cost(name, dims)
            row = {"kernel": name, "dims": list(dims), "launches": len(ts), "avg_us": float(np.mean(ts)) * 1e3,
                   "total_ms_per_step": float(np.sum(ts)) / max(1, timed_steps), "unit": unit, "amount": amount}
            gkey = (name,) + tuple(str(d) for d in dims if not (isinstance(d, str) and (d.startswith("rows=") or d.startswith("sb="))))
            if gkey in graph_us:
                row["alone_avg_us"] = graph_us[gkey]
            if gkey in step_us:
                row["in_step_avg_us"] = step_us[gkey]
            table.append(row)
    if in_graph:
        # Inside the replayed step a GEMM-family launch has no idle device in front of it: the classes are priced on their
        # graph-replay durations (the class's launches of one step, ten copies in one graph, 20 replays between two HIP events);
        # the per-launch events of the eager pass (host-paced) ride along
        for row in table:
            if "alone_avg_us" in row:
                row["eager_avg_us"] = row["avg_us"]
                # the in-step duration where it was measured (the classes that can be the dominant one), the alone-replay otherwise
                row["avg_us"] = row.get("in_step_avg_us", row["alone_avg_us"])
                row["total_ms_per_step"] = row["avg_us"] * 1e-3 * row["launches"] / max(1, timed_steps)
    table.sort(key=lambda r: -r["total_ms_per_step"])
    # the dominant kernel of the line: the GEMM-shaped class with the largest time per step (the ball query has a roofline object of its
    # own, `roofline_ballquery`: on the bf16 B=64 x 2048 line its scan kernel used to outweigh every single GEMM class and took this slot,
    # VERDICT r5 weak 9); a byte-priced class only when no GEMM-shaped one was launched
    ranked = [r for r in table if r["unit"] == "flops"] or [r for r in table if r["unit"] is not None]
    for row in ranked:
        sec = row["avg_us"] * 1e-6
        if row["unit"] == "flops":
            ach = row["amount"] / sec / 1e12
            peak = PEAK_BF16_MFMA_TF if row["kernel"].endswith("_bf16") else PEAK_F32_MFMA_TF
            roofline = {"kernel": row["kernel"], "dims": row["dims"], "bound": "mfma", "achieved": round(ach, 2),
                        "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4)}
        else:
            ach = row["amount"] / sec / 1e9
            roofline = {"kernel": row["kernel"], "dims": row["dims"], "bound": "hbm", "achieved": round(ach, 2),
                        "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 5)}
        roofline["avg_launch_us"] = round(row["avg_us"], 2)
        if row["unit"] == "flops" and dtype == "fp32" and mlp_hip_split3():
            # the product is fp32 (error against fp64 = the fp32 MFMA's, profiles/r04/gemm_split3_ab.txt) but EXECUTED as six bf16 MFMAs
            # over three-part operands: `frac` prices the algorithmic fp32 flops against the fp32 matrix peak, this entry the
            # executed instructions against the bf16 one
            roofline["mfma_pipe"] = {"instructions": "6 x v_mfma_f32_32x32x16_bf16 per 16 k (operands as three bf16 parts, fp32 accumulation); "
                                                     "RS_GEMM_SPLIT3=0: 8 x v_mfma_f32_32x32x2_f32",
                                     "executed_tflops": round(6 * ach, 1), "bf16_peak": PEAK_BF16_MFMA_TF,
                                     "frac_of_bf16_peak": round(6 * ach / PEAK_BF16_MFMA_TF, 4)}
        if "eager_avg_us" in row:
            if "in_step_avg_us" in row:
                # `frac` / `achieved` / `avg_launch_us` are the IN-STEP figures; the optimistic alone-replay rides along
                alone_sec = row["alone_avg_us"] * 1e-6
                roofline["alone_avg_launch_us"] = round(row["alone_avg_us"], 2)
                roofline["alone_frac"] = round(row["amount"] / alone_sec / (1e12 if row["unit"] == "flops" else 1e9) / roofline["peak"], 4)
                roofline["timed"] = ("avg_launch_us / frac: IN the step's GEMM family -- the recorded step's GEMM + weight-gradient launches replayed as one hipGraph with a "
                                     "device wall-clock stamp (rs_timestamp, s_memrealtime) in front of and behind this class's launches; stamp-to-stamp interval minus one "
                                     "calibrated stamp-to-stamp gap (stamp_gap_us), 10 replays (HIP events cannot be recorded inside a replayed graph on this runtime; the rocprofv3 kernel trace of "
                                     "the replayed step by kernel instance and grid is profiles/r06/cls_graph_kernel_stats_by_grid.csv); alone_*: this class's launches "
                                     "alone, ten copies per graph, back to back on a warm cache (optimistic); eager_avg_launch_us: per-launch HIP events of an eager "
                                     "pass (host-paced: idle device in front of every launch)")
                roofline["stamp_gap_us"] = in_graph.get("stamp_gap_us")
            else:
                roofline["timed"] = ("avg_launch_us: inside a replayed hipGraph -- this class's launches of one recorded step, ten copies per graph, back to "
                                     "back, 20 replays between two HIP events (gemm_family_in_graph); "
                                     "eager_avg_launch_us: per-launch HIP events of an eager pass (host-paced: idle device in front of every launch)")
            roofline["eager_avg_launch_us"] = round(row["eager_avg_us"], 2)
        roofline["launches_per_step"] = row["launches"] // max(1, timed_steps)
        roofline["traffic"] = traffic_from_profiles(row["kernel"], row["dims"])
        roofline["traffic_source"] = ("profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this launch class, committed "
                                      "(tools/gpu_profile.sh, tools/traffic_from_pmc.py); NOT measured in this run") if roofline["traffic"] is not None else None
        ab = algorithmic_bytes(row["kernel"], row["dims"])
        roofline["algorithmic_bytes"] = ab
        if ab and row["unit"] == "flops":
            # a GEMM-shaped launch is priced against the roof that binds it: narrow layers (K, N <= 64) move more bytes per
            # flop than the chip's balance (157 TF / 8 TB/s = 20 flop/B) and are HBM-bound, the wide ones are MFMA-bound
            t_hbm, t_mfma = ab / (PEAK_HBM_GBS * 1e9), row["amount"] / (roofline["peak"] * 1e12)
            roofline["frac_mfma"], roofline["frac_hbm"] = roofline["frac"], round(t_hbm / sec, 4)
            if t_hbm > t_mfma:
                roofline.update(bound="hbm", achieved=round(ab / sec / 1e9, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(t_hbm / sec, 4))
        break
    # every MFMA launch of the step together (all shared-MLP GEMMs + weight-gradient GEMMs)
    fl = sum(r["amount"] * r["launches"] for r in table if r["unit"] == "flops")
    tm = sum(r.get("eager_avg_us", r["avg_us"]) * r["launches"] for r in table if r["unit"] == "flops") * 1e-6      # (per-launch events of the eager pass)
    if roofline is not None and tm > 0:
        roofline["all_mfma_launches"] = {"achieved": round(fl / tm / 1e12, 2), "unit": "TFLOP/s",
                                         # (bf16 run: row GEMMs on the bf16 pipe, weight gradients on the fp32 one -- no single peak)
                                         "frac": round(fl / tm / 1e12 / PEAK_F32_MFMA_TF, 4) if dtype == "fp32" else None,
                                         "ms_per_step": round(tm * 1e3 / max(1, timed_steps), 4)}
        if in_graph:
            # every GEMM + weight-gradient launch of one step as ONE replayed graph (no idle device between launches)
            fl_step = fl / max(1, timed_steps)
            sec = in_graph["ms_per_step"] * 1e-3
            eager = roofline["all_mfma_launches"]
            roofline["all_mfma_launches"] = {
                "achieved": round(fl_step / sec / 1e12, 2), "unit": "TFLOP/s",
                "frac": round(fl_step / sec / 1e12 / PEAK_F32_MFMA_TF, 4) if dtype == "fp32" else None,
                "ms_per_step": round(in_graph["ms_per_step"], 4), "launches_per_step": in_graph["launches"],
                "timed": f"the {in_graph['launches']} GEMM-family launches of one recorded step captured as one hipGraph, {in_graph['replays']} replays between two HIP events",
                "eager": {"achieved": eager["achieved"], "frac": eager["frac"], "ms_per_step": eager["ms_per_step"],
                          "timed": "sum of per-launch HIP events of the eager pass"}}
    return roofline, table


ARITHMETIC = {True: "fp32 via 3xbf16 split, 6 MFMA", False: "fp32 MFMA (v_mfma_f32_32x32x2_f32)"}


def child_leg(args, extra=(), env_extra=None, steps=30, timeout=240):
    """One short run of this script in a child process -- same workload, batch and step, no CPU baseline, no per-launch timing, no
    further legs -- with `extra` flags / `env_extra` environment on top: the parsed JSON line, or None when it could not run.  The
    legs are reported extras next to the headline, never a reason to lose it: bounded by `timeout` seconds each, the test hook
    REPSURF_BENCH_DUMP is not inherited (a child would overwrite the parent's dump)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", str(max(2, min(args.warmup, 5))),
           "--batch", str(args.batch), "--points", str(args.points), "--model", args.model, "--workload", args.workload,
           "--dtype", args.dtype, "--min-seconds", "0.2", "--no-cpu-baseline", "--no-kernel-timing", "--no-alt-arithmetic", "--no-extra-legs"]
    if args.data != "uniform" and "--data" not in extra:
        cmd += ["--data", args.data]
    for flag, on in (("--no-pipeline", args.no_pipeline), ("--no-optim", args.no_optim), ("--no-graph", args.no_graph)):
        if on and flag not in extra:
            cmd.append(flag)
    cmd += list(extra)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "REPSURF_BENCH_DUMP")}
    env.update(env_extra or {})
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:  # noqa: BLE001 - a reported extra, never a reason to lose the headline line
        print(f"[bench] child leg {list(extra)} {env_extra or {}} failed: {e!r}", file=sys.stderr)
        return None


def alt_arithmetic_ms(args):
    """The same workload and step under the fp32-MFMA product instances (RS_GEMM_SPLIT3=0 is read once per process: a child
    process), 30 timed steps: its ms_per_step, reported next to the headline as `fp32_mfma_ms_per_step`."""
    rec = child_leg(args, env_extra={"RS_GEMM_SPLIT3": "0"})
    return None if rec is None else float(rec["ms_per_step"])


def bracketing_legs(args, value, cpu):
    """what the headline's special circumstances are worth, on the driver's own line.  Three short child runs of the
    same step (classification workload):
      dense_clouds_per_s       REPSURF_COMPACT=0 -- every ball-query slot goes through the shared MLPs, as in the reference and in the CPU
                               baseline (the headline processes the distinct slots only: exact, but 5-7x fewer rows on uniform cubes);
      real_scans_clouds_per_s  --data real -- the reference's scanned objects (surfaces: 4-6x more distinct slots than uniform cubes);
      eager_clouds_per_s       --no-graph -- the reference's own loop shape (classification/tool/train_cls_scanobjectnn.py:212-234:
                               zero_grad, classifier(points), loss, backward, step, launched eagerly from Python): what a user gets by
                               changing PYTHONPATH only, host-paced; 60 steps, mean (with the interpreter's
                               collection pauses) and `eager_ms_per_step_median` (the steady state).
    gpu_over_cpu_dense = dense GPU step / dense CPU step: both sides run the same dense formulation."""
    out = {}
    for key, extra, env in (("dense_clouds_per_s", (), {"REPSURF_COMPACT": "0"}),
                            ("real_scans_clouds_per_s", ("--data", "real"), None),
                            ("eager_clouds_per_s", ("--no-graph",), None)):
        if key == "real_scans_clouds_per_s" and (args.data == "real" or args.points != 1024):
            continue
        if key == "eager_clouds_per_s" and args.no_graph:
            continue
        rec = child_leg(args, extra, env, steps=60 if key == "eager_clouds_per_s" else 20)
        out[key] = None if rec is None else rec["value"]
        if rec is not None:
            out[key.replace("_clouds_per_s", "_ms_per_step")] = rec["ms_per_step"]
            if "ms_per_step_median" in rec:
                out[key.replace("_clouds_per_s", "_ms_per_step_median")] = rec["ms_per_step_median"]
    if cpu and out.get("dense_clouds_per_s"):
        out["gpu_over_cpu_dense"] = round(out["dense_clouds_per_s"] / cpu["value"], 1)
    return out


TWO_STREAM_NOTE = ("geometry graph beside the network graph: 0 of 24 000 steps (3 x 8 000, tools/pipelined_flake.py) gave another loss than the one-stream "
                   "step on the final tree -- the same tree with compiler-vectorized packed-fp32 code in the geometry kernels: 51 / 125 / 123 of 8 000 "
                   "(profiles/r06/two_stream_validation.txt, DESIGN.md section 6)")

def mlp_hip_split3():
    from repsurf_amd import mlp_hip
    return mlp_hip.gemm_split3()


GEMM_PRODUCTS = {True: "fp32 products as six bf16 MFMAs over three-part operands (fp32 tensors / prologues / accumulation / BatchNorm sums; error against "
                       "fp64 equal to the fp32 MFMA's: profiles/r04/gemm_split3_ab.txt; RS_GEMM_SPLIT3=0 runs the fp32 MFMAs)",
                 False: "v_mfma_f32_32x32x2_f32 (RS_GEMM_SPLIT3=0)"}


def seg_geometry_lines(coord, offset):
    """FPS (us per pick) and the two kNN searches of the first stage, timed with HIP events on the launch stream."""
    from repsurf_amd import ops

    def timed(fn, reps=10):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    new_offset = ops.strided_offset(offset, 4)
    host, new_host = ops.host_offsets(offset), ops.host_offsets(new_offset)
    t = timed(lambda: ops.furthestsampling_offset(coord, offset, new_offset))
    picks = new_host[0] - 1
    fps = {"stage1": {"us": round(t * 1e6, 1), "us_per_pick": round(t * 1e6 / max(picks, 1), 4), "clouds": len(host),
                      "rows_per_cloud": host[0], "picks_per_cloud": new_host[0]}}
    idx = ops.furthestsampling_offset(coord, offset, new_offset)
    centres = coord[idx.long()].contiguous()
    n, m = coord.shape[0], centres.shape[0]
    knn = {}
    for name, k, q, qo in (("umbrella_k9", 9, coord, offset), ("group_k32", 32, centres, new_offset)):
        t = timed(lambda: ops.knnquery_offset(k, coord, q, offset, qo))
        nbytes = 4.0 * (3 * n + 3 * q.shape[0] + 2 * k * q.shape[0])             # xyz + queries + (idx, dist2)
        grid = bool(ops.KNN_GRID and n >= len(host) * ops.KNN_GRID_MIN_ROWS[1 if k <= 16 else 2])
        knn[name] = {"us": round(t * 1e6, 1), "algorithmic_bytes": nbytes, "achieved_GBs": round(nbytes / t / 1e9, 1),
                     "frac_of_hbm": round(nbytes / t / 1e9 / PEAK_HBM_GBS, 5),
                     "method": "per-cloud uniform grid, build + query (rs_knn_grid_*)" if grid else "scan of the whole cloud per query (rs_knnquery_offset)",
                     # what a scan of every row per query would have to sustain to be as fast (the grid evaluates a few hundred rows per query)
                     "scan_equivalent_G_pair_tests_per_s": round(float(q.shape[0]) * host[0] / t / 1e9, 1)}
    return fps, knn


def main_seg(args):
    """BASELINE configs[3]: RepSurf-U S3DIS segmentation (repsurf_umb_ssg), B clouds x P points x (xyz + rgb) per GPU,
    zero_grad -> forward -> cross-entropy -> backward -> Adam, same contract as the classification line."""
    from repsurf_amd import dist as rdist
    rank, world, local = rdist.env()
    if world != args.gpus:         # under a launcher the launcher's world size is the truth (no launcher: main() spawned --gpus ranks)
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if "REPSURF_BENCH_DEVICE" in os.environ:
        local = int(os.environ["REPSURF_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    rdist.init(backend=os.environ.get("REPSURF_DIST_BACKEND", "nccl"), device=device)
    seg_root = os.path.join(ROOT, "repsurf_amd", "segmentation")
    cls_root = os.path.join(ROOT, "repsurf_amd", "classification")
    if cls_root in sys.path:
        sys.path.remove(cls_root)
    sys.path.insert(0, seg_root)
    from repsurf_amd import _lib, mlp
    from repsurf_amd.graph import PipelinedStep
    from repsurf_amd.optim import Adam
    from models.repsurf.repsurf_umb_ssg import Model
    mlp.set_precision(args.dtype)
    clouds = 16 if args.batch == 32 else args.batch            # (--batch default is the classification one)
    pts = 4096 if args.points == 1024 else args.points
    margs = argparse.Namespace(return_polar=False, in_channel=6, group_size=8, num_class=13)
    torch.manual_seed(0)
    model = Model(margs).to(device).train()
    cpu_state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    optim = None if args.no_optim else Adam(model.parameters(), lr=1e-3)
    r = np.random.RandomState(rdist.rank_seed(125, rank))
    n = clouds * pts
    coord_h = (r.rand(n, 3) * 2 - 1).astype(np.float32)
    rgb_h = r.rand(n, 3).astype(np.float32)
    label_h = r.randint(0, 13, n).astype(np.int64)
    off_h = (np.arange(1, clouds + 1) * pts).astype(np.int32)
    from repsurf_amd import ops
    coord, rgb, label = (torch.from_numpy(a).to(device) for a in (coord_h, rgb_h, label_h))
    offset = ops.offsets_tensor(off_h.tolist(), device)
    np.random.seed(rdist.rank_seed(13, rank))                  # numpy generator: the constructor's normal flips
    from repsurf_amd.head import CrossEntropyLoss
    criterion = CrossEntropyLoss(ignore_index=255)             # the train loop's nn.CrossEntropyLoss(ignore_index=255), 3 launches
    inputs = [coord, rgb, offset]

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            rdist.barrier()
        torch.cuda.synchronize()

    if args.ragged:
        # eight different ragged batches (sizes in [pts / 2, pts], mean 3/4 pts), cycled: every step sees other cloud boundaries than the
        # step before -- host offsets, grids, BatchNorm row counts and every intermediate shape change per step
        batches = []
        for i in range(8):
            sizes = r.randint(pts // 2, pts + 1, clouds)
            nn = int(sizes.sum())
            batches.append(([torch.from_numpy((r.rand(nn, 3) * 2 - 1).astype(np.float32)).to(device), torch.from_numpy(r.rand(nn, 3).astype(np.float32)).to(device),
                             ops.offsets_tensor(np.cumsum(sizes).tolist(), device)], torch.from_numpy(r.randint(0, 13, nn).astype(np.int64)).to(device), nn))

        def ragged_step(i):
            inp, lab, _ = batches[i % len(batches)]
            for q in model.parameters():
                q.grad = None
            loss = criterion(model(inp), lab)
            loss.backward()
            if optim is not None:
                optim.step()
            return loss

        def timed_eager(steps):
            for i in range(max(args.warmup, len(batches))):
                ragged_step(i)
            fence()
            t0 = time.perf_counter()
            for i in range(steps):
                loss = ragged_step(i)
            fence()
            return rdist.max_over_ranks(time.perf_counter() - t0, device), loss
        rows = sum(batches[i % len(batches)][2] for i in range(args.steps))
        graphed = world == 1 and not args.no_graph
        if graphed:
            # ONE captured network graph for every batch layout (repsurf_amd.graph.RaggedSegStep: launches sized for a row capacity,
            # counts read from a device table; the next batch's geometry launched eagerly on a side stream under the running graph)
            from repsurf_amd.graph import RaggedSegStep

            def timed_ragged(overlap):
                # (the step restores parameters / statistics / optimizer state after its warm-up passes; both forms start from the same weights
                #  only approximately -- the timed steps train -- which does not matter to the time)
                rs = RaggedSegStep(model, criterion, optim, batches[0][0], batches[0][1], capacity=clouds * pts, warmup=max(2, args.warmup),
                                   max_cloud_rows=pts, overlap=overlap)
                w = max(args.warmup, len(batches))
                for i in range(w):
                    rs(batches[(i + 1) % len(batches)][0], batches[(i + 1) % len(batches)][1], sync=False)
                fence()
                t0 = time.perf_counter()
                for i in range(args.steps):
                    nxt = batches[(w + i + 1) % len(batches)]
                    loss_ = rs(nxt[0], nxt[1], sync=False)
                fence()
                dt_ = time.perf_counter() - t0
                lv = float(loss_.item())
                rs.close()
                return dt_, lv
            # the overlapped form (the default) first: a serialized instance run earlier in the process leaves the two streams on one hardware
            # queue for the instance that follows (measured: 4.67 ms instead of 2.89)
            dt, loss_val = timed_ragged(True)
            serial_dt, _ = timed_ragged(False)
            serial_ms = serial_dt / args.steps * 1e3
            eager_dt, _ = timed_eager(min(args.steps, 16))
            eager_ms = eager_dt / min(args.steps, 16) * 1e3
        else:
            dt, loss = timed_eager(args.steps)
            loss_val, eager_ms = float(loss.item()), None
        if rank == 0:
            launch = ("ONE captured network hipGraph for every batch layout: launches sized for the row capacity, row counts read from a device table "
                      "(include/repsurf_hip.h: rows_dev), the next batch's geometry launched eagerly on a side stream BESIDE the graph "
                      "(repsurf_amd.graph.RaggedSegStep, overlap=True: the default)") if graphed else "eager launches"
            out = {"metric": "point-clouds/sec fwd+bwd, RepSurf-U S3DIS seg, RAGGED packed batches" + ("" if graphed else " (eager launches)"),
                   "value": round(clouds * world * args.steps / dt, 2), "unit": "clouds/s", "points_per_s": round(rows * world / dt), "n_gpus": world,
                   "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                   "vs_baseline": None, "dtype": "f32", "data": "synthetic uniform clouds, 8 different ragged batches cycled, random-init weights",
                   "config": {"workload": f"configs[3] shape with ragged clouds: B={clouds} clouds of {pts // 2}..{pts} points (mean {int(np.mean([b_[2] for b_ in batches]))} rows per batch, "
                                          f"capacity {clouds * pts}): what the reference's loader emits (segmentation/util/data_util.py:15-23)",
                              "launch": launch, "parallelism": f"dp{world}", "optimizer_step": not args.no_optim, "loss": round(loss_val, 5)}}
            if graphed:
                out["serialized_ms_per_step"] = round(serial_ms, 4)      # RaggedSegStep(overlap=False): the geometry BEHIND the graph
            if eager_ms is not None:
                out["eager_ms_per_step"] = round(eager_ms, 4)
                out["eager_clouds_per_s"] = round(clouds * 1e3 / eager_ms, 2)
            print(json.dumps(out), flush=True)
        rdist.finish()
        return
    if args.no_graph:
        # eager launches in program order, one dispatch per instrumented call: what the PMC passes of tools/gpu_profile.sh align
        # with the launch log (--launch-log); not a throughput configuration
        def eager():
            for q in model.parameters():
                q.grad = None
            loss = criterion(model(inputs), label)
            loss.backward()
            if optim is not None:
                optim.step()
            return loss
        for _ in range(args.warmup):
            eager()
        fence()
        _lib.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = eager()
        fence()
        dt = time.perf_counter() - t0
        _lib.profile_enable(False)
        _lib.profile_collect()
        if rank == 0:
            if args.launch_log:
                os.makedirs(os.path.dirname(os.path.abspath(args.launch_log)), exist_ok=True)
                json.dump([[n_, list(d)] for n_, d in _lib.profile_sequence()], open(args.launch_log, "w"))
            print(json.dumps({"metric": "point-clouds/sec fwd+bwd, RepSurf-U S3DIS seg, 4096-pt clouds @ B=16 per GPU (EAGER launches: profiling aid)",
                              "value": round(clouds * world * args.steps / dt, 2), "unit": "clouds/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
                              "config": {"workload": "configs[3], eager launches", "launch": "eager"}, "loss": round(float(loss.item()), 5)}))
        return
    pstep = PipelinedStep(model, criterion, optim, inputs, label, warmup=max(2, args.warmup), sharded=world > 1)
    step = lambda: pstep(sync=False)      # noqa: E731
    mode = "2 hipgraphs on 2 streams: geometry (kNN, FPS, 3-NN weights) of batch s+1 under the network of batch s" + (
        "" if world == 1 else ("; RCCL all-reduce + Adam inside the network graph" if pstep.collective_captured else
                               " + eager rccl all-reduce + Adam graph"))
    steps_timed = timed_step_count(args, step, fence, world, device, rdist)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps_timed):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    timing = not args.no_kernel_timing
    timed_steps = 0
    in_graph = None
    if timing and rank == 0:
        import copy
        twin = copy.deepcopy(model)
        topt = None if args.no_optim else Adam(twin.parameters(), lr=1e-3)

        def eager_step():
            for p in twin.parameters():
                p.grad = None
            criterion(twin(inputs), label).backward()
            if topt is not None:
                topt.step()
        for _ in range(2):
            eager_step()
        timed_steps = min(args.steps, 3)
        _lib.profile_enable(True)
        for _ in range(timed_steps):
            eager_step()
        torch.cuda.synchronize()
        _lib.profile_enable(False)
        in_graph = gemm_family_in_graph(copy.deepcopy(model), criterion, inputs, label)
    dt = rdist.max_over_ranks(dt, device)
    prof = _lib.profile_collect()
    if rank == 0:
        ms = dt / steps_timed * 1e3
        roofline, table = roofline_from_profile(prof, timed_steps, args.dtype, in_graph) if timing else (None, [])
        if args.breakdown:
            os.makedirs(os.path.dirname(os.path.abspath(args.breakdown)), exist_ok=True)
            json.dump({"ms_per_step": ms, "kernels": table}, open(args.breakdown, "w"), indent=1)
        fps_line, knn_line = seg_geometry_lines(coord, offset) if timing else (None, None)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import geom_oracle, seg_ref
            geom_oracle.build()
            threads = min(os.cpu_count() or 1, args.cpu_threads)
            torch.set_num_threads(threads)
            ts = []
            for i in range(1 + args.cpu_steps):
                t1 = time.perf_counter()
                seg_ref.step(cpu_state, coord_h, rgb_h, off_h, label_h, None)
                ts.append(time.perf_counter() - t1)
            cdt = float(np.mean(ts[1:]))
            port = {"value": round(clouds / cdt, 3), "unit": "clouds/s", "cores": torch.get_num_threads(), "kind": "port",
                    "sample": f"{args.cpu_steps} timed steps (after 1 warm-up) of the same {clouds} x {pts}-point batch, fwd + "
                              f"cross-entropy + bwd through oracle/seg_ref.py (torch {torch.__version__} CPU dense ops, C/OpenMP geometry, "
                              f"{threads} threads)",
                    "s_per_step": round(cdt, 4)}
            cpu = port
            ref = reference_seg_cpu_baseline(pts, threads)
            if ref is not None:
                # the reference's own code is the baseline: its unmodified Python over ITS OWN kernels compiled as host code
                # (oracle/_ref); a bounded sample -- 2 of the batch's clouds -- because those kernels run block after block on one core
                cpu = {"value": ref["clouds_per_s"], "unit": "clouds/s", "cores": ref["threads"], "kind": "reference",
                       "sample": f"the reference's own segmentation step ({ref['source']}: segmentation/models/repsurf/repsurf_umb_ssg.py over its modules/, "
                                 f"`pointops_cuda` = the reference's *_cuda_kernel.cu files compiled unmodified as host code, oracle/Makefile.ref), "
                                 f"1 timed step (after 1 warm-up) of {ref['clouds']} x {ref['points']}-point clouds (a bounded sample of the {clouds}-cloud batch: "
                                 f"the host-compiled kernels run single-threaded), zero_grad+fwd+CE+bwd, torch {ref['torch']} CPU dense ops with {ref['threads']} threads",
                       "s_per_step": ref["s_per_step"], "port": port}
        out = {"metric": "point-clouds/sec fwd+bwd, RepSurf-U S3DIS seg, 4096-pt clouds @ B=16 per GPU", "value": round(clouds * world * steps_timed / dt, 2),
               "unit": "clouds/s", "points_per_s": round(n * world * steps_timed / dt), "n_gpus": world, "steps": args.steps, "steps_timed": steps_timed,
               "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32" if args.dtype == "fp32" else "bf16 MFMA operands + bf16 conv-output storage, f32 accumulate / BatchNorm sums / gradients / parameters",
               "data": "synthetic uniform [-1,1]^3 clouds + uniform rgb, random-init weights",
               "config": {"workload": f"configs[3]: RepSurf-U S3DIS segmentation (repsurf_umb_ssg), B={clouds}x{pts}x6 per GPU, {args.dtype}, "
                                      f"encoder + FP decoder + classifier, fwd+CE+bwd" + ("" if args.no_optim else "+Adam step"),
                          "global_batch": clouds * world, "points": pts, "parallelism": f"dp{world}", "launch": mode,
                          "gemm_products": GEMM_PRODUCTS[mlp_hip_split3()] if args.dtype == "fp32" else "bf16 operands",
                          "optimizer_step": not args.no_optim, "loss": round(float(loss.item()), 5)},
               "roofline": roofline, "fps_us_per_pick": fps_line, "knn": knn_line, "cpu_baseline": cpu}
        if cpu:
            out["gpu_over_cpu"] = round(out["value"] / cpu["value"], 1)
        out["arithmetic"] = ARITHMETIC[mlp_hip_split3()] if args.dtype == "fp32" else "bf16 MFMA operands, fp32 accumulate"
        if world == 1 and args.dtype == "fp32" and mlp_hip_split3() and not args.no_alt_arithmetic:
            out["fp32_mfma_ms_per_step"] = alt_arithmetic_ms(args)
        if mode.startswith("2 hipgraphs") and args.dtype == "fp32" and mlp_hip_split3():
            out["two_stream_note"] = TWO_STREAM_NOTE
        print(json.dumps(out), flush=True)
    held = locals().get("pstep")
    pstep = step = None
    rdist.finish(held)           # close() the pipelined step (graphs with recorded collectives go before their communicator), then tear down


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: become `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <the same arguments>` -- one rank per GPU, rank 0 prints the line
    (the reference spawns its ranks from the entry script too: segmentation/tool/train.py:478-484).  Under a launcher
    (RANK / WORLD_SIZE in the environment) this is never reached."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def dry_run(args):
    """--dry-run: the ranks exist, found each other and can reduce; nothing else."""
    from repsurf_amd import dist as rdist
    rank, world, local = rdist.env()
    cuda = torch.cuda.is_available() and torch.cuda.device_count() >= world      # (RCCL refuses two ranks on one device: gloo then)
    device = torch.device("cuda", local) if cuda else torch.device("cpu")
    rdist.init(backend=os.environ.get("REPSURF_DIST_BACKEND", "nccl" if cuda else "gloo"), device=device)
    t = torch.tensor([float(rank + 1)], device=device)
    if world > 1:
        rdist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "requested": args.gpus, "rank_sum": float(t.item()),
                          "backend": dist.get_backend() if dist.is_initialized() else None}), flush=True)
    rdist.finish()


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    if args.dry_run:
        return dry_run(args)
    if args.workload == "seg":
        return main_seg(args)
    from repsurf_amd import dist as rdist
    rank, world, local = rdist.env()
    if world != args.gpus:         # under a launcher the launcher's world size is the truth (no launcher: main() spawned --gpus ranks)
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    # test hooks (tests/test_graph_gpu.py runs the world_size-2 path on a ONE-GPU box): both ranks on one device, gloo
    # carrying the gradient all-reduce.  Never set by the driver's launch.
    if "REPSURF_BENCH_DEVICE" in os.environ:
        local = int(os.environ["REPSURF_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    rdist.init(backend=os.environ.get("REPSURF_DIST_BACKEND", "nccl"), device=device)   # "nccl" is RCCL on ROCm (xGMI inside the node)

    import importlib
    from repsurf_amd import _lib, mlp, ops
    from repsurf_amd.optim import Adam
    from util.utils import SmoothClsLoss
    Model = importlib.import_module(f"models.repsurf.{args.model}").Model
    mlp.set_precision(args.dtype)

    torch.manual_seed(0)                     # identical initial weights on every rank
    model = Model(model_args()).to(device).train()
    cpu_state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    use_graph = not args.no_graph
    # eager launches: DistributedDataParallel (one 64 MB bucket = a single gradient all-reduce);
    # graph replay: the model itself, gradients in one flat buffer, one explicit all-reduce per step
    net = model if use_graph else rdist.wrap(model, device)
    criterion = SmoothClsLoss()
    optim = None if args.no_optim else Adam(model.parameters(), lr=1e-3)
    batch_of = real_batch if args.data == "real" else synthetic_batch
    points, label = batch_of(rdist.rank_seed(125, rank), args.batch, args.points, device)
    torch.manual_seed(rdist.rank_seed(13, rank))   # CPU generator: FPS starts / normal flips differ per rank

    def step():
        for p in model.parameters():
            p.grad = None
        loss = criterion(net(points), label)
        loss.backward()
        if optim is not None:
            optim.step()
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            rdist.barrier()
        torch.cuda.synchronize()

    mode = "eager"
    in_graph = None
    timing = not args.no_kernel_timing
    if use_graph:
        # Capture FIRST (an eager step on the default stream before capture leaves AccumulateGrad nodes on
        # the wrong stream and the capture faults).  Per-launch HIP events cannot be recorded inside a
        # replayed graph, so the kernel timings for `roofline` come from a short eager pass on a copy of
        # the model AFTER the timed region; the throughput comes from graph replay of the identical step.
        from repsurf_amd.graph import GraphedStep, PipelinedStep, ShardedGraphedStep
        if world == 1 and not args.no_pipeline and hasattr(net, "geometry"):
            # every replay = geometry of batch s+1 (side stream) + network of batch s; the loader's next batch is the
            # same synthetic batch here (both input buffers hold it)
            pstep = PipelinedStep(net, criterion, optim, points, label, warmup=max(2, args.warmup))
            step = lambda: pstep(sync=False)      # noqa: E731 - the loop synchronises the device around the timed region
            mode = ("2 hipgraphs on 2 streams: geometry of batch s+1 under the network of batch s; the SAME synthetic batch "
                    "sits in both pipeline buffers (every replay still runs one full geometry pass and one full network pass)")
        elif world == 1:
            step = GraphedStep(net, criterion, optim, points, label, warmup=max(2, args.warmup))
            mode = "hipgraph"
        else:
            try:
                if args.no_pipeline or not hasattr(net, "geometry"):
                    step = ShardedGraphedStep(net, criterion, optim, points, label, warmup=max(2, args.warmup))
                    mode = "hipgraph x2 + rccl all-reduce"
                else:
                    pstep = PipelinedStep(net, criterion, optim, points, label, warmup=max(2, args.warmup), sharded=True)
                    step = lambda: pstep(sync=False)      # noqa: E731
                    mode = ("2 hipgraphs on 2 streams (geometry of batch s+1 under the network of batch s); " +
                            ("the network graph holds forward + backward + gradient pack + RCCL all-reduce + Adam: one replay per rank-step"
                             if pstep.collective_captured else "network graph -> eager rccl all-reduce -> Adam graph"))
            except Exception as e:  # noqa: BLE001 - keep the scaling run alive: eager DDP is slower but equivalent
                print(f"[bench rank {rank}] graph capture failed ({e!r}); falling back to eager DDP", file=sys.stderr)
                for p in model.parameters():
                    p.grad = None
                net = rdist.wrap(model, device)
                use_graph = False
                for _ in range(args.warmup):
                    step()
                if timing:
                    args.timed_steps = args.steps
                    _lib.profile_enable(True)
    else:
        for _ in range(args.warmup):
            step()
        if timing:
            args.timed_steps = args.steps
            _lib.profile_enable(True)
    steps_timed = timed_step_count(args, step, fence, world, device, rdist) if use_graph else args.steps
    fence()
    t0 = time.perf_counter()
    stamps = [t0]
    for _ in range(steps_timed):
        loss = step()
        stamps.append(time.perf_counter())      # (host issue times: what paces an eagerly launched step)
    fence()
    dt = time.perf_counter() - t0
    if timing and not use_graph:
        _lib.profile_enable(False)
    if timing and use_graph and rank == 0:
        import copy
        twin = copy.deepcopy(model)
        topt = None if args.no_optim else Adam(twin.parameters(), lr=1e-3)

        def eager_step():
            for p in twin.parameters():
                p.grad = None
            criterion(twin(points), label).backward()
            if topt is not None:
                topt.step()
        for _ in range(2):
            eager_step()
        args.timed_steps = min(args.steps, 5)
        _lib.profile_enable(True)
        for _ in range(args.timed_steps):
            eager_step()
        torch.cuda.synchronize()
        _lib.profile_enable(False)
        in_graph = gemm_family_in_graph(copy.deepcopy(model), criterion, points, label)      # (a copy no eager step has run on: see the capture note above)
    dt = rdist.max_over_ranks(dt, device)
    allreduce_us = rdist.time_allreduce(sum(p.numel() for p in model.parameters()), device) if world > 1 else None
    if "REPSURF_BENCH_DUMP" in os.environ:       # test hook: every rank's parameters after the timed loop
        torch.cuda.synchronize()
        torch.save(torch.cat([p.detach().flatten() for p in model.parameters()]).cpu(),
                   os.path.join(os.environ["REPSURF_BENCH_DUMP"], f"params_rank{rank}.pt"))
    prof = _lib.profile_collect()            # {abi name: [(ms, dims), ...]} from HIP events on the launch stream

    if rank == 0:
        ms = dt / steps_timed * 1e3
        value = args.batch * world * steps_timed / dt
        roofline, table = roofline_from_profile(prof, getattr(args, "timed_steps", args.steps), args.dtype, in_graph)
        if args.launch_log:
            os.makedirs(os.path.dirname(os.path.abspath(args.launch_log)), exist_ok=True)
            json.dump([[n, list(d)] for n, d in _lib.profile_sequence()], open(args.launch_log, "w"))
        if args.breakdown:
            os.makedirs(os.path.dirname(os.path.abspath(args.breakdown)), exist_ok=True)
            json.dump({"ms_per_step": ms, "kernels": table}, open(args.breakdown, "w"), indent=1)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args, cpu_state)
        # (skipped under --launch-log: a PMC pass matches dispatches to the logged calls, extra launches would shift them)
        ball_line, fps_line = geometry_lines(device, args.points) if (timing and not args.launch_log) else (None, None)
        out = {"metric": "point-clouds/sec fwd+bwd, RepSurf-U 1024-pt cls @ B=32 per GPU", "value": round(value, 2),
               "unit": "clouds/s", "n_gpus": world, "steps": args.steps, "steps_timed": steps_timed, "warmup": args.warmup,
               "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32" if args.dtype == "fp32" else "bf16 MFMA operands + bf16 conv-output storage, f32 accumulate / BatchNorm sums / gradients / parameters",
               "arithmetic": ARITHMETIC[mlp_hip_split3()] if args.dtype == "fp32" else "bf16 MFMA operands, fp32 accumulate",
               "data": ("synthetic uniform [-1,1]^3 clouds" if args.data == "uniform" else
                        "the reference's 4 scanned objects (tests/golden/geom_real.npz) tiled to the batch, rotated + jittered per copy") + ", random-init weights",
               "config": {"workload": f"configs[{1 if args.dtype == 'fp32' else 4}]: RepSurf-U ({args.model}) classifier, B={args.batch}x{args.points} pts "
                                      f"per GPU, {args.dtype}, full encoder + head, fwd+loss+bwd"
                                      + ("" if args.no_optim else "+Adam step"),
                          "global_batch": args.batch * world, "points": args.points,
                          "parallelism": f"dp{world}", "mlp_backend": "hip", "launch": mode,
                          "gemm_products": GEMM_PRODUCTS[mlp_hip_split3()] if args.dtype == "fp32" else "bf16 operands",
                          "grouped_rows": ("compacted: distinct ball-query slots only (exact; DESIGN 6)" if mlp.COMPACT_GROUPS else
                                           "dense: every ball-query slot (REPSURF_COMPACT=0)"),
                          "distinct_slot_fraction_sa1_sa2": distinct_slot_fraction(points),
                          "optimizer_step": not args.no_optim, "loss": round(float(loss.item()), 5),
                          "allreduce_us": None if allreduce_us is None else round(allreduce_us, 1),
                          "allreduce_bytes": 4 * sum(p.numel() for p in model.parameters()) if world > 1 else None},
               "roofline": roofline, "roofline_ballquery": ball_line, "fps_us_per_pick": fps_line, "cpu_baseline": cpu}
        if cpu:
            out["gpu_over_cpu"] = round(value / cpu["value"], 1)
        if not use_graph:
            # an eager step is paced by the host: the median of the per-step issue intervals is the steady state, the mean (ms_per_step)
            # also carries Python's full collections (one 80-90 ms pause every few dozen steps on this image: tools/eager_step_times.py)
            out["ms_per_step_median"] = round(float(np.median(np.diff(stamps))) * 1e3, 4)
        if world > 1:
            out["allreduce_us"] = None if allreduce_us is None else round(allreduce_us, 1)      # (also in config: the step's one data-path collective, alone)
        if world == 1 and args.dtype == "fp32" and mlp_hip_split3() and not args.no_alt_arithmetic:
            out["fp32_mfma_ms_per_step"] = alt_arithmetic_ms(args)      # the same step under RS_GEMM_SPLIT3=0, 30 steps, child process
        if mode.startswith("2 hipgraphs") and args.dtype == "fp32" and mlp_hip_split3():
            out["two_stream_note"] = TWO_STREAM_NOTE
        if world == 1 and args.dtype == "fp32" and use_graph and mlp.COMPACT_GROUPS and not args.no_extra_legs:
            out.update(bracketing_legs(args, value, cpu))
        print(json.dumps(out), flush=True)
    held = locals().get("pstep")
    pstep = step = None
    rdist.finish(held)           # close() the pipelined step (graphs with recorded collectives go before their communicator), then tear down


def timed_step_count(args, step, fence, world, device, rdist):
    """max(--steps, ceil(--min-seconds / step time)): the step time comes from a short untimed probe (also warm-up), the
    count is agreed over the ranks so that every rank replays the same number of steps."""
    fence()
    t0 = time.perf_counter()
    probe = max(3, min(args.steps, 10))
    for _ in range(probe):
        step()
    fence()
    est = rdist.max_over_ranks((time.perf_counter() - t0) / probe, device)
    return max(args.steps, int(np.ceil(args.min_seconds / max(est, 1e-6))))


def traffic_key(kernel, dims):
    return kernel + "|" + ",".join(str(d) for d in dims if not (isinstance(d, str) and d.startswith("rows=")))


def traffic_from_profiles(kernel, dims):
    """HBM bytes per launch of this ABI call at these sizes, from the rocprofv3 PMC passes
    (FETCH_SIZE and WRITE_SIZE in separate runs, gfx950 FETCH_SIZE x2 correction) that
    tools/traffic_from_pmc.py summarises into profiles/traffic.json; None until collected."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(path):
        try:
            rec = json.load(open(path)).get(traffic_key(kernel, dims))
            return rec["hbm_bytes"] if rec else None
        except (OSError, ValueError, KeyError):
            return None
    return None


if __name__ == "__main__":
    main()
