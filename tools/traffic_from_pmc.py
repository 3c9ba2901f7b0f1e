#!/usr/bin/env python3
"""HBM traffic per launch of each instrumented ABI call, from rocprofv3 PMC passes.

Inputs (all produced by tools/gpu_profile.sh on the GPU box):
  * one launch log per PMC pass -- `bench.py --no-graph --launch-log X.json` writes the ordered list of
    (abi call, sizes) of its timed steps;
  * the matching `*_counter_collection.csv` of `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`
    (separate passes: the TCC counters do not fit in one, MI355X_MICROARCH.md "HBM").
Counter collection serialises dispatches and keeps their order, and every ABI call below launches exactly
one kernel of its family, so the last L dispatches of a family line up 1:1 with the L log entries.

Corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE is reported in KB and, on gfx950, counts 64 B per
128-B request for wide coalesced reads -> x2.  WRITE_SIZE (KB) is used as reported (uncalibrated).
Output: profiles/traffic.json  {"<abi>|<sizes>": {"hbm_bytes", "read_bytes", "write_bytes", "launches",
"kernel"}} -- bench.py's roofline.traffic looks its dominant launch up here.
"""
import argparse
import collections
import csv
import json

import re

# ABI call -> the kernels it can launch (exactly ONE dispatch of this set per call).  Round 1 matched "wgrad_kernel" /
# "gemm_rows_kernel" only: calls that took the narrow streaming kernels (wgrad_small_kernel, gemm_small_kernel) had no
# dispatch of their own, the 1:1 alignment slipped and several size classes showed the same bytes.
FAMILY = {
    "rs_mlp_gemm_rows": r"gemm_rows_kernel|gemm_small_kernel|gemm_narrow_kernel",
    "rs_mlp_wgrad": r"wgrad_kernel|wgrad_small_kernel|wgrad_narrow_kernel",
    "rs_ballquery": r"ballquery_kernel|ballquery_grid_kernel",
    "rs_furthestsampling": r"fps_reg_kernel|fps_global_kernel",
    "rs_umbrella_features": r"umbrella_kernel",
    "rs_pool_max": r"pool_max_kernel|pool_max_long_kernel",
    "rs_pool_max_backward": r"pool_max_bwd_kernel",
    "rs_group_features_compact": r"compact_features_kernel|compact_features4_kernel",
    "rs_group_features_compact_backward": r"compact_scatter_kernel|compact_scatter4_kernel",
}


def key(name, dims):
    """sizes + operand / epilogue modes (same key as bench.traffic_key); the per-launch "rows=<n>" note is dropped"""
    return name + "|" + ",".join(str(d) for d in dims if not (isinstance(d, str) and d.startswith("rows=")))


def per_family(csv_path, counter):
    fam = collections.defaultdict(list)
    with open(csv_path) as f:
        rows = [r for r in csv.DictReader(f) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        for abi, sub in FAMILY.items():
            if re.search(r"(?<![a-z_])(" + sub + r")", r["Kernel_Name"]):
                name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
                fam[abi].append((float(r["Counter_Value"]), name.split("(")[0]))
    return fam


def collect(log_path, csv_path, counter, warm_steps, logged_steps):
    """The run did `warm_steps` unlogged steps, then `logged_steps` logged ones (bench.py --warmup W --steps K --no-graph);
    anything the process launched afterwards (bench.py's geometry micro-timings) comes later in dispatch order.  The
    logged calls of a family are therefore its dispatches [per_step * W, per_step * (W + K))."""
    log = json.load(open(log_path))
    fam = per_family(csv_path, counter)
    out = collections.defaultdict(list)
    for abi in FAMILY:
        calls = [(n, d) for n, d in log if n == abi]
        disp = fam.get(abi, [])
        if not calls:
            continue
        if len(calls) % logged_steps:
            print(f"warning: {abi}: {len(calls)} logged calls are not a multiple of {logged_steps} steps, skipped")
            continue
        per_step = len(calls) // logged_steps
        lo = per_step * warm_steps
        if len(disp) < lo + len(calls):
            print(f"warning: {abi}: {len(disp)} dispatches < {lo} + {len(calls)}, skipped")
            continue
        for (n, d), (v, kname) in zip(calls, disp[lo:lo + len(calls)]):
            out[key(n, d)].append((v, kname))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch-log", required=True)
    ap.add_argument("--fetch-csv", required=True)
    ap.add_argument("--write-log", required=True)
    ap.add_argument("--write-csv", required=True)
    ap.add_argument("--out", default="profiles/traffic.json")
    ap.add_argument("--warm-steps", type=int, default=1)
    ap.add_argument("--logged-steps", type=int, default=2)
    a = ap.parse_args()
    rd = collect(a.fetch_log, a.fetch_csv, "FETCH_SIZE", a.warm_steps, a.logged_steps)
    wr = collect(a.write_log, a.write_csv, "WRITE_SIZE", a.warm_steps, a.logged_steps)
    res = {}
    for k in sorted(set(rd) | set(wr)):
        r = [v for v, _ in rd.get(k, [])]
        w = [v for v, _ in wr.get(k, [])]
        read_b = 2.0 * 1024.0 * sum(r) / len(r) if r else None        # KB -> B, gfx950 x2
        write_b = 1024.0 * sum(w) / len(w) if w else None
        kname = (rd.get(k) or wr.get(k))[0][1]
        res[k] = {"hbm_bytes": (read_b or 0.0) + (write_b or 0.0), "read_bytes": read_b, "write_bytes": write_b,
                  "launches": max(len(r), len(w)), "kernel": kname,
                  "note": "FETCH_SIZE KB x2 (gfx950 correction) + WRITE_SIZE KB; separate rocprofv3 --pmc passes"}
    json.dump(res, open(a.out, "w"), indent=1, sort_keys=True)
    print(f"{len(res)} launch classes -> {a.out}")


if __name__ == "__main__":
    main()
