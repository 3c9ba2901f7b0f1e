#!/usr/bin/env python3
"""Substitute the @NAME@ placeholders of DESIGN.md with the numbers of the committed evidence lines (profiles/r06/*.json, *.csv)."""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles", "r06")


def line(name):
    return json.loads(open(os.path.join(P, name)).read().strip().splitlines()[-1])


def fmt(v, nd=0):
    if nd == 0:
        return f"{int(round(v)):,}".replace(",", " ")
    return f"{v:.{nd}f}"


c, seg, rag, s3 = line("bench_cls.json"), line("bench_seg.json"), line("bench_seg_ragged.json"), line("bench_seg_ragged_s3dis.json")
vals = {
    "CLS_VALUE": fmt(c["value"]), "CLS_MS": fmt(c["ms_per_step"], 4), "CLS_FP32": fmt(c["fp32_mfma_ms_per_step"], 3),
    "CLS_DENSE": fmt(c["dense_clouds_per_s"]), "CLS_REAL": fmt(c["real_scans_clouds_per_s"]), "CLS_EAGER": fmt(c["eager_clouds_per_s"]),
    "CLS_EAGER_MED": fmt(c.get("eager_ms_per_step_median", c["eager_ms_per_step"]), 2),
    "CLS_NOPIPE": fmt(line("bench_cls_nopipe.json")["value"]), "CLS_2X": fmt(line("bench_cls_2x.json")["value"]), "CLS_BF16": fmt(line("bench_cls_bf16_b64.json")["value"]),
    "SEG_VALUE": fmt(seg["value"]), "SEG_MS": fmt(seg["ms_per_step"], 3),
    "RAG_MS": fmt(rag["ms_per_step"], 2), "RAG_SER_MS": fmt(rag["serialized_ms_per_step"], 2), "RAG_EAGER": fmt(rag["eager_ms_per_step"], 2),
    "RAG_PTS": fmt(rag["points_per_s"] / 1e6, 1),
    "S3_MS": fmt(s3["ms_per_step"], 1), "S3_SER_MS": fmt(s3["serialized_ms_per_step"], 1),
    "CPU_VALUE": fmt(c["cpu_baseline"]["value"], 1), "G_OVER_C": fmt(c["gpu_over_cpu"]), "G_OVER_C_DENSE": fmt(c["gpu_over_cpu_dense"]),
    "DOM_FRAC": fmt(c["roofline"]["frac"], 3), "DOM_US": fmt(c["roofline"]["avg_launch_us"], 1),
}
rows = list(csv.DictReader(open(os.path.join(P, "cls_graph_kernel_stats_by_grid.csv"))))
steps = max(int(r["calls"]) for r in rows if "head_out_fwd" in r["kernel"])
fam = sum(float(r["total_ms"]) for r in rows if r["kernel"].startswith(("gemm_", "wgrad_"))) / steps * 1e3
vals["FAM_US"] = fmt(fam)
vals["FAM_FRAC"] = fmt(42.58e9 / (fam * 1e-6) / 1e12 / 157.3, 3)
for r in csv.DictReader(open(os.path.join(P, "cls_pmc_summary.csv"))):
    if "gemm_rows_kernel<64, 64, 4, 6" in r["kernel"]:
        vals["DOM_BUSY"] = r["mfma_util"]
s = open(os.path.join(ROOT, "DESIGN.md")).read()
missing = sorted(set(re.findall(r"@([A-Z0-9_]+)@", s)) - set(vals))
if missing:
    sys.exit(f"no value for {missing}")
for k, v in vals.items():
    s = s.replace(f"@{k}@", str(v))
open(os.path.join(ROOT, "DESIGN.md"), "w").write(s)
print(vals)
