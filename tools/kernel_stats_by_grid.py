#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> average duration per (kernel template instance, grid size): the --stats summary mixes all
shapes an instance ran (e.g. the pooled data gradient of the 4 096-row and of the 48 k-row stage), which is not what
bench.py's `roofline.avg_launch_us` (one launch class = one shape) should be compared with.
    python tools/kernel_stats_by_grid.py gpurun_out/prof_r02_cls/graph_kernel_trace.csv > profiles/r02/cls_graph_kernel_stats_by_grid.csv"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
acc = defaultdict(list)
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*", "", name)
    grid = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])),
            int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_Z"])))
    acc[(name, grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "workgroups_x", "workgroups_y", "workgroups_z", "calls", "avg_us", "min_us", "max_us", "total_ms"])
for (name, grid), ds in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    w.writerow([name, *grid, len(ds), round(sum(ds) / len(ds) / 1e3, 2), round(min(ds) / 1e3, 2), round(max(ds) / 1e3, 2), round(sum(ds) / 1e6, 3)])
