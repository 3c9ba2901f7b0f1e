#!/usr/bin/env python3
"""Summarise the rocprofv3 PMC passes of tools/gpu_profile.sh into one CSV (profiles/<tag>/pmc_summary.csv):
per kernel instance + grid: launches, FETCH_SIZE/WRITE_SIZE (KB as reported, and HBM MB with the gfx950 x2 read
correction), MFMA-busy cycles, GRBM_GUI_ACTIVE, and the MFMA utilisation
    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD * 256 CU) / (GRBM_GUI_ACTIVE / 8 XCD)."""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(d, "pmc_*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = nm.split("(")[0][-80:] + " grid=" + r["Grid_Size"]
        a = agg[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"] or 0)
        a[1] += 1
rows = []
for k, c in agg.items():
    def avg(n):
        return c[n][0] / c[n][1] if n in c and c[n][1] else 0.0
    n = max(v[1] for v in c.values())
    gui = avg("GRBM_GUI_ACTIVE")
    mf = avg("SQ_VALU_MFMA_BUSY_CYCLES")
    util = mf / 1024.0 / (gui / 8.0) if gui else 0.0
    # TCC_EA_RDREQ counts read requests to HBM, TCC_EA_RDREQ_32B the 32-byte ones among them (the others are 64 B):
    # bytes = 32 * RDREQ_32B + 64 * (RDREQ - RDREQ_32B) -- an independent check of the FETCH_SIZE x2 correction
    rd, rd32 = avg("TCC_EA_RDREQ"), avg("TCC_EA_RDREQ_32B")
    rows.append((gui * n, k, n, avg("FETCH_SIZE"), avg("FETCH_SIZE") * 2 / 1024, avg("WRITE_SIZE"), avg("WRITE_SIZE") / 1024,
                 mf, gui, util, rd, rd32, (32.0 * rd32 + 64.0 * (rd - rd32)) / 1e6))
rows.sort(reverse=True)
w = csv.writer(open(os.path.join(d, "pmc_summary.csv"), "w"))
w.writerow(["kernel", "launches", "FETCH_SIZE_KB_avg", "hbm_read_MB_corrected(2x)", "WRITE_SIZE_KB_avg", "hbm_write_MB",
            "SQ_VALU_MFMA_BUSY_CYCLES_avg", "GRBM_GUI_ACTIVE_avg(8 XCD sum)", "mfma_util", "TCC_EA_RDREQ_avg", "TCC_EA_RDREQ_32B_avg",
            "hbm_read_MB_from_RDREQ(32B*n32+64B*rest)"])
for _, k, n, f, fm, wr, wm, mf, gui, util, rd, rd32, rdmb in rows:
    if "rocclr" in k or "at::native" in k or "elementwise" in k:
        continue
    w.writerow([k, n, round(f, 1), round(fm, 1), round(wr, 1), round(wm, 1), int(mf), int(gui), round(util, 3), int(rd), int(rd32), round(rdmb, 1)])
print("rows", len(rows))
