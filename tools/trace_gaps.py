#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV: for every kernel, the gap between the latest end of
any earlier kernel and its start (0 when it overlaps one), summed per step; plus the busy time of the union of all kernel
intervals.  usage: trace_gaps.py <kernel_trace.csv> <steps-kernel-name-substring>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "adam_kernel"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
steps = sum(1 for e in ev if key in e[2])
# steady-state window: from the 5th occurrence of the step kernel to the last
marks = [e[1] for e in ev if key in e[2]]
t0, t1 = marks[4], marks[-1]
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
nsteps = len(marks) - 5
busy = 0; gap = 0; cur_end = t0; gaps = []
for s, e, n in win:
    if s > cur_end:
        gap += s - cur_end; gaps.append((s - cur_end, n))
    else:
        pass
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e
print(f"steps in window {nsteps}; wall {(t1 - t0) / nsteps / 1e3:.1f} us/step; union busy {busy / nsteps / 1e3:.1f} us/step; idle {gap / nsteps / 1e3:.1f} us/step; "
      f"kernels/step {len(win) / nsteps:.1f}; sum of durations {sum(e - s for s, e, _ in win) / nsteps / 1e3:.1f} us/step")
import collections
by = collections.Counter(); cnt = collections.Counter()
for g, n in gaps:
    by[n.split("(")[0][:60]] += g; cnt[n.split("(")[0][:60]] += 1
print("largest idle-before-kernel contributors (us/step, count/step, avg us):")
for n, g in by.most_common(25):
    print(f"  {n:60s} {g / nsteps / 1e3:7.1f} {cnt[n] / nsteps:5.1f} {g / cnt[n] / 1e3:6.2f}")
