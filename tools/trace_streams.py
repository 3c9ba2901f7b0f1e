#!/usr/bin/env python3
"""Per-queue view of a rocprofv3 --kernel-trace CSV of the pipelined step (two graphs replaying side by side): for the queue that
runs the network (the one with the step kernel) the sum of kernel durations, the idle time between its kernels and, for every
kernel name, how much of its duration overlaps kernels of the OTHER queues; for the other queues their busy time.
usage: trace_streams.py <kernel_trace.csv> [step-kernel-substring]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "adam_kernel"
qcol = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r[qcol]) for r in rows]
ev.sort()
marks = [e[1] for e in ev if key in e[2]]
mainq = collections.Counter(e[3] for e in ev if key in e[2]).most_common(1)[0][0]
t0, t1 = marks[4], marks[-1]
n = len(marks) - 5
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
main = [e for e in win if e[3] == mainq]
other = [e for e in win if e[3] != mainq]
dur = sum(e[1] - e[0] for e in main)
gap = sum(max(0, b[0] - a[1]) for a, b in zip(main, main[1:]))
print(f"window: {n} steps, wall {(t1 - t0) / n / 1e3:.1f} us/step; network queue {mainq}: {len(main) / n:.1f} kernels/step, durations {dur / n / 1e3:.1f} us/step, idle between its kernels {gap / n / 1e3:.1f} us/step")
byq = collections.defaultdict(int)
for e in other: byq[e[3]] += e[1] - e[0]
for q, d in byq.items(): print(f"  queue {q}: busy {d / n / 1e3:.1f} us/step, {sum(1 for e in other if e[3] == q) / n:.1f} kernels/step")
# overlap of every network kernel with kernels of the other queues
other.sort()
import bisect
starts = [e[0] for e in other]
agg = collections.defaultdict(lambda: [0, 0, 0])
for s, e, name, _ in main:
    ov = 0
    i = max(0, bisect.bisect_left(starts, s) - 64)
    while i < len(other) and other[i][0] < e:
        ov += max(0, min(e, other[i][1]) - max(s, other[i][0])); i += 1
    k = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    a = agg[k]; a[0] += 1; a[1] += e - s; a[2] += min(ov, e - s)
print("network kernels: calls/step, avg us, share of their time beside a kernel of another queue")
for k, (c, d, o) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"  {k:70s} {c / n:5.1f} x {d / c / 1e3:8.1f}   {100.0 * o / d:5.1f} %")
# the other queues' kernels
agg2 = collections.defaultdict(lambda: [0, 0])
for s, e, name, _ in other:
    k = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    agg2[k][0] += 1; agg2[k][1] += e - s
print("kernels of the other queues: calls/step, avg us, us/step")
for k, (c, d) in sorted(agg2.items(), key=lambda kv: -kv[1][1])[:20]:
    print(f"  {k:70s} {c / n:5.1f} x {d / c / 1e3:8.1f} = {d / n / 1e3:8.1f}")
