#!/usr/bin/env python3
"""One step's kernels on a time line (development aid): from a rocprofv3 --kernel-trace csv, the kernels between two
consecutive k_pre launches -- start offset, duration, queue -- and the gaps in which nothing runs.
usage: step_timeline.py <dir with *kernel_trace.csv> [step index, default: the last complete one]"""
import csv
import glob
import sys

d = sys.argv[1]
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
pre = [i for i, r in enumerate(rows) if "k_pre<" in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(pre) - 2
a, b = pre[k], pre[k + 1]
t0 = rows[a][0]
print("step %d of %d: %.1f us from k_pre to the next k_pre" % (k, len(pre) - 1, (rows[b][0] - t0) / 1e3))
busy_until = t0
for s, e, n, q in rows[a:b]:
    name = n.split("(")[0].replace("void ", "")[:44]
    gap = (s - busy_until) / 1e3
    print("  %9.1f us  +%8.1f us  q%-3s %-44s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, name, ("   <- nothing ran for %.1f us" % gap) if gap > 1.0 else ""))
    busy_until = max(busy_until, e)
print("  idle before the next k_pre: %.1f us" % ((rows[b][0] - busy_until) / 1e3))
