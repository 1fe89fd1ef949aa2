#!/usr/bin/env python3
"""Print the kernel timeline (start / end, ms relative to the first kernel shown) of the LAST bench step from a rocprofv3 --kernel-trace
directory: shows whether kernels on different HIP streams really overlap.  usage: trace_overlap.py DIR [pattern]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "aukit" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id", "?")))
rows.sort()
pat = sys.argv[2] if len(sys.argv) > 2 else "blockmaps"
starts = [i for i, r in enumerate(rows) if pat in r[2]]
if starts:
    rows = rows[starts[-1]:]
t0 = rows[0][0]
for s, e, k, q in rows[:40]:
    print(f"{(s - t0) / 1e6:9.3f} .. {(e - t0) / 1e6:9.3f} ms  q{q:>3}  {k}")
