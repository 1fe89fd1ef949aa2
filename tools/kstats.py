#!/usr/bin/env python3
"""Print our kernels' rows of a rocprofv3 --kernel-trace --stats output directory.  usage: kstats.py DIR [max rows]"""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "aukit" in r["Name"]]
    for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
        print("%-86s %5s x %8.3f ms" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e6))
