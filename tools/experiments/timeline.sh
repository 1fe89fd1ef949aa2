#!/bin/bash
# tools/experiments/timeline.sh WORKLOAD [bench args] — start / end / queue of every aukit kernel of the last two steps of a short bench run (GPU box)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
RAW=/tmp/aukit_tl_$$; rm -rf $RAW; mkdir -p $RAW
timeout ${KT_TIMEOUT:-250} rocprofv3 --kernel-trace --output-format csv -d $RAW -o tl -- python3 bench.py --workload "$@" --steps 6 --warmup 2 --cpu-streams 0 > $RAW/log.txt 2>&1 < /dev/null
python3 - "$RAW" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "aukit" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0].replace("void aukit::", "").replace("aukit::", "")[:40]))
rows.sort()
# the last two occurrences of the step's anchor (k_flac_find or the first kernel name seen twice)
t0 = rows[-1][1]
sel = [r for r in rows if r[0] > t0 - 30_000_000]
base = sel[0][0]
for s, e, q, k in sel:
    print(f"{(s-base)/1e6:9.3f} -> {(e-base)/1e6:9.3f} ms  q{q}  {k}")
PY
rm -rf $RAW
