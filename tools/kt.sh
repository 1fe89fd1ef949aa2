#!/bin/bash
# tools/kt.sh SCRIPT [args...] — rocprofv3 kernel-trace of any python script → per-kernel average times of aukit kernels (GPU box).
# Bounded: the profiled run is wrapped in `timeout` (KT_TIMEOUT seconds, default 300) so a stuck run cannot eat the gpurun budget.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
RAW=/tmp/aukit_kt_$$; rm -rf $RAW; mkdir -p $RAW
timeout ${KT_TIMEOUT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o kt -- python3 "$@" > $RAW/log.txt 2>&1 < /dev/null
echo "rc=$? $(tail -1 $RAW/log.txt | cut -c1-200)"
python3 - "$RAW" <<'PY'
import csv, glob, sys, collections
tot = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "aukit" not in k: continue
        k = k[:110]
        d = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
        t = tot.setdefault(k, [0, 0.0]); t[0] += 1; t[1] += d
for k, (n, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{ms/n:10.3f} ms avg x{n:4d}  {k}")
PY
rm -rf $RAW
