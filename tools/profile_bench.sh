#!/bin/bash
# rocprofv3 passes over the headline bench (run on the GPU box via gpurun).
# Keeps only small summaries under gpurun_out/prof/: kernel stats + per-dispatch PMC rows of our kernels.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
RAW=/tmp/aukit_prof_raw
OUT=gpurun_out/prof
rm -rf $RAW; mkdir -p $RAW $OUT
ARGS="bench.py --steps 5 --warmup 1 --cpu-streams 0 $BENCH_EXTRA"
TAG=${PROF_TAG:-r1}

rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats -o stats -- python3 $ARGS > $OUT/${TAG}_stats.log 2>&1
find $RAW/stats -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
# per-dispatch durations of our kernels only
python3 - "$RAW/stats" "$OUT/${TAG}_kernel_trace_aukit.csv" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        r = csv.DictReader(fh)
        for row in r:
            if "aukit" in row.get("Kernel_Name", ""):
                rows.append({"Kernel_Name": row["Kernel_Name"][:120], "dur_us": (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3,
                             "VGPR": row.get("VGPR_Count", ""), "SGPR": row.get("SGPR_Count", ""), "LDS": row.get("LDS_Block_Size", ""),
                             "grid": row.get("Grid_Size", ""), "wg": row.get("Workgroup_Size", "")})
with open(sys.argv[2], "w", newline="") as fh:
    if rows:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
PY

# kernel timeline of the last step (which kernels overlap on which queue), when the caller names the step's first kernel
[ -n "$TIMELINE_PAT" ] && python3 tools/trace_overlap.py $RAW/stats "$TIMELINE_PAT" > $OUT/${TAG}_timeline.txt

pass_pmc() {  # $1 = name, rest = counters
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $RAW/$name -o $name -- python3 $ARGS > $OUT/${TAG}_$name.log 2>&1
  python3 - "$RAW/$name" "$OUT/${TAG}_$name.csv" <<'PY'
import csv, glob, sys, collections
acc = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "aukit" not in row.get("Kernel_Name", ""):
                continue
            key = (row["Dispatch_Id"], row["Kernel_Name"][:100])
            acc.setdefault(key, {})[row["Counter_Name"]] = float(row["Counter_Value"])
with open(sys.argv[2], "w", newline="") as fh:
    names = sorted({c for v in acc.values() for c in v})
    w = csv.writer(fh); w.writerow(["dispatch", "kernel"] + names)
    for (d, k), v in acc.items():
        w.writerow([d, k] + [v.get(n, "") for n in names])
PY
  rm -rf $RAW/$name
}
pass_pmc pmc_sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
pass_pmc pmc_fetch FETCH_SIZE
pass_pmc pmc_write WRITE_SIZE
pass_pmc pmc_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU
pass_pmc pmc_tcc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum
rm -rf $RAW
ls -la $OUT
