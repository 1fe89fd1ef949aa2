# kernel-trace stats of the FLAC pipeline at one stream count: bash tools/r05_flac_prof.sh <streams> <tag>
export TMPDIR=/tmp; S=${1:-2048}; TAG=${2:-flac}
OUT=$PWD/gpurun_out/r05f; mkdir -p $OUT; RAW=/tmp/raw_$TAG; rm -rf $RAW
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o st -- python3 bench.py --workload flac_pipeline --streams $S --steps 10 --warmup 2 --extra-windows 0 --cpu-streams 0 > $OUT/${TAG}_bench.log 2>&1
f=$(find $RAW -name "*kernel_stats.csv" | head -1); cp $f $OUT/${TAG}_kernel_stats.csv
python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=0
for r in rows:
    if 'aukit' in r['Name']:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f}")
PY
grep -o '"ms_per_step": [0-9.]*' $OUT/${TAG}_bench.log | tail -1
python3 tools/trace_overlap.py $RAW k_flac_find 2>/dev/null | tail -30
