# kernel-trace stats of the DFPWM transcode at one stream count: bash tools/r05_dfx_prof.sh <streams> <tag>   (extra env passes through)
export TMPDIR=/tmp; S=${1:-2048}; TAG=${2:-dfx}
OUT=$PWD/gpurun_out/r05c; mkdir -p $OUT; RAW=/tmp/raw_$TAG; rm -rf $RAW
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o st -- python3 bench.py --workload dfpwm_transcode --streams $S --steps 10 --warmup 2 --extra-windows 0 --cpu-streams 0 > $OUT/${TAG}_bench.log 2>&1
f=$(find $RAW -name "*kernel_stats.csv" | head -1); cp $f $OUT/${TAG}_kernel_stats.csv
python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if 'df' in r['Name'] or 'dfx' in r['Name']:
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} min {float(r['MinNs'])/1e3:9.1f} max {float(r['MaxNs'])/1e3:9.1f}")
PY
tail -1 $OUT/${TAG}_bench.log | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('ms_per_step', j['ms_per_step'])"
