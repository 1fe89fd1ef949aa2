#!/bin/bash
# A/B two builds of the library on the same box: aukit_amd/libaukit_hip_old.so vs the current one.
# Headline bench, N interleaved runs each (default 6), prints every run and the medians.  Extra args go to bench.py.
N=${AB_RUNS:-6}
cp aukit_amd/libaukit_hip.so /tmp/new.so
: > /tmp/ab_old.txt; : > /tmp/ab_new.txt
for i in $(seq $N); do
  for v in old new; do
    if [ $v = old ]; then cp aukit_amd/libaukit_hip_old.so aukit_amd/libaukit_hip.so; else cp /tmp/new.so aukit_amd/libaukit_hip.so; fi
    python bench.py --steps 40 --warmup 5 --cpu-streams 0 "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e3,1))" | tee -a /tmp/ab_$v.txt | sed "s/^/$v /"
  done
done
cp /tmp/new.so aukit_amd/libaukit_hip.so
python3 - <<'PY'
import statistics
for v in ("old", "new"):
    x = [float(l) for l in open(f"/tmp/ab_{v}.txt")]
    print(v, "median", statistics.median(x), "max", max(x), "min", min(x), "Gsamples/s")
PY
