#!/bin/bash
# tools/ab_env.sh VAR=VALUE [bench args] — same-box A/B of one environment switch: N interleaved bench runs with and without it
SW=$1; shift
N=${AB_RUNS:-5}
for i in $(seq $N); do
  for v in off on; do
    if [ $v = on ]; then r=$(env $SW python bench.py --steps 40 --warmup 5 --cpu-streams 0 "$@" 2>/dev/null | tail -1); else r=$(python bench.py --steps 40 --warmup 5 --cpu-streams 0 "$@" 2>/dev/null | tail -1); fi
    echo "$v $(echo "$r" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'median', round(d.get('windows',{}).get('median',0),4), round(d['roofline']['frac'],4), d['roofline'].get('kernel','')[:70])")"
  done
done
