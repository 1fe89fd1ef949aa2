#!/bin/bash
# tools/ab_libs.sh "name1 name2 ..." [bench args] — interleaved bench runs of tools/variants/libaukit_<name>.so ("base" = the built library)
NAMES=$1; shift
N=${AB_RUNS:-3}
for i in $(seq $N); do
  for v in $NAMES; do
    if [ $v = base ]; then L=aukit_amd/libaukit_hip.so; else L=tools/variants/libaukit_$v.so; fi
    r=$(AUKIT_LIB=$L python bench.py --steps 40 --warmup 5 --cpu-streams 0 "$@" 2>/dev/null | tail -1)
    echo "$v $(echo "$r" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'median', round(d.get('windows',{}).get('median',0),4), round(d['roofline']['frac'],4), d['roofline'].get('kernel','')[:70])")"
  done
done
