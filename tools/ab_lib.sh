#!/bin/bash
# A/B two builds of the library on the same box: aukit_amd/libaukit_hip_old.so vs the current one (headline bench, 3 runs each, interleaved)
cp aukit_amd/libaukit_hip.so /tmp/new.so
for i in 1 2 3; do
  for v in old new; do
    if [ $v = old ]; then cp aukit_amd/libaukit_hip_old.so aukit_amd/libaukit_hip.so; else cp /tmp/new.so aukit_amd/libaukit_hip.so; fi
    python bench.py --steps 20 --warmup 3 --cpu-streams 0 "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e3,1), 'Gsamples/s', d['roofline']['kernel'], round(d['roofline']['frac'],3))"
  done
done
cp /tmp/new.so aukit_amd/libaukit_hip.so
