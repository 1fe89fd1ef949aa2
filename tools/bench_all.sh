#!/bin/bash
# runs every bench workload once and prints a compact summary (GPU box)
for w in pcm16_cubic g711_cubic ima_stream dfpwm_transcode; do
  python bench.py --workload $w --steps ${STEPS:-10} --warmup 2 --cpu-streams 0 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']
    print('$w', 'Gsamples/s=%.1f'%(d['value']/1e3), 'ms=%.3f'%r['kernel_ms'], 'GB/s=%.0f'%r['achieved'], r['kernel'])
except Exception as e: print('$w FAILED', e)
"
done
