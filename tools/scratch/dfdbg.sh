timeout 300 python -m pytest tests -m gpu -x -q -k "config4 or fused" 2>&1 | tail -1
for d in ${DBGS:-0 8 4 12}; do
echo "dbg $d"; AUKIT_DFPWM_STATS=1 AUKIT_DFPWM_FUSED_DBG=$d timeout 120 python bench.py --workload dfpwm_transcode --steps 4 --warmup 1 --cpu-streams 0 --extra-windows 0 > /tmp/o.txt 2>&1; grep -a "clock probe" /tmp/o.txt | tail -2; grep -a slowest /tmp/o.txt | tail -1 | sed -e 's/.*workgroups, //'; tail -1 /tmp/o.txt | python3 -c "
import sys,json
print('ms/step', round(json.loads(sys.stdin.read())['ms_per_step'],3))"
done
