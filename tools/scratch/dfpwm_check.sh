timeout 600 python -m pytest tests -m gpu -x -q -k "dfpwm or config4 or config_4 or fused" 2>&1 | tail -5
AUKIT_DFPWM_STATS=1 timeout 120 python bench.py --workload dfpwm_transcode --steps 3 --warmup 1 --cpu-streams 0 --extra-windows 0 2>&1 | grep -a "dfpwm fused" | tail -1
rm -rf /tmp/df
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/df -o df -- python3 bench.py --workload dfpwm_transcode --steps 5 --warmup 1 --cpu-streams 0 --extra-windows 0 > /dev/null 2>&1
python3 tools/kstats.py /tmp/df 12
timeout 120 python bench.py --workload dfpwm_transcode --steps 10 --warmup 1 --cpu-streams 0 2>&1 | tail -1 | cut -c1-200
AUKIT_DFPWM_FUSED=0 timeout 120 python bench.py --workload dfpwm_transcode --steps 10 --warmup 1 --cpu-streams 0 2>&1 | tail -1 | cut -c1-200
