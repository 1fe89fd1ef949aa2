timeout 150 python -m pytest tests -m gpu -x -q -k "flac or fuzz or golden" 2>&1 | tail -2
AUKIT_FLAC_SLOW_RESTORE=1 timeout 150 python -m pytest tests -m gpu -x -q -k "flac" 2>&1 | tail -1
rm -rf /tmp/fl
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fl -o fl -- python3 bench.py --workload flac_pipeline --steps 5 --warmup 1 --cpu-streams 0 --extra-windows 0 > /dev/null 2>&1
python3 tools/kstats.py /tmp/fl 12
timeout 120 python bench.py --workload flac_pipeline --steps 10 --warmup 1 --cpu-streams 0 2>&1 | tail -1 | cut -c1-200
