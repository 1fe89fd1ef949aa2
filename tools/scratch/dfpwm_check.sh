python -m pytest tests -m gpu -x -q -k "dfpwm or fuzz or golden" 2>&1 | tail -2
rm -rf /tmp/df
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/df -o df -- python3 bench.py --workload dfpwm_transcode --steps 5 --warmup 1 --cpu-streams 0 --extra-windows 0 > /dev/null 2>&1
python3 tools/kstats.py /tmp/df 12
python bench.py --workload dfpwm_transcode --steps 10 --warmup 1 --cpu-streams 0 2>&1 | tail -1 | cut -c1-200
