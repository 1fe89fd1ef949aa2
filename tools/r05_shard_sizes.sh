mkdir -p gpurun_out/r05a
for s in 2048 4096 8192 16384; do python bench.py --workload dfpwm_transcode --streams $s --extra-windows 1 --cpu-streams 0 2>gpurun_out/r05a/df_$s.err | tail -1 >> gpurun_out/r05a/shard_lines.jsonl; done
for s in 256 512 1024 2048; do python bench.py --workload flac_pipeline --streams $s --extra-windows 1 --cpu-streams 0 2>gpurun_out/r05a/fl_$s.err | tail -1 >> gpurun_out/r05a/shard_lines.jsonl; done
cat gpurun_out/r05a/shard_lines.jsonl | cut -c1-400
