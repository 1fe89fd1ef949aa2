#!/bin/bash
# round evidence: bench lines of every workload + rocprofv3 kernel stats / PMC passes per workload (GPU box; ~3 min)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
STEPS=10 bash tools/bench_all.sh
for spec in "wavef64:" "fastwave:--exact-math 0" "pcmstream:--workload pcm16_stream" "g711cubic:--workload g711_cubic" "stereo:--workload pcm16_stereo" "stereostream:--workload pcm16_stereo_stream" "g711stream:--workload g711_stream" "ima:--workload ima_stream" "imapipe:--workload ima_pipeline" "msadpcm:--workload msadpcm_stream" "qoa:--workload qoa_stream" "dfpwm:--workload dfpwm_transcode" "dfpwm2048:--workload dfpwm_transcode --streams 2048" "flac:--workload flac_pipeline" "flac256:--workload flac_pipeline --streams 256"; do
  tag=${spec%%:*}; extra=${spec#*:}
  pat=""; [ $tag = dfpwm ] && pat=blockmaps; [ $tag = dfpwm2048 ] && pat=blockmaps; [ $tag = flac256 ] && pat=k_flac_find; [ $tag = flac ] && pat=k_flac_find; [ $tag = qoa ] && pat=k_qoa_walk
  TIMELINE_PAT="$pat" BENCH_EXTRA="$extra" PROF_TAG="r1_$tag" bash tools/profile_bench.sh > /dev/null 2>&1
done
rm -f gpurun_out/prof/*.log
ls gpurun_out/prof | wc -l
