#!/bin/bash
# the round's profile set (run on the GPU box; summaries land in gpurun_out/, the ones to keep are copied to profiles/ by hand):
# pipelined kernel stats (the bench command's headline leg), isolated kernel stats, PMC traffic passes, the host-boundary trace, stream overlap,
# the default-mode chain (kernel durations and the gaps in front of them per hardware queue), the decoders' batched launches
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out; r=${ROUND:-r05}
bash tools/kstats.sh ${r}_bench1080p --steps 8 --warmup 2 --streams-per-gpu 0 --no-preset-line
bash tools/kstats.sh ${r}_bench4k --workload 4k --steps 4 --warmup 1 --streams-per-gpu 0
bash tools/kstats_iso.sh 1080p ${r}_iso1080p --streams-per-gpu 0 --no-preset-line
bash tools/kstats_iso.sh 4k ${r}_iso4k --streams-per-gpu 0
bash tools/pmc_traffic.sh 1080p --streams-per-gpu 0 --no-preset-line
bash tools/pmc_traffic.sh 4k --streams-per-gpu 0
bash tools/gpu_hosttrace.sh 1080p > gpurun_out/${r}_hosttrace_1080p.txt 2>&1
bash tools/gpu_overlap.sh 1080p > gpurun_out/${r}_stream_overlap.txt 2>&1; bash tools/gpu_overlap.sh 4k >> gpurun_out/${r}_stream_overlap.txt 2>&1
bash tools/gpu_overlap.sh 1080p 2 >> gpurun_out/${r}_stream_overlap.txt 2>&1; bash tools/gpu_overlap.sh 1080p 4 >> gpurun_out/${r}_stream_overlap.txt 2>&1
bash tools/measure/chain_gaps.sh default > gpurun_out/${r}_chain_gaps_default_mode.txt 2>&1
python tools/measure/batch_times.py 1080p > gpurun_out/${r}_batch_times_1080p.json 2> gpurun_out/batch_times.err
python tools/measure/batch_times.py 4k 12 > gpurun_out/${r}_batch_times_4k.json 2>> gpurun_out/batch_times.err
