#!/bin/bash
# the round's profile set: pipelined kernel stats (the bench command itself), isolated kernel stats, PMC traffic passes
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
bash tools/kstats.sh r03_bench1080p --steps 8 --warmup 2
bash tools/kstats.sh r03_bench4k --workload 4k --steps 4 --warmup 1
bash tools/kstats_iso.sh 1080p r03_iso1080p
bash tools/kstats_iso.sh 4k r03_iso4k
bash tools/pmc_traffic.sh 1080p
bash tools/pmc_traffic.sh 4k
# the host-boundary leg under the kernel + memory-copy trace (copy engine beside the kernels), and how the streams share the GPU
bash tools/gpu_hosttrace.sh 1080p > gpurun_out/r03_hosttrace_1080p.txt 2>&1
bash tools/gpu_overlap.sh 1080p > gpurun_out/r03_stream_overlap.txt 2>&1; bash tools/gpu_overlap.sh 4k >> gpurun_out/r03_stream_overlap.txt 2>&1
