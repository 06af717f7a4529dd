#!/bin/bash
# the round's profile set: pipelined kernel stats (the bench command itself), isolated kernel stats, PMC traffic passes
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
bash tools/kstats.sh r02_bench1080p --steps 8 --warmup 2
bash tools/kstats.sh r02_bench4k --workload 4k --steps 4 --warmup 1
bash tools/kstats_iso.sh 1080p r02_iso1080p
bash tools/kstats_iso.sh 4k r02_iso4k
bash tools/pmc_traffic.sh 1080p
bash tools/pmc_traffic.sh 4k
