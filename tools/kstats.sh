#!/bin/bash
# per-kernel statistics of one (pipelined, default) bench.py run on the GPU box: tools/kstats.sh <tag> [bench args...]
R=${GRAFT_REPO_ROOT:-$PWD}; tag=$1; shift
cd /tmp; export TMPDIR=/tmp
KVAZZUP_BENCH_NOPROF=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o p -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-host-boundary --repeats 1 "$@" > $R/gpurun_out/prof_$tag.log 2>&1
f=$(find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/${tag}_kernel_stats.csv; rm -rf $R/gpurun_out/prof_$tag      # (the per-dispatch trace is large: gpurun brings back at most 64 MiB)
cut -d, -f1-4 $R/gpurun_out/${tag}_kernel_stats.csv | cut -c1-110 | head -20
