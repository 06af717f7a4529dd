#!/bin/bash
# per-kernel statistics of one bench.py run (GPU box): tools/kstats.sh <out-dir> [bench args...]
R=${GRAFT_REPO_ROOT:-$PWD}; out=$1; shift
cd /tmp; export TMPDIR=/tmp
KVAZZUP_BENCH_NOPROF=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$out -o p -- python3 $R/bench.py "$@" > /dev/null 2>&1
find $R/gpurun_out/$out -name "*kernel_stats.csv" | head -1 | xargs cut -d, -f1-4 | cut -c1-110
