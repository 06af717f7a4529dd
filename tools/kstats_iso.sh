#!/bin/bash
# isolated per-kernel statistics (nothing pipelined: owf 0, one decoder thread): tools/kstats_iso.sh <workload> <tag> [more bench args]
R=${GRAFT_REPO_ROOT:-$PWD}; wl=${1:-1080p}; tag=${2:-iso}
cd /tmp; export TMPDIR=/tmp
KVAZZUP_BENCH_NOPROF=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o p -- python3 $R/bench.py --workload $wl --no-cpu-baseline --no-secondary --no-host-boundary --repeats 1 --steps 2 --warmup 1 --owf 0 --decoder-frame-threads 1 "${@:3}" > $R/gpurun_out/prof_$tag.log 2>&1
f=$(find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/${tag}_kernel_stats.csv; rm -rf $R/gpurun_out/prof_$tag      # (the per-dispatch trace is large: gpurun brings back at most 64 MiB)
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/${tag}_kernel_stats.csv")))
for r in rows:
    n=r['Name'].split('(')[0].replace('kvzx::','').replace('void ','')
    if n.startswith('k_') : print("%-28s calls %5s avg %9.2f us  total %8.2f ms" % (n[:28], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
