#!/bin/bash
# round 6: where the host-boundary leg's filter threads spend their time (the library's own timeline), and the same leg with more copy helpers
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
hb() { KVAZZUP_BENCH_NOPROF=1 $3 python bench.py --host-io --no-cpu-baseline --no-secondary --no-preset-line --streams-per-gpu 0 --repeats 3 --steps 10 --warmup 2 $2 2>/dev/null | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); c=l['config']; print('$1', l['value'], c['runs_fps'], 'cores', c['host_cpu_cores_busy'], l['filter_busy_ms_per_picture'])"; }
{
hb default "" ""
KVAZZUP_BENCH_COPY_THREADS=8 hb copy8 "" ""
KVAZZUP_BENCH_COPY_THREADS=2 hb copy2 "" ""
hb owf8 "--owf 8" ""
hb frame48 "--decoder-frame-threads 48" ""
KVAZZUP_AMD_TIMELINE=/tmp/tl.txt hb timeline "--repeats 1 --steps 4" ""
python tools/host_timeline_dec.py /tmp/tl.txt | head -24
} > gpurun_out/r06_host_boundary_threads.txt 2>&1; cat gpurun_out/r06_host_boundary_threads.txt
