#!/bin/bash
# round 6: hardware queue counts between the default (4) and 8 -- host-boundary and resident legs
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
hb() { KVAZZUP_BENCH_NOPROF=1 python bench.py --host-io --no-cpu-baseline --no-secondary --no-preset-line --streams-per-gpu 0 --repeats 3 --steps 10 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); c=l['config']; print('$1 host', l['value'], c['runs_fps'], 'cores', c['host_cpu_cores_busy'], l['filter_busy_ms_per_picture'])"; }
res() { KVAZZUP_BENCH_NOPROF=1 python bench.py --no-host-boundary --no-cpu-baseline --no-secondary --no-preset-line --streams-per-gpu 0 --repeats 3 --steps 10 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); c=l['config']; print('$1 resident', l['value'], c['runs_fps'])"; }
{
for q in 3 5 6 7; do GPU_MAX_HW_QUEUES=$q hb hwq$q; GPU_MAX_HW_QUEUES=$q res hwq$q; done
hb default; res default
} > gpurun_out/r06_hwq_ab2.txt 2>&1; cat gpurun_out/r06_hwq_ab2.txt
