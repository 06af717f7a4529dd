#!/bin/bash
# round 6: the host-boundary leg with more hardware queues for the process (HIP maps streams onto GPU_MAX_HW_QUEUES queues, 4 by default: streams that share one run in order)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
hb() { KVAZZUP_BENCH_NOPROF=1 python bench.py --host-io --no-cpu-baseline --no-secondary --no-preset-line --streams-per-gpu 0 --repeats 3 --steps 10 --warmup 2 $2 2>/dev/null | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); c=l['config']; print('$1', l['value'], c['runs_fps'], 'cores', c['host_cpu_cores_busy'], l['filter_busy_ms_per_picture'])"; }
res() { KVAZZUP_BENCH_NOPROF=1 python bench.py --no-host-boundary --no-cpu-baseline --no-secondary --no-preset-line --streams-per-gpu 0 --repeats 3 --steps 10 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); c=l['config']; print('$1 resident', l['value'], c['runs_fps'])"; }
{
for q in 4 8 16 2; do
GPU_MAX_HW_QUEUES=$q hb hwq$q ""
done
hb default ""
for q in 8 16; do GPU_MAX_HW_QUEUES=$q res hwq$q; done
res default
} > gpurun_out/r06_hwq_ab.txt 2>&1; cat gpurun_out/r06_hwq_ab.txt
