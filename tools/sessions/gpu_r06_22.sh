#!/bin/bash
# round 6: which priority level (= which pool of hardware queues) the DECODER's streams get, beside the encoder's "hnn" -- host-boundary and resident legs
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
hb() { KVAZZUP_BENCH_NOPROF=1 python bench.py --host-io --no-cpu-baseline --no-secondary --no-preset-line --streams-per-gpu 0 --repeats 3 --steps 10 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); c=l['config']; print('$1 host', l['value'], c['runs_fps'], 'cores', c['host_cpu_cores_busy'], l['filter_busy_ms_per_picture'])"; }
res() { KVAZZUP_BENCH_NOPROF=1 python bench.py --no-host-boundary --no-cpu-baseline --no-secondary --no-preset-line --streams-per-gpu 0 --repeats 3 --steps 10 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); c=l['config']; print('$1 resident', l['value'], c['runs_fps'])"; }
{
for p in hnnnnn hnnlll hnnlnn hnnnll hnnhnn hnnlhh hnnhll nnnnnn; do
KVAZZUP_AMD_PRIO=$p hb $p; KVAZZUP_AMD_PRIO=$p res $p
done
KVAZZUP_AMD_DL=own hb dl_own; KVAZZUP_AMD_DL=own res dl_own
} > gpurun_out/r06_prio_sweep.txt 2>&1; cat gpurun_out/r06_prio_sweep.txt
