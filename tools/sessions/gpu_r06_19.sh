#!/bin/bash
# round 6: the host-boundary leg with the decoder's waits polled instead of napped (KVAZZUP_AMD_SPIN), and with shorter naps
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
hb() { KVAZZUP_BENCH_NOPROF=1 python bench.py --host-io --no-cpu-baseline --no-secondary --no-preset-line --streams-per-gpu 0 --repeats 3 --steps 10 --warmup 2 $2 2>/dev/null | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); c=l['config']; print('$1', l['value'], c['runs_fps'], 'cores', c['host_cpu_cores_busy'], l['filter_busy_ms_per_picture'])"; }
{
for i in 1 2; do
hb nap ""
KVAZZUP_AMD_SPIN=1 hb spin ""
KVAZZUP_AMD_NAP_US=5 hb nap5 ""
done
} > gpurun_out/r06_spin_ab.txt 2>&1; cat gpurun_out/r06_spin_ab.txt
