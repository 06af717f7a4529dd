#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
KVAZZUP_BENCH_THREADS=1 timeout 600 python bench.py --workload 4k --steps 12 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/t31.json 2> gpurun_out/t31.err
grep '^thread' gpurun_out/t31.err | awk '{n[$3]++; s[$3]+=$4} END {for (k in n) printf "%-16s x%2d  %.3f s\n", k, n[k], s[k]}'
python -c "
import json; d=json.loads(open('gpurun_out/t31.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['host_cpu_cores_busy'], d['kernels_us'].get('host_cabac_parse'), d['ms_per_step']*d['steps'])"
