#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_tilesplit.py tests/test_gpu_filters.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
KVAZZUP_BENCH_THREADS=1 timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/t29.json 2> gpurun_out/t29.err
grep '^thread' gpurun_out/t29.err | awk '{n[$3]++; s[$3]+=$4} END {for (k in n) printf "%-16s x%2d  %.3f s\n", k, n[k], s[k]}'
for i in 1 2 3; do timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/t29_$i.json 2> gpurun_out/t29_$i.err; python - <<PY
import json
d=json.loads(open('gpurun_out/t29_$i.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['host_cpu_cores_busy'], d['secondary']['value'], d['secondary']['host_cpu_cores_busy'])
PY
done
