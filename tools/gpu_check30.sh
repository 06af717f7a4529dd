#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_foreign.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
KVAZZUP_BENCH_THREADS=1 timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/t30.json 2> gpurun_out/t30.err
grep '^thread' gpurun_out/t30.err | awk '{n[$3]++; s[$3]+=$4} END {for (k in n) printf "%-16s x%2d  %.3f s\n", k, n[k], s[k]}'
python -c "
import json; d=json.loads(open('gpurun_out/t30.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['host_cpu_cores_busy'], d['kernels_us'].get('host_cabac_parse'))"
for i in 1 2 3; do timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/t30_$i.json 2> gpurun_out/t30_$i.err; python - <<PY
import json
d=json.loads(open('gpurun_out/t30_$i.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['host_cpu_cores_busy'], d['secondary']['value'], d['secondary']['host_cpu_cores_busy'])
PY
done
