#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_foreign.py tests/test_gpu_filters.py tests/test_gpu_golden.py -m gpu -x -q > gpurun_out/r02_dec.log 2>&1; echo "dec rc $?"; tail -3 gpurun_out/r02_dec.log | cut -c1-300
KVAZZUP_AMD_TRACE=1 timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r02_bench_q.json 2> gpurun_out/r02_bench_q.err; echo "bench rc $?"
grep "thread ms" gpurun_out/r02_bench_q.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r02_bench_q.json').read().strip().splitlines()[-1])
print(d['value'], d['kernels_us'], d['filter_busy_ms_per_picture'], d['config']['host_cpu_cores_busy'])
PY
