#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r02_pytest_gpu.log | cut -c1-600
for t in 256; do
KVAZZUP_AMD_DEC_INTRA_THREADS=$t timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --secondary-steps 4 > gpurun_out/r02_bench_t$t.json 2> gpurun_out/r02_bench_t$t.err; echo "bench rc $?"
python - <<PY
import json
d=json.loads(open('gpurun_out/r02_bench_t$t.json').read().strip().splitlines()[-1])
print($t, d['value'], d['kernels_us'], d['filter_busy_ms_per_picture'], d['config']['host_cpu_cores_busy'], d['config']['psnr_y'], d['config']['bits_per_picture'])
print(d['secondary']['value'], d['secondary']['kernels_us'])
PY
done
