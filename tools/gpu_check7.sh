#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r02_pytest_gpu.log | cut -c1-400
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_driver.json 2> gpurun_out/r02_bench_driver.err; echo "bench rc $?"
python - <<PY
import json
d=json.loads(open('gpurun_out/r02_bench_driver.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['kernels_us'], d['filter_busy_ms_per_picture'], d['config']['host_cpu_cores_busy'], d['config']['psnr_y'])
print(d['secondary']['value'], d['secondary']['kernels_us'], d['secondary']['psnr_y'])
print(d['cpu_baseline'])
PY
