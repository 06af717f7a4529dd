#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r02_pytest_gpu.log | cut -c1-400
bash tools/kstats_iso.sh 1080p r02_iso1080p | grep "k_pad\|k_deblock\|k_dec_deblock\|k_me \|k_inter_recon\|k_dec_inter"
bash tools/kstats_iso.sh 4k r02_iso4k | grep "k_pad\|k_deblock\|k_dec_deblock\|k_me \|k_inter_recon\|k_dec_inter"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_q.json 2> gpurun_out/r02_bench_q.err; echo "bench rc $?"
python - <<PY
import json
d=json.loads(open('gpurun_out/r02_bench_q.json').read().strip().splitlines()[-1])
print(d['value'], d['kernels_us'], d['config']['host_cpu_cores_busy'])
print(d['secondary']['value'], d['secondary']['kernels_us'])
PY
