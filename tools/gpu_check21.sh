#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_decoder.py tests/test_gpu_foreign.py -m gpu -x -q 2>&1 | tail -3
bash tools/kstats_iso.sh 4k t21_iso4k 2>&1 | grep "k_"
bash tools/kstats_iso.sh 1080p t21_iso1080p 2>&1 | grep "k_"
cd $R
for wl in 4k 1080p; do timeout 600 python bench.py --workload $wl --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/t21_$wl.json 2> gpurun_out/t21_$wl.err; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/t21_$wl.json').read().strip().splitlines()[-1]); print('$wl', d['value'], d['config']['host_cpu_cores_busy'])
except Exception as e: print('$wl failed', e); print(open('gpurun_out/t21_$wl.err').read()[-800:])
PY
done
