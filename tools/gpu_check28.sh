#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_configs.py tests/test_gpu_filters.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/t28_$i.json 2> gpurun_out/t28_$i.err; python - <<PY
import json
d=json.loads(open('gpurun_out/t28_$i.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['host_cpu_cores_busy'], d['config'].get('owf'), d['config'].get('decoder_frame_threads'), d['secondary']['value'])
PY
done
