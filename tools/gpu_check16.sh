#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_filters.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
run() { tag=$1; wl=$2; D=$3; shift; shift; shift; timeout 600 env "$@" python bench.py --workload $wl --decoder-frame-threads $D --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r02_ft_$tag.json 2> gpurun_out/r02_ft_$tag.err; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r02_ft_$tag.json').read().strip().splitlines()[-1])
    print('$tag', d['value'], d['config']['host_cpu_cores_busy'], d['config']['decoder_frame_threads'], d['kernels_us'].get('host_cabac_parse'), d['filter_busy_ms_per_picture'])
except Exception as e: print('$tag failed', e); print(open('gpurun_out/r02_ft_$tag.err').read()[-800:])
PY
}
run 4k_d12 4k 12 X=1
run 4k_d16 4k 16 X=1
run 4k_d24 4k 24 X=1
run 4k_d32 4k 32 X=1
run 4k_def 4k 0 X=1
run 1080_def 1080p 0 X=1
run 1080_d16 1080p 16 X=1
run 1080_d24 1080p 24 X=1
