#!/bin/bash
# GPU arithmetic coder: parity subset, then A/B bench lines (host pool vs k_cabac_rows at several owf)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_filters.py tests/test_gpu_golden.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r02_pytest_cabac.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r02_pytest_cabac.log | cut -c1-600
run() { tag=$1; owf=$2; shift; shift; timeout 600 env "$@" python bench.py --gpus 1 --steps 12 --warmup 3 --owf $owf --no-cpu-baseline --no-secondary > gpurun_out/r02_cabac_$tag.json 2> gpurun_out/r02_cabac_$tag.err; echo "$tag rc $?"; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r02_cabac_$tag.json').read().strip().splitlines()[-1])
    print('$tag', d['value'], d['config']['host_cpu_cores_busy'], {k:v for k,v in d['kernels_us'].items() if k in ('k_cabac_rows','host_arith_coder','k_tokenize','k_tok_compact','host_cabac_parse')}, d['filter_busy_ms_per_picture'])
except Exception as e: print('$tag failed', e); print(open('gpurun_out/r02_cabac_$tag.err').read()[-1500:])
PY
}
run host 3 KVAZZUP_AMD_ENTROPY=host
run gpu3 3 KVAZZUP_AMD_ENTROPY=gpu
run gpu6 6 KVAZZUP_AMD_ENTROPY=gpu
run gpu8 8 KVAZZUP_AMD_ENTROPY=gpu
