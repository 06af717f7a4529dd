#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
run() { tag=$1; shift; timeout 600 env "$@" python bench.py --workload 4k --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r02_4k_$tag.json 2> gpurun_out/r02_4k_$tag.err; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r02_4k_$tag.json').read().strip().splitlines()[-1])
    print('$tag', d['value'], d['config']['host_cpu_cores_busy'], d['kernels_us'].get('host_cabac_parse'), d['filter_busy_ms_per_picture'])
except Exception as e: print('$tag failed', e); print(open('gpurun_out/r02_4k_$tag.err').read()[-800:])
PY
}
run base X=1
run rp64 KVAZZUP_AMD_ROWPARSE_KB=64
run rp32 KVAZZUP_AMD_ROWPARSE_KB=32
run rp64b KVAZZUP_AMD_ROWPARSE_KB=64 KVAZZUP_AMD_PARSE_THREADS=8
