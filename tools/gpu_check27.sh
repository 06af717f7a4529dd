#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
run() { tag=$1; shift; timeout 600 env "$@" python bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary $EXTRA > gpurun_out/t27_$tag.json 2> gpurun_out/t27_$tag.err; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/t27_$tag.json').read().strip().splitlines()[-1]); print('$tag', d['value'], d['config']['host_cpu_cores_busy'], d['config'].get('owf'))
except Exception as e: print('$tag failed', e); print(open('gpurun_out/t27_$tag.err').read()[-600:])
PY
}
for rep in a b c; do
EXTRA="--owf 3 --decoder-frame-threads 24" run owf3_d24_$rep X=1
EXTRA="--owf 4 --decoder-frame-threads 24" run owf4_d24_$rep KVAZZUP_AMD_MAX_DEPTH=8
EXTRA="--owf 5 --decoder-frame-threads 24" run owf5_d24_$rep KVAZZUP_AMD_MAX_DEPTH=8
EXTRA="--owf 6 --decoder-frame-threads 24" run owf6_d24_$rep KVAZZUP_AMD_MAX_DEPTH=8
EXTRA="--owf 5 --decoder-frame-threads 32" run owf5_d32_$rep KVAZZUP_AMD_MAX_DEPTH=8
EXTRA="--owf 3 --decoder-frame-threads 32" run owf3_d32_$rep X=1
done
for rep in a b; do
EXTRA="--owf 5 --workload 4k --decoder-frame-threads 24" run 4k_owf5_d24_$rep KVAZZUP_AMD_MAX_DEPTH=8
EXTRA="--owf 3 --workload 4k --decoder-frame-threads 24" run 4k_owf3_d24_$rep X=1
EXTRA="--owf 5 --workload 4k --decoder-frame-threads 32" run 4k_owf5_d32_$rep KVAZZUP_AMD_MAX_DEPTH=8
EXTRA="--owf 3 --workload 4k --decoder-frame-threads 32" run 4k_owf3_d32_$rep X=1
done
exit 0
