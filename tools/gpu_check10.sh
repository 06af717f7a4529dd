#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
run() { tag=$1; shift; timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/r02_subme_$tag.json 2> gpurun_out/r02_subme_$tag.err; echo "$tag rc $?"; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r02_subme_$tag.json').read().strip().splitlines()[-1])
    print('$tag', d['value'], d['config']['psnr_y'], d['config']['bits_per_picture'], d['config']['host_cpu_cores_busy'], {k:v for k,v in d['kernels_us'].items() if k in ('k_me','k_subpel','k_inter_recon','k_dec_inter')})
    if 'secondary' in d: print('   4K', d['secondary']['value'], d['secondary']['psnr_y'], d['secondary']['bits_per_picture'], {k:v for k,v in d['secondary']['kernels_us'].items() if k in ('k_me','k_subpel','k_inter_recon','k_dec_inter')})
except Exception as e: print('$tag failed', e); print(open('gpurun_out/r02_subme_$tag.err').read()[-1500:])
PY
}
run s0
run s2 --subme 2
run s4 --subme 4
run s4full --subme 4 --full-search --no-secondary
run s0full --full-search --no-secondary
