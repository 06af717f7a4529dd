#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_foreign.py tests/test_gpu_filters.py -m gpu -x -q > gpurun_out/r02_dec.log 2>&1; echo "dec rc $?"; tail -5 gpurun_out/r02_dec.log | cut -c1-600
KVAZZUP_AMD_DEC_INTRA_THREADS=64 timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_foreign.py -m gpu -x -q > gpurun_out/r02_dec64.log 2>&1; echo "dec64 rc $?"; tail -3 gpurun_out/r02_dec64.log | cut -c1-600
for t in 256 64; do
KVAZZUP_AMD_DEC_INTRA_THREADS=$t timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --secondary-steps 4 > gpurun_out/r02_bench_t$t.json 2> gpurun_out/r02_bench_t$t.err; echo "bench rc $?"
python - <<PY
import json
d=json.loads(open('gpurun_out/r02_bench_t$t.json').read().strip().splitlines()[-1])
print($t, d['value'], {k:v for k,v in d['kernels_us'].items() if 'dec' in k or 'parse' in k}, d['filter_busy_ms_per_picture'], d['config']['host_cpu_cores_busy'])
print(d['secondary']['value'], {k:v for k,v in d['secondary']['kernels_us'].items() if 'dec' in k or 'parse' in k})
PY
done
