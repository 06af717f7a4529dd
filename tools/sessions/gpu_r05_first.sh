#!/bin/bash
# round 5, first GPU call: the full GPU suite, then the driver's bench line with the new legs
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q -n 3 2>&1 | tail -6
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_first.json 2> gpurun_out/r05_bench_first.err; echo "bench rc $?"; tail -c 600 gpurun_out/r05_bench_first.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_bench_first.json'))
print('value',d['value'],'rates',d.get('rates'))
print('latency',json.dumps(d.get('latency_us'))[:3000])
print('bounds',json.dumps(d.get('bounds'))[:1500])
print('uvgcomm',json.dumps(d.get('uvgcomm_defaults'))[:600])
print('sec', {k:v for k,v in (d.get('secondary') or {}).items() if k in ('value','uvgcomm_defaults','latency_us')})
print('all_intra', (d.get('all_intra') or {}).get('value'), 'default_mode', (d.get('default_mode') or {}).get('value'))
print('cores', d['config']['host_cpu_cores_busy'])
PY
