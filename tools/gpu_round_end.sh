#!/bin/bash
# the way the driver ends a round: the full GPU suite (serial, as the driver runs it), then the driver's bench line -- kept under profiles/
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out; r=${ROUND:-r05}
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
cd $R; timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${r}_bench_driver_line.json 2> gpurun_out/${r}_bench_driver_line.err; echo "bench rc $?"; tail -c 400 gpurun_out/${r}_bench_driver_line.err
python - <<PY
import json
d=json.load(open('gpurun_out/${r}_bench_driver_line.json'))
print('value',d['value'],'rates',{k:v for k,v in d['rates'].items() if k!='note'})
l=d.get('latency_us') or {}
for k in ('uvgcomm_default_owf0','uvgcomm_owf2','throughput_setting'):
    print(k, (l.get(k) or {}).get('encoding_delay_us'), (l.get(k) or {}).get('total_delay_us'))
s=d.get('secondary') or {}
print('4k', s.get('value'), (s.get('host_boundary') or {}).get('value'), (s.get('uvgcomm_defaults') or {}).get('value'), ((s.get('latency_us') or {}).get('uvgcomm_default_owf0') or {}).get('encoding_delay_us'), ((s.get('latency_us') or {}).get('uvgcomm_default_owf0') or {}).get('total_delay_us'))
print('all_intra', (d.get('all_intra') or {}).get('value'), 'default_mode', (d.get('default_mode') or {}).get('value'), 'bounds', {k:(v.get('value'), v.get('bits_per_picture')) for k,v in (d.get('bounds') or {}).items()})
print('streams', [(m.get('streams'), m.get('value')) for m in d.get('streams_per_gpu') or []])
print('roofline', {k:d['roofline'][k] for k in ('kernel','achieved','frac','avg_launch_us','traffic')}, 'cores', d['config']['host_cpu_cores_busy'], 'cpu', d.get('cpu_baseline',{}).get('value'))
print(d['kernels_us'])
PY
