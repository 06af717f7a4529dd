#!/bin/bash
# round 5, second GPU call: the full GPU suite on the new kernels (k_inter_signal per 8x8, chain_claim, recon sink), the OWF-0 timelines, a short bench
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q -n 3 2>&1 | tail -6
timeout 300 python tools/measure/owf0_timeline.py 1080p 60 > gpurun_out/r05_owf0_timeline.txt 2>&1; timeout 300 python tools/measure/owf0_timeline.py 4k 30 >> gpurun_out/r05_owf0_timeline.txt 2>&1; cat gpurun_out/r05_owf0_timeline.txt
bash tools/kstats_iso.sh 1080p r05a_iso1080p --streams-per-gpu 0 --no-preset-line 2>&1 | tail -20
timeout 900 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --streams-per-gpu 0 > gpurun_out/r05_bench_second.json 2> gpurun_out/r05_bench_second.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_bench_second.json'))
print('rates',d.get('rates'))
l=d.get('latency_us') or {}
for k in ('uvgcomm_default_owf0','uvgcomm_owf2'):
    print(k, (l.get(k) or {}).get('encoding_delay_us'), (l.get(k) or {}).get('total_delay_us'))
s=d.get('secondary') or {}
print('4k', s.get('value'), (s.get('uvgcomm_defaults') or {}).get('value'), ((s.get('latency_us') or {}).get('uvgcomm_default_owf0') or {}).get('encoding_delay_us'))
print('all_intra', (d.get('all_intra') or {}).get('value'), 'default_mode', (d.get('default_mode') or {}).get('value'), 'flat', ((d.get('bounds') or {}).get('flat') or {}).get('value'))
print(d['kernels_us'])
PY
