#!/bin/bash
# round 6, first GPU session: the VCN query with the right render-node range, the new me-source tests + the whole GPU suite, default-mode chain with the search
# on the input stream (chain gaps, A/B against me-source=0), a first bench line
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
./tools/probe/probe_vcn > gpurun_out/r06_probe_vcn.txt 2>&1; cat gpurun_out/r06_probe_vcn.txt
timeout 900 python -m pytest tests/test_gpu_me_source.py -x -q 2>&1 | tail -5
timeout 1500 python -m pytest tests -m gpu -q -n 3 --deselect tests/test_gpu_me_source.py 2>&1 | tail -5
cd $R; bash tools/measure/chain_gaps.sh default > gpurun_out/r06_chain_gaps_default_mode.txt 2>&1; grep -E "queue|k_me |k_intra_analyse<true>|k_subpel|k_inter_recon|window" gpurun_out/r06_chain_gaps_default_mode.txt | head -20
cd $R
dm() { python bench.py --no-cpu-baseline --no-secondary --no-host-boundary --streams-per-gpu 0 --repeats 3 --steps 8 --warmup 1 --custom preset=veryfast --custom bitrate=1000000 --custom rc-algorithm=lambda $2 2>/dev/null | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('$1', l['value'], l['config']['runs_fps'], l['config']['bits_per_picture'], l['config']['psnr_y'], l['config']['host_cpu_cores_busy'])"; }
{ dm source-search ""; dm recon-search "--custom me-source=0"; dm source-search ""; dm recon-search "--custom me-source=0"; } > gpurun_out/r06_me_source_ab.txt 2>&1; cat gpurun_out/r06_me_source_ab.txt
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_first.json 2> gpurun_out/r06_bench_first.err; echo "bench rc $?"; tail -c 300 gpurun_out/r06_bench_first.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_bench_first.json'))
print('value',d['value'],'rates',{k:v for k,v in d['rates'].items() if k!='note'})
s=d.get('secondary') or {}
print('4k', s.get('value'), s.get('runs_fps'))
dm=d.get('default_mode') or {}
print('default_mode', dm.get('value'), dm.get('bits_per_picture'), dm.get('psnr_y'), dm.get('search_on_reconstruction'))
print('all_intra', (d.get('all_intra') or {}).get('value'))
print('streams', [(m.get('streams'), m.get('value')) for m in d.get('streams_per_gpu') or []])
r=d['roofline']; print('roofline', {k:r.get(k) for k in ('kernel','achieved','frac','frac_traffic','avg_launch_us','traffic','algorithmic_bytes_per_launch')}); print(r.get('traffic_over_algorithmic'))
print(d['kernels_us'])
PY
