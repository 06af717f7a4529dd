#!/bin/bash
# round 6, seventh GPU session: k_inter_recon's rate-control / fractional forms at 7 waves per SIMD (72 registers, one spilled) against 5-6 (78): parity of the
# forms concerned, default-mode rate and kernel times
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
dm() { python bench.py --no-cpu-baseline --no-secondary --no-host-boundary --streams-per-gpu 0 --repeats 3 --steps 8 --warmup 1 --custom preset=veryfast --custom bitrate=1000000 --custom rc-algorithm=lambda $2 2>/dev/null | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('$1', l['value'], l['config']['runs_fps'], {k:v for k,v in l['kernels_us'].items() if k in ('k_inter_recon','k_subpel','k_sao','k_intra_recon<P>','k_deblock')})"; }
{
dm waves5 ""; 
KVAZZUP_AMD_LIBRARY=$R/kvazzup_amd/libkvazzup_amd_rw7.so dm waves7 ""
dm waves5 "";
KVAZZUP_AMD_LIBRARY=$R/kvazzup_amd/libkvazzup_amd_rw7.so dm waves7 ""
} > gpurun_out/r06_recon_waves_ab.txt 2>&1; cat gpurun_out/r06_recon_waves_ab.txt
KVAZZUP_AMD_LIBRARY=$R/kvazzup_amd/libkvazzup_amd_rw7.so timeout 900 python -m pytest tests/test_gpu_me_source.py tests/test_gpu_encoder.py -q -n 3 -k "subme or search or rate or intra_units or default" 2>&1 | tail -3
