#!/bin/bash
# round 6, eighth GPU session: the search ahead of the chain capped to N workgroups per compute unit (KVAZZUP_AMD_ME_PER_CU): default-mode rate and the chain's kernel times
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
dm() { python bench.py --no-cpu-baseline --no-secondary --no-host-boundary --streams-per-gpu 0 --repeats 3 --steps 8 --warmup 1 --custom preset=veryfast --custom bitrate=1000000 --custom rc-algorithm=lambda $2 2>/dev/null | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('$1', l['value'], l['config']['runs_fps'], {k:v for k,v in l['kernels_us'].items() if k in ('k_me','k_intra_analyse<P>','k_inter_recon','k_subpel','k_sao','k_intra_recon<P>','k_deblock')})"; }
{
for rep in 1 2; do
dm cap0 ""
for c in 3 4 5 6; do KVAZZUP_AMD_ME_PER_CU=$c dm cap$c ""; done
done
} > gpurun_out/r06_me_cap_ab.txt 2>&1; cat gpurun_out/r06_me_cap_ab.txt
