#!/bin/bash
# round 6, third GPU session: the crawl's knobs (compact streams, GPU_MAX_HW_QUEUES) alone and beside busy neighbours; what intra units in P pictures + the search
# on the input picture would make of the ultrafast headline (1080p, 4K): rate, bits, PSNR, cores
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
hl() { python bench.py --workload $2 --no-cpu-baseline --no-secondary --no-host-boundary --no-preset-line --streams-per-gpu 0 --repeats 3 --steps 12 --warmup 2 $3 2>/dev/null | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); c=l['config']; print('%-44s %8.1f frames/s %s bits/picture %.1f psnr %.3f cores %.1f  k_us %s' % ('$1', l['value'], c['runs_fps'], c['bits_per_picture'], c['psnr_y'], c['host_cpu_cores_busy'], {k:v for k,v in l['kernels_us'].items() if k in ('k_me','k_inter_recon','k_intra_recon<P>','k_intra_analyse<P>','host_cabac_parse','host_arith_coder')}))"; }
{
hl "1080p ultrafast as it is" 1080p ""
hl "1080p + intra-in-p=1" 1080p "--custom intra-in-p=1"
hl "1080p + me-source=1" 1080p "--custom me-source=1"
hl "1080p + intra-in-p=1 + me-source=1" 1080p "--custom intra-in-p=1 --custom me-source=1"
hl "1080p ultrafast as it is" 1080p ""
hl "1080p + intra-in-p=1 + me-source=1" 1080p "--custom intra-in-p=1 --custom me-source=1"
hl "4k ultrafast as it is" 4k ""
hl "4k + intra-in-p=1 + me-source=1" 4k "--custom intra-in-p=1 --custom me-source=1"
hl "4k + me-source=1" 4k "--custom me-source=1"
} > gpurun_out/r06_ultrafast_tools_ab.txt 2>&1; cat gpurun_out/r06_ultrafast_tools_ab.txt
cd $R; bash tools/measure/crawl_root_cause.sh > gpurun_out/r06_crawl_root_cause.txt 2>&1; cat gpurun_out/r06_crawl_root_cause.txt
