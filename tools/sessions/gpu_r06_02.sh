#!/bin/bash
# round 6, second GPU session: the chain's head folded into k_subpel (me-source tests again, default-mode A/B and chain gaps), the 8-party parity test,
# the 4K outliers with their neighbours, the multi-process crawl with synthetic neighbours
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_me_source.py "tests/test_gpu_configs.py::test_config3_eight_party_call_on_one_gpu" "tests/test_gpu_configs.py::test_default_mode_pipelined_with_intra_pictures_on_the_side_stream" tests/test_gpu_filters.py -x -q 2>&1 | tail -5
cd $R; bash tools/measure/chain_gaps.sh default > gpurun_out/r06_chain_gaps_default_mode.txt 2>&1; grep -E "^queue|k_me |k_intra_analyse<true>|k_subpel|k_inter_recon|k_picture|window" gpurun_out/r06_chain_gaps_default_mode.txt | head -20
cd $R
dm() { python bench.py --no-cpu-baseline --no-secondary --no-host-boundary --streams-per-gpu 0 --repeats 3 --steps 8 --warmup 1 --custom preset=veryfast --custom bitrate=1000000 --custom rc-algorithm=lambda $2 2>/dev/null | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('$1', l['value'], l['config']['runs_fps'], l['config']['bits_per_picture'], l['config']['psnr_y'], l['config']['host_cpu_cores_busy'])"; }
{ dm source-search ""; dm recon-search "--custom me-source=0"; dm source-search ""; dm recon-search "--custom me-source=0"; dm source-search-owf12 "--owf 12"; } > gpurun_out/r06_me_source_ab.txt 2>&1; cat gpurun_out/r06_me_source_ab.txt
cd $R; bash tools/measure/outliers.sh 4k > gpurun_out/r06_outliers_4k.txt 2>&1; head -60 gpurun_out/r06_outliers_4k.txt
cd $R; bash tools/measure/crawl_root_cause.sh > gpurun_out/r06_crawl_root_cause.txt 2>&1; cat gpurun_out/r06_crawl_root_cause.txt
