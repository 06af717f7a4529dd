#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_encoder.py tests/test_gpu_foreign.py -m gpu -x -q 2>&1 | tail -3
bash tools/kstats_iso.sh 1080p t25_iso1080p 2>&1 | grep "k_dec_intra\|k_intra_recon\|k_tokenize"
bash tools/kstats_iso.sh 4k t25_iso4k 2>&1 | grep "k_dec_intra\|k_intra_recon\|k_tokenize"
PMC_PASSES="fetch:FETCH_SIZE write:WRITE_SIZE" bash tools/pmc_traffic.sh 1080p > /dev/null 2>&1; python3 -c "
import json; d=json.load(open('$R/gpurun_out/pmc_traffic_1080p.json')); k=d.get('kernels',d)
for n in ('k_intra_recon','k_dec_intra','k_tokenize','k_tok_compact'): print(n, k[n].get('traffic_bytes'))"
