#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_foreign.py tests/test_gpu_configs.py tests/test_gpu_tilesplit.py -m gpu -x -q 2>&1 | tail -5
bash tools/kstats_iso.sh 4k t23_iso4k 2>&1 | grep "k_dec_intra\|k_intra_recon"
bash tools/kstats_iso.sh 1080p t23_iso1080p 2>&1 | grep "k_dec_intra\|k_intra_recon"
