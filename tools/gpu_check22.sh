#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_tilesplit.py -m gpu -x -q 2>&1 | tail -3
bash tools/kstats_iso.sh 4k t22_iso4k 2>&1 | grep "k_inter_signal\|k_tok_compact"
bash tools/kstats_iso.sh 1080p t22_iso1080p 2>&1 | grep "k_inter_signal\|k_tok_compact"
bash tools/kstats_iso.sh 4k t22_iso4k_gpuent --gpu-entropy 2>&1 | grep "k_tok_compact\|k_cabac"
