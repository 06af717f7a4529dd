#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q 2>&1 | tail -3
python tools/tok_timeline.py 3840 2160; python tools/tok_timeline.py 1920 1080
bash tools/kstats_iso.sh 4k t18_iso4k 2>&1 | grep "k_tok"
bash tools/kstats_iso.sh 1080p t18_iso1080p 2>&1 | grep "k_tok"
