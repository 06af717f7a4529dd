#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
ROUND=r05 bash tools/gpu_profiles.sh 2>&1 | tail -5
timeout 300 python tools/measure/owf0_timeline.py 1080p 60 > gpurun_out/r05_owf0_timeline.txt 2>&1; timeout 300 python tools/measure/owf0_timeline.py 4k 30 >> gpurun_out/r05_owf0_timeline.txt 2>&1; grep -c . gpurun_out/r05_owf0_timeline.txt
timeout 300 python tools/tok_timeline.py 1920 1080 > gpurun_out/r05_tokenizer_timeline.txt 2>&1; tail -3 gpurun_out/r05_tokenizer_timeline.txt
timeout 300 python tools/measure/b_decode_rate.py > gpurun_out/r05_b_decode_rate.txt 2>&1; tail -5 gpurun_out/r05_b_decode_rate.txt
