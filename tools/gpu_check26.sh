#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
(echo "== 1080p"; bash tools/gpu_overlap.sh 1080p 2>&1 | grep -v "^W2026\|output_stream\|simple_timer"; echo "== 4k"; bash tools/gpu_overlap.sh 4k 2>&1 | grep -v "^W2026\|output_stream\|simple_timer") > $R/gpurun_out/r02_stream_overlap.txt 2>&1
tail -40 $R/gpurun_out/r02_stream_overlap.txt
