#!/bin/bash
# round 5, fourth GPU call: tokenizer with sub-blocks in registers, parser rows without false sharing -- encoder / decoder suites, OWF-0 timelines, iso stats
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q -n 3 2>&1 | tail -4
timeout 300 python tools/measure/owf0_timeline.py 1080p 60 > gpurun_out/r05_owf0_timeline.txt 2>&1; timeout 300 python tools/measure/owf0_timeline.py 4k 30 >> gpurun_out/r05_owf0_timeline.txt 2>&1; grep -v "^stages" gpurun_out/r05_owf0_timeline.txt
bash tools/kstats_iso.sh 1080p r05c_iso1080p --streams-per-gpu 0 --no-preset-line 2>&1 | tail -16
bash tools/kstats_iso.sh 4k r05c_iso4k --streams-per-gpu 0 --no-preset-line 2>&1 | tail -16
