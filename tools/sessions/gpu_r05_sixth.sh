#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 600 bash tools/measure/queue_share/run.sh > gpurun_out/r05_queue_share.txt 2>&1; cat gpurun_out/r05_queue_share.txt
# GPU arithmetic-decoder probe: gpu-entropy coder + the decoder's mirror of it, isolated kernel statistics
KVAZZUP_AMD_PARSE_PROBE=1 bash tools/kstats_iso.sh 1080p r05_probe1080p --streams-per-gpu 0 --no-preset-line --gpu-entropy 2>&1 | grep -E "k_cabac|k_tokenize"; grep "parse probe" gpurun_out/prof_r05_probe1080p.log | tail -2
KVAZZUP_AMD_PARSE_PROBE=1 bash tools/kstats_iso.sh 4k r05_probe4k --streams-per-gpu 0 --no-preset-line --gpu-entropy 2>&1 | grep -E "k_cabac|k_tokenize"; grep "parse probe" gpurun_out/prof_r05_probe4k.log | tail -2
# pipelined: many pictures' probes side by side (owf 6)
KVAZZUP_AMD_PARSE_PROBE=1 bash tools/kstats.sh r05_probe_pipelined --steps 6 --warmup 2 --streams-per-gpu 0 --no-preset-line --gpu-entropy 2>&1 | grep -E "cabac" ; tail -c 300 gpurun_out/prof_r05_probe_pipelined.log
timeout 2400 python -m pytest tests -m gpu -x -q -n 3 2>&1 | tail -4
bash tools/kstats_iso.sh 1080p r05e_iso1080p --streams-per-gpu 0 --no-preset-line 2>&1 | grep -E "k_inter_signal|k_tokenize"
