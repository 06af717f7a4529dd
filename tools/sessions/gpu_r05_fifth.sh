#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null | cut -c1-200; lscpu | grep -E "Model name|Thread|Core|Socket|L3" 
timeout 300 python tools/measure/owf0_timeline.py 1080p 60 2>&1 | grep -v "^stages" | tail -12
for t in 4 8 32; do echo "PARSE_THREADS $t"; KVAZZUP_AMD_PARSE_THREADS=$t timeout 300 python tools/measure/owf0_timeline.py 1080p 60 2>&1 | grep -E "dec0 +-> dlaunch0|threads took|^  [0-9]+:"; done
timeout 1200 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_filters.py tests/test_gpu_configs.py -m gpu -x -q -n 3 2>&1 | tail -3
bash tools/kstats_iso.sh 1080p r05d_iso1080p --streams-per-gpu 0 --no-preset-line 2>&1 | grep -E "k_inter_signal|k_tokenize|k_tok_compact"
