#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_filters.py tests/test_gpu_configs.py tests/test_gpu_golden.py -m gpu -x -q -n 3 --deselect tests/test_gpu_configs.py::test_config3_bench_command_with_two_ranks 2>&1 | tail -4
bash tools/kstats_iso.sh 1080p r05f_iso1080p --streams-per-gpu 0 --no-preset-line 2>&1 | grep -E "k_tokenize|k_tok_compact|k_inter_signal"
bash tools/kstats_iso.sh 4k r05f_iso4k --streams-per-gpu 0 --no-preset-line 2>&1 | grep -E "k_tokenize|k_tok_compact"
timeout 300 python tools/tok_timeline.py 1920 1080 2>&1 | tail -9
