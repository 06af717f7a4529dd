#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_lost_pictures.py tests/test_gpu_filters.py tests/test_gpu_configs.py -q -m gpu -n 4 > gpurun_out/r06_lost_tests.txt 2>&1; grep -E "^FAILED|passed|failed|^E  " gpurun_out/r06_lost_tests.txt | head -30
