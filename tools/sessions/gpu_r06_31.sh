#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_everything.py tests/test_gpu_foreign.py tests/test_gpu_decoder.py -q -m gpu -n 4 -k "corrupt or survive or fuzz or lost or damaged" > gpurun_out/r06_corrupt.txt 2>&1; grep -E "^FAILED|passed|failed|^E  " gpurun_out/r06_corrupt.txt | head -60
