#!/bin/bash
# round 6: concealment with the per-buffer mark: its tests, the corrupted-stream tests, the soak
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_lost_pictures.py tests/test_gpu_everything.py tests/test_gpu_foreign.py tests/test_gpu_decoder.py tests/test_gpu_random_access.py -q -m gpu -n 4 2>&1 | tail -3
timeout 2400 python tools/measure/soak_lost_pictures.py 401 800 2>&1 | grep -v "never arrived" | tail -4
