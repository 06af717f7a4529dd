#!/bin/bash
# round 6: lost access units (concealment v1) through the HIP decoder; the decoder suites behind it
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_lost_pictures.py -q -m gpu -n 4 > gpurun_out/r06_lost_tests.txt 2>&1; grep -E "^FAILED|passed|failed|Error|differ" gpurun_out/r06_lost_tests.txt | head -40
timeout 1500 python -m pytest tests/test_gpu_random_access.py tests/test_gpu_foreign.py tests/test_gpu_everything.py tests/test_gpu_slices.py tests/test_gpu_decoder.py tests/test_gpu_hash.py tests/test_gpu_golden.py tests/test_gpu_longterm.py -q -m gpu -n 4 2>&1 | tail -3
