#!/bin/bash
# round 6: VUI extras / picture rate location; the random access suite
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_random_access.py -q -m gpu -n 4 > gpurun_out/r06_ra_tests.txt 2>&1; grep -E "^FAILED|passed|failed|Error|assert" gpurun_out/r06_ra_tests.txt | head -40
