#!/bin/bash
# round 6: no_output_of_prior_pics_flag through the HIP decoder; the random access suite again; the reordering suites
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_random_access.py -q -m gpu -n 4 > gpurun_out/r06_ra_tests.txt 2>&1; grep -E "^FAILED|passed|failed|Error" gpurun_out/r06_ra_tests.txt | head -40
timeout 1500 python -m pytest tests/test_gpu_foreign.py tests/test_gpu_everything.py tests/test_gpu_slices.py tests/test_gpu_decoder.py tests/test_gpu_hash.py tests/test_gpu_golden.py -q -m gpu -n 4 2>&1 | tail -3

