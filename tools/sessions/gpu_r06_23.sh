#!/bin/bash
# round 6: minimum coding blocks of 16 / 32 samples in the decoder -- parity, then the other decoder suites
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_mincb.py -q -n 3 2>&1 | tail -12
timeout 1500 python -m pytest tests/test_gpu_foreign.py tests/test_gpu_decoder.py tests/test_gpu_ctb.py tests/test_gpu_slices.py tests/test_golden_streams.py -m gpu -q -n 3 2>&1 | tail -4
