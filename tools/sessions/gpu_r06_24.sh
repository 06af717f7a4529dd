#!/bin/bash
# round 6: boundaries closed to the in-loop filters (Kvazaar's tiles; slices with the flag off) -- parity, then the other decoder suites and the filters
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_lfacross.py -q -n 3 2>&1 | tail -12
timeout 1500 python -m pytest tests/test_gpu_foreign.py tests/test_gpu_decoder.py tests/test_gpu_ctb.py tests/test_gpu_slices.py tests/test_gpu_mincb.py tests/test_golden_streams.py tests/test_gpu_filters.py tests/test_gpu_batch.py -m gpu -q -n 3 2>&1 | tail -4
