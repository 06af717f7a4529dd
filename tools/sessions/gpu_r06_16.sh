#!/bin/bash
# round 6: pictures of free slices (segments that begin at any coding tree block, independent slices inside a tile) -- parity against the checker, then every
# decoder suite again (the assembly of Kvazaar's forms shares the code)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_slices.py -x -q -n 3 2>&1 | tail -15
timeout 1500 python -m pytest tests/test_gpu_foreign.py tests/test_gpu_decoder.py tests/test_gpu_ctb.py tests/test_gpu_golden.py tests/test_gpu_filters.py -q -n 3 2>&1 | tail -5
