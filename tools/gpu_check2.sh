#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_filters.py tests/test_gpu_golden.py -m gpu -x -q > gpurun_out/r02_dec_own.log 2>&1; echo "own rc $?"; tail -15 gpurun_out/r02_dec_own.log
timeout 1200 python -m pytest tests/test_gpu_foreign.py -m gpu -q > gpurun_out/r02_dec_foreign.log 2>&1; echo "foreign rc $?"; tail -60 gpurun_out/r02_dec_foreign.log | cut -c1-400
