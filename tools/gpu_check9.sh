#!/bin/bash
# subme: parity, then bench lines with / without the refinement
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_golden.py tests/test_gpu_filters.py -m gpu -x -q -k "subme or golden or arithmetic or filter" > gpurun_out/r02_pytest_subme.log 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/r02_pytest_subme.log | cut -c1-900
