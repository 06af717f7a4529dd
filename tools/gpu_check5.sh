#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -q > gpurun_out/r02_configs.log 2>&1; echo "configs rc $?"; tail -30 gpurun_out/r02_configs.log | cut -c1-500
