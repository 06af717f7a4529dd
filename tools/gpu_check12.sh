#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_golden.py -m gpu -x -q -k "subme or golden" > gpurun_out/r02_pytest_subme.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r02_pytest_subme.log | cut -c1-900
bash tools/kstats_iso.sh 1080p r02_iso1080p_subme4 --subme 4
bash tools/kstats_iso.sh 1080p r02_iso1080p_subme4_full --subme 4 --full-search
