#!/bin/bash
# round 6: the slice assembly fix (soak seed 1578) -- slice suites, lost segments, soaks again
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_slices.py tests/test_gpu_foreign.py tests/test_gpu_everything.py tests/test_gpu_decoder.py tests/test_gpu_lost_pictures.py tests/test_gpu_random_access.py tests/test_gpu_configs.py -q -m gpu -n 4 2>&1 | tail -3
{ timeout 1500 python tools/measure/soak_lost_pictures.py 6001 6400 2>&1 | grep -v "never arrived" | tail -3
  timeout 1500 python tools/measure/soak_everything.py 9001 9300 2>&1 | grep -v "never arrived" | tail -2
} > gpurun_out/r06_soaks_more2.txt 2>&1; cut -c1-300 gpurun_out/r06_soaks_more2.txt
