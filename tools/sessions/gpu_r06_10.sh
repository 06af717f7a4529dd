#!/bin/bash
# round 6, GPU session: coding tree blocks of 32 / 16 samples in the decoder (branch ctb): the new tests, then the decoder's existing suite (64-sample CTBs must not move)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ctb.py -q -n 3 -rf 2>&1 | grep -E "FAILED|passed|failed|samples differ|error flags|Error" | cut -c1-330 > gpurun_out/ctb_tests.txt; tail -40 gpurun_out/ctb_tests.txt
timeout 1500 python -m pytest tests/test_gpu_foreign.py tests/test_gpu_decoder.py tests/test_gpu_batch.py tests/test_golden_streams.py tests/test_gpu_filters.py -q -n 3 2>&1 | tail -3
