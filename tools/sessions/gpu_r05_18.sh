#!/bin/bash
# the decoder's early upload of 4x4 record rows: decoder / filter / foreign-stream parity suites, the OWF-0 timeline with and without (KVAZZUP_AMD_DEC_EARLY_UP=0)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_foreign.py tests/test_gpu_filters.py tests/test_gpu_golden.py tests/test_gpu_configs.py -m gpu -x -q -n 3 --deselect tests/test_gpu_configs.py::test_config3_bench_command_with_two_ranks 2>&1 | tail -3
for v in 1 0 1 0; do
  echo "== KVAZZUP_AMD_DEC_EARLY_UP=$v"
  KVAZZUP_AMD_DEC_EARLY_UP=$v timeout 300 python tools/measure/owf0_timeline.py 1080p 60 2>&1 | grep -E "dlaunch1  -> dec1|dec0      -> dlaunch0|feed0 -> out1|total delay"
  KVAZZUP_AMD_DEC_EARLY_UP=$v timeout 300 python tools/measure/owf0_timeline.py 4k 30 2>&1 | grep -E "dlaunch1  -> dec1|dec0      -> dlaunch0|feed0 -> out1|total delay"
done
