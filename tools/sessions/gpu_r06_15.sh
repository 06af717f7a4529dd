#!/bin/bash
# round 6: soaks on the final library -- 3 000 pictures of uvgComm's default mode (me-source: the search ahead on the input stream, owf 6, rate control v2) against the
# checker, 1 500 all-intra pictures through the filter graph, the corrupted-stream tests with 20 000 trials each
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
{ timeout 1500 python tools/measure/soak_default_mode.py 3000 2>&1 | tail -3
  timeout 1200 python tools/measure/soak_all_intra.py 2>&1 | tail -3
  KVZ_FUZZ_TRIALS=20000 timeout 1500 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_foreign.py tests/test_gpu_everything.py -q -n 3 -k "corrupt or hostile or fuzz or garbage or truncat" 2>&1 | tail -3
} > gpurun_out/r06_soaks.txt 2>&1; cat gpurun_out/r06_soaks.txt
