#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
KVAZZUP_AMD_TIMELINE=/tmp/tl.txt KVAZZUP_BENCH_NOPROF=1 python bench.py --host-io --no-cpu-baseline --no-secondary --no-preset-line --streams-per-gpu 0 --repeats 1 --steps 4 --warmup 2 2>/dev/null | tail -1 | cut -c1-120
python tools/debug/dec_thread_raw.py /tmp/tl.txt > gpurun_out/r06_dec_thread_raw.txt; cat gpurun_out/r06_dec_thread_raw.txt
