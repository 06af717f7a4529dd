#!/bin/bash
# full GPU suite, then the round's profile set and the driver's bench line
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out; r=${ROUND:-r05}
timeout 2400 python -m pytest tests -m gpu -x -q -n 3 2>&1 | tail -4
bash tools/gpu_profiles.sh > gpurun_out/profiles.log 2>&1; tail -5 gpurun_out/profiles.log
cd $R; timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${r}_bench_driver_line.json 2> gpurun_out/${r}_bench_driver_line.err; tail -c 1500 gpurun_out/${r}_bench_driver_line.json
