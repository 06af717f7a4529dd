#!/bin/bash
# first GPU pass of the round: parity tests, the driver's bench command, the rank launcher, the qsad microbenchmark
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest_gpu.log 2>&1; echo "pytest rc $?" 
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_driver.json 2> gpurun_out/r02_bench_driver.err; echo "bench rc $?"
timeout 600 python bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_gpus2.json 2> gpurun_out/r02_bench_gpus2.err; echo "bench2 rc $?"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/qsad_bench.hip -o /tmp/qsad_bench && /tmp/qsad_bench > gpurun_out/r02_qsad_bench.txt 2>&1
tail -3 gpurun_out/r02_pytest_gpu.log; cut -c1-600 gpurun_out/r02_bench_driver.json; cut -c1-300 gpurun_out/r02_bench_gpus2.json; cat gpurun_out/r02_qsad_bench.txt
