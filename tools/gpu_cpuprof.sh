#!/bin/bash
# CPU profile of the default bench (sampling profiler, library built with -g): tools/gpu_cpuprof.sh; the report is made where
# llvm-symbolizer and the -g library are:  python tools/cpu_sampler_report.py gpurun_out/cpu_samples.txt kvazzup_amd/libkvazzup_amd_g.so
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
make -s -C $R/kvazzup_amd/csrc BUILD=build_g TARGET=../libkvazzup_amd_g.so EXTRA=-g > /dev/null 2>&1   # (the -g twin of the library: built here, or before the gpurun call to save box time)
timeout 600 env KVAZZUP_BENCH_THREADS=1 CPU_SAMPLER_REGION=1 CPU_SAMPLER_OUT=$R/gpurun_out/cpu_samples.txt KVAZZUP_AMD_LIBRARY=$R/kvazzup_amd/libkvazzup_amd_g.so LD_PRELOAD=$R/tools/libcpusampler.so python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-secondary --no-host-boundary --no-preset-line --streams-per-gpu= > gpurun_out/cpuprof_bench.json 2> gpurun_out/cpuprof_bench.err; echo "rc $?"
tail -c 600 gpurun_out/cpuprof_bench.json | cut -c1-300
ls -la gpurun_out/cpu_samples.txt
timeout 200 python tools/cpu_sampler_report.py gpurun_out/cpu_samples.txt kvazzup_amd/libkvazzup_amd_g.so 70 < /dev/null > gpurun_out/cpuprof_report.txt 2>&1; grep '^thread' gpurun_out/cpuprof_bench.err | head -45; head -110 gpurun_out/cpuprof_report.txt
