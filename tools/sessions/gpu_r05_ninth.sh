#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 600 env KVAZZUP_BENCH_THREADS=1 CPU_SAMPLER_REGION=1 CPU_SAMPLER_OUT=$R/gpurun_out/cpu_samples.txt KVAZZUP_AMD_LIBRARY=$R/kvazzup_amd/libkvazzup_amd_g.so LD_PRELOAD=$R/tools/libcpusampler.so python bench.py --steps 100 --warmup 5 --repeats 1 --no-cpu-baseline --no-secondary --no-host-boundary --streams-per-gpu 0 --no-preset-line > gpurun_out/cpuprof_bench.json 2> gpurun_out/cpuprof_bench.err; echo "rc $?"
python -c "
import json; d=json.load(open('gpurun_out/cpuprof_bench.json')); print(d['value'], d['config']['host_cpu_cores_busy'])"
grep '^thread' gpurun_out/cpuprof_bench.err | head -60
timeout 300 python tools/cpu_sampler_report.py gpurun_out/cpu_samples.txt kvazzup_amd/libkvazzup_amd_g.so 90 < /dev/null > gpurun_out/r05_cpu_profile.txt 2>&1; head -150 gpurun_out/r05_cpu_profile.txt
rm -f gpurun_out/cpu_samples.txt
