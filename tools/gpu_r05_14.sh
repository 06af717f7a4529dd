#!/bin/bash
# intra analysis in the vertical form: encoder parity suites, isolated kernel times (1080p, 4K), the all-intra rate
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_filters.py tests/test_gpu_configs.py tests/test_gpu_golden.py -m gpu -x -q -n 3 --deselect tests/test_gpu_configs.py::test_config3_bench_command_with_two_ranks 2>&1 | tail -4
bash tools/kstats_iso.sh 1080p r05g_iso1080p --streams-per-gpu 0 2>&1 | grep -E "k_intra_analyse|k_intra_recon"
bash tools/kstats_iso.sh 4k r05g_iso4k --streams-per-gpu 0 --no-preset-line 2>&1 | grep -E "k_intra_analyse"
for i in 1 2 3; do KVAZZUP_BENCH_NOPROF=1 timeout 300 python bench.py --steps 4 --warmup 1 --repeats 1 --no-host-boundary --no-cpu-baseline --secondary-steps 1 --streams-per-gpu= 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('all_intra', d['all_intra']['value'], d['all_intra']['runs_fps'], 'default_mode', d['default_mode']['value'])"; done
