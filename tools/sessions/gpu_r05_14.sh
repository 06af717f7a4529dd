#!/bin/bash
# intra analysis: encoder parity suites, isolated kernel times (1080p, 4K; KVAZZUP_AMD_ANALYSE_PER_CU=0 = the kernel itself, uncapped)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_filters.py tests/test_gpu_configs.py tests/test_gpu_golden.py tests/test_gpu_decoder.py -m gpu -x -q -n 3 --deselect tests/test_gpu_configs.py::test_config3_bench_command_with_two_ranks 2>&1 | tail -4
KVAZZUP_AMD_ANALYSE_PER_CU=0 bash tools/kstats_iso.sh 1080p r05g_iso1080p --streams-per-gpu 0 2>&1 | grep -E "k_intra_analyse"
bash tools/kstats_iso.sh 1080p r05g_iso1080p --streams-per-gpu 0 2>&1 | grep -E "k_intra_analyse"
KVAZZUP_AMD_ANALYSE_PER_CU=0 bash tools/kstats_iso.sh 4k r05g_iso4k --streams-per-gpu 0 --no-preset-line 2>&1 | grep -E "k_intra_analyse"
