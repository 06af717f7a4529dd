#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for v in -1 1 2 3; do echo "== KVAZZUP_AMD_SIGNAL_VARIANT=$v"; KVAZZUP_AMD_SIGNAL_VARIANT=$v bash tools/kstats_iso.sh 1080p r05_sig$v --streams-per-gpu 0 --no-preset-line 2>&1 | grep -E "k_inter_signal"; done
echo "== all-intra with the second input stream"
for v in "" "KVAZZUP_AMD_IDR_INLINE=1 KVAZZUP_AMD_DEC_ONE_CHAIN=1"; do env $v timeout 300 python bench.py --steps 4 --warmup 1 --repeats 2 --no-cpu-baseline --no-secondary --no-host-boundary --streams-per-gpu 0 --no-preset-line --custom period=1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['config']['runs_fps'], d['config']['host_cpu_cores_busy'], {k:d['kernels_us'][k] for k in ('k_intra_recon','k_dec_intra','k_intra_analyse') if k in d['kernels_us']})"; done
timeout 1200 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_configs.py tests/test_gpu_filters.py -m gpu -x -q -n 3 --deselect tests/test_gpu_configs.py::test_config3_bench_command_with_two_ranks 2>&1 | tail -3
