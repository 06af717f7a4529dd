#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q -n 3 2>&1 | tail -4
cat gpurun_out/two_rank_crawl.jsonl 2>/dev/null | cut -c1-600
for v in "" "KVAZZUP_AMD_DEC_ONE_CHAIN=1" "KVAZZUP_AMD_IDR_INLINE=1" "KVAZZUP_AMD_DEC_ONE_CHAIN=1 KVAZZUP_AMD_IDR_INLINE=1"; do
  echo "== all-intra 1080p, $v"
  env $v timeout 300 python bench.py --steps 4 --warmup 1 --repeats 2 --no-cpu-baseline --no-secondary --no-host-boundary --streams-per-gpu 0 --no-preset-line --custom period=1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['config']['runs_fps'], d['config']['host_cpu_cores_busy'], {k:d['kernels_us'][k] for k in ('k_intra_recon','k_dec_intra','host_cabac_parse') if k in d['kernels_us']})"
done
