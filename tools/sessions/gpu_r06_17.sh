#!/bin/bash
# round 6: after the decoder's free slices (k_dec_intra: reference samples in two runs, the row piece above-right loaded without the one above) -- isolated kernel
# times of the decoder's intra chain at 1080p (all-intra: every picture an IDR) against profiles/r06_iso1080p_kernel_stats.csv, and the bench line
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
bash tools/kstats_iso.sh 1080p r06c_iso1080p --streams-per-gpu 0 --no-preset-line 2>&1 | grep -E "k_dec_"
cd $R; bash tools/kstats_iso.sh 1080p r06c_iso1080p_intra --streams-per-gpu 0 --no-preset-line --custom period=1 2>&1 | grep -E "k_dec_"
cd $R; timeout 900 python bench.py > gpurun_out/r06c_bench_driver_line.json 2> gpurun_out/r06c_bench.err; tail -c 600 gpurun_out/r06c_bench.err; python - <<'PY'
import json
l = json.loads(open("gpurun_out/r06c_bench_driver_line.json").read().strip().splitlines()[-1])
print("value", l["value"], "rates", {k: (v if not isinstance(v, dict) else {a: b for a, b in v.items() if isinstance(b, (int, float))}) for k, v in l["rates"].items()})
PY
