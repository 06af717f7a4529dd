#!/bin/bash
# several pipelines in one process on one GPU: shared role streams (default) against one set of streams per instance (round 3)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
B="python bench.py --steps ${STEPS:-8} --warmup 2 --repeats 1 --no-secondary --no-host-boundary --no-cpu-baseline --no-preset-line --streams-per-gpu ${KS:-2,4}"
for rep in 1 2; do
  for share in 1 0; do
    KVAZZUP_AMD_SHARE_STREAMS=$share timeout 600 $B > gpurun_out/multi_share${share}_$rep.json 2> gpurun_out/multi_share${share}_$rep.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/multi_share${share}_$rep.json").read().strip().splitlines()[-1])
print("share=${share} rep=$rep single", d["value"], "cores", d["config"]["host_cpu_cores_busy"], "multi", [(m.get("streams"), m.get("value"), m.get("host_cpu_cores_busy"), m.get("error")) for m in d["streams_per_gpu"]])
PY
  done
done
