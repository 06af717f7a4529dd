#!/bin/bash
# tools/measure/ab_env.sh VAR "v1 v2 ..." [bench args]: the same bench line under several values of one environment variable, alternating, on one box
VAR=$1; VALS=$2; shift 2
for rep in 1 2; do for v in $VALS; do
  env $VAR=$v timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 12 "$@" > gpurun_out/ab_${VAR}_${v}_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_${VAR}_${v}_$rep.json").read().strip().splitlines()[-1])
hb=d.get("host_boundary") or {}
print("$VAR=$v rep $rep resident", d["value"], d["config"]["runs_fps"], "host", hb.get("value"), hb.get("runs"), "parse", d["kernels_us"].get("host_cabac_parse"))
PY
done; done
