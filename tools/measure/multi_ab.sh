#!/bin/bash
# several pipelines in one process on one GPU.  MODES: space-separated "share,batch" pairs -- KVAZZUP_AMD_SHARE_STREAMS (0: one set of HIP streams per
# instance, round 3) and KVAZZUP_AMD_BATCH (0: every decoder launches its own pictures) -- run alternately, REPS times
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
B="python bench.py --steps ${STEPS:-8} --warmup 2 --repeats 1 --no-secondary --no-host-boundary --no-cpu-baseline --no-preset-line --streams-per-gpu ${KS:-2,4}"
for rep in $(seq 1 ${REPS:-2}); do
  for mode in ${MODES:-1,1 1,0}; do
    share=${mode%,*}; batch=${mode#*,}
    f=gpurun_out/multi_s${share}b${batch}_$rep
    KVAZZUP_AMD_SHARE_STREAMS=$share KVAZZUP_AMD_BATCH=$batch timeout 600 $B > $f.json 2> $f.err
    python - <<PY
import json
d = json.loads(open("$f.json").read().strip().splitlines()[-1])
print("share=$share batch=$batch rep=$rep single", d["value"], "cores", d["config"]["host_cpu_cores_busy"])
for m in d["streams_per_gpu"]:
    print("   ", m.get("streams"), m.get("value"), "cores", m.get("host_cpu_cores_busy"), m.get("error"), json.dumps(m.get("decoder_batches")))
PY
  done
done
