#!/bin/bash
# tools/measure/two_rank_neighbours.sh -- what makes `bench.py --gpus 2` on ONE GPU crawl beside pytest's workers (tests/test_gpu_configs.py): neighbours that
# burn CPU inside the job's 16-CPU quota, neighbours that merely hold HIP contexts with live streams, or both.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
two() { timeout 300 python bench.py --gpus 2 --steps 4 --warmup 1 --repeats 1 --no-cpu-baseline --no-host-boundary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('  two ranks: %.1f frames/s, cores busy %.1f, throttled %.0f ms of %.0f ms' % (d['value'], c['host_cpu_cores_busy'], c['host_cpu_throttled_ms'], d['ms_per_step']*d['steps']))"; }
burn() { python - <<'PY' &
import threading, time
def f():
    t=time.time()
    while time.time()-t<60: pass
ts=[threading.Thread(target=f) for _ in range(1)]
import multiprocessing as mp
ps=[mp.Process(target=f) for _ in range(6)]
[p.start() for p in ps]; [p.join() for p in ps]
PY
}
hold() { python - <<'PY' &
import sys, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, orc
from kvazzup_amd.codec import Encoder, Decoder
ge=Encoder(640,384,options=(("qp",32),("period",64),("me-range",8))); gd=Decoder()
for t in range(4):
    au,rec=ge.encode(orc.synth_frame(0,1,640,384,t)); gd.decode_au(au,t)
ge.close(); gd.close()          # (the library keeps the role streams of a closed instance)
time.sleep(60)
PY
}
echo "== alone"; two
echo "== beside three idle processes that hold HIP contexts and the library's streams"; hold; hold; hold; sleep 8; two; kill %1 %2 %3 2>/dev/null; wait 2>/dev/null
echo "== beside 18 CPU-burning processes (no GPU)"; burn; burn; burn; sleep 1; two; kill %1 %2 %3 2>/dev/null; wait 2>/dev/null
