#!/bin/bash
# round 6: the decoder's download queued at launch (behind the picture's event) instead of when a later call finds the event complete -- the host-boundary leg A/B
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
hb() { KVAZZUP_BENCH_NOPROF=1 python bench.py --host-io --no-cpu-baseline --no-secondary --no-preset-line --streams-per-gpu 0 --repeats 3 --steps 10 --warmup 2 $2 2>/dev/null | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); c=l['config']; print('$1', l['value'], c['runs_fps'], 'cores', c['host_cpu_cores_busy'], l['filter_busy_ms_per_picture'])"; }
{
for i in 1 2; do
hb at_launch ""
KVAZZUP_AMD_DEC_DL_AT_LAUNCH=0 hb on_query ""
done
hb at_launch_4k "--workload 4k"
KVAZZUP_AMD_DEC_DL_AT_LAUNCH=0 hb on_query_4k "--workload 4k"
KVAZZUP_AMD_TIMELINE=/tmp/tl.txt hb timeline "--repeats 1 --steps 4" ""
python tools/host_timeline_dec.py /tmp/tl.txt | head -12
timeout 600 python -m pytest tests/test_gpu_filters.py tests/test_gpu_decoder.py -q -n 3 2>&1 | tail -2
} > gpurun_out/r06_dl_at_launch_ab.txt 2>&1; cat gpurun_out/r06_dl_at_launch_ab.txt
