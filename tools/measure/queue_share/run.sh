#!/bin/bash
# tools/measure/queue_share/run.sh -- N processes at once, each with 8 streams at 3 priority levels (this library's shape), default and capped hardware queues
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R/tools/measure/queue_share
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/queue_share queue_share.hip 2>&1 | grep -v warning | head -3
for cap in default 2 1; do
  for n in 1 2 3 5 8; do
    echo "== $n process(es), GPU_MAX_HW_QUEUES=$cap"
    for i in $(seq 1 $n); do
      if [ "$cap" = default ]; then /tmp/queue_share 8 3 1500 "n=$n" & else GPU_MAX_HW_QUEUES=$cap /tmp/queue_share 8 3 1500 "n=$n" & fi
    done
    wait
  done
done
