#!/bin/bash
# tools/measure/cpu_placement.sh -- where the host threads run: the box's topology, the cgroup's CPU quota and how often it throttled the bench,
# and the bench line with the process confined to 16 / 24 / 32 logical CPUs chosen by hand (one per physical core, NUMA node of the GPU or not)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/r05_cpu_placement.txt
: > $OUT
{
  lscpu | grep -E "Model name|Socket|Core|Thread|NUMA|MHz"
  echo "-- cgroup"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null; nproc
  echo "-- allowed"; taskset -pc $$
  echo "-- gpu numa"; cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' '; echo
  echo "-- siblings of cpu0"; cat /sys/devices/system/cpu/cpu0/topology/thread_siblings_list
} >> $OUT 2>&1
stat0() { grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; }
one() {   # label, cpu list ("" = unconfined)
  local s0="$(stat0)"
  local cmd="python bench.py --steps 40 --warmup 4 --no-host-boundary --no-preset-line --no-cpu-baseline --no-secondary --streams-per-gpu="
  if [ -n "$2" ]; then cmd="taskset -c $2 $cmd"; fi
  local v=$(KVAZZUP_BENCH_NOPROF=1 $cmd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('host_cpu_cores_busy'))")
  echo "$1: $v   | before: $s0 | after: $(stat0)" >> $OUT
}
NODE=$(cat /sys/class/drm/card*/device/numa_node 2>/dev/null | sort -n | tail -1)
[ "$NODE" -lt 0 ] 2>/dev/null && NODE=0
CPUS=$(cat /sys/devices/system/node/node$NODE/cpulist 2>/dev/null)
echo "-- node $NODE cpus $CPUS" >> $OUT
# first logical CPU of each physical core of that node
FIRST=$(for c in $(python - <<P
import re
s=open('/sys/devices/system/node/node$NODE/cpulist').read().strip()
o=[]
for part in s.split(','):
    a,_,b=part.partition('-'); o+=range(int(a),int(b or a)+1)
print(' '.join(map(str,o)))
P
); do sib=$(cat /sys/devices/system/cpu/cpu$c/topology/thread_siblings_list | cut -d, -f1 | cut -d- -f1); [ "$sib" = "$c" ] && echo $c; done | tr '\n' ' ')
set -- $FIRST
L16=$(echo $FIRST | tr ' ' '\n' | head -16 | paste -sd,)
L24=$(echo $FIRST | tr ' ' '\n' | head -24 | paste -sd,)
L32=$(echo $FIRST | tr ' ' '\n' | head -32 | paste -sd,)
echo "-- 16 cores: $L16" >> $OUT
one free ""
for rep in 1 2 3; do
  one free ""
  one pinned16 "$L16"
  one pinned24 "$L24"
  one pinned32 "$L32"
done
cat $OUT
