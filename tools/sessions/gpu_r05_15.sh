#!/bin/bash
# the intra mode search capped at N workgroups per compute unit (KVAZZUP_AMD_ANALYSE_PER_CU; 0 = every wave slot): the headline and 4K, interleaved runs
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
OUT=gpurun_out/r05_analyse_cap.txt; : > $OUT
one() {
  KVAZZUP_AMD_ANALYSE_PER_CU=$1 KVAZZUP_BENCH_NOPROF=1 timeout 600 python bench.py --steps 40 --warmup 3 --repeats 1 --no-host-boundary --no-cpu-baseline --no-preset-line --secondary-steps 16 --streams-per-gpu= 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('per_cu $1: value', d['value'], ' 4k', d['secondary']['value'])" >> $OUT
}
for rep in 1 2 3 4; do for n in 0 3 4; do one $n; done; for n in 4 3 0; do one $n; done; done
cat $OUT
python - <<'P'
import re, collections
v = collections.defaultdict(list); k4 = collections.defaultdict(list)
for l in open("gpurun_out/r05_analyse_cap.txt"):
    m = re.match(r"per_cu (\d): value ([\d.]+)\s+4k ([\d.]+)", l)
    if m: v[m.group(1)].append(float(m.group(2))); k4[m.group(1)].append(float(m.group(3)))
for n in sorted(v): print("per_cu", n, "1080p mean %.0f median %.0f   4k mean %.0f median %.0f" % (sum(v[n]) / len(v[n]), sorted(v[n])[len(v[n]) // 2], sum(k4[n]) / len(k4[n]), sorted(k4[n])[len(k4[n]) // 2]))
P
