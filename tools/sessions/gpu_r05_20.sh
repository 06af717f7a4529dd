#!/bin/bash
# decoder frame threads of the headline leg (bench.py --decoder-frame-threads; default 32) under the 16-CPU quota: interleaved runs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; OUT=gpurun_out/r05_frame_threads.txt; : > $OUT
one() { KVAZZUP_BENCH_NOPROF=1 python bench.py --steps 40 --warmup 3 --repeats 1 --decoder-frame-threads $1 --no-host-boundary --no-preset-line --no-cpu-baseline --no-secondary --streams-per-gpu= 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('threads $1:', d['value'])" >> $OUT; }
for rep in 1 2 3; do for n in 32 12 16 20 24 40; do one $n; done; done
sort $OUT | awk '{print}'; python - <<'P'
import re, collections
v = collections.defaultdict(list)
for l in open("gpurun_out/r05_frame_threads.txt"):
    m = re.match(r"threads (\d+): ([\d.]+)", l)
    if m: v[int(m.group(1))].append(float(m.group(2)))
for n in sorted(v): print("threads %2d: mean %.0f  (%s)" % (n, sum(v[n]) / len(v[n]), ", ".join("%.0f" % x for x in v[n])))
P
