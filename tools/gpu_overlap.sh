#!/bin/bash
# how much the kernels of the pipeline's streams overlap on the GPU: tools/gpu_overlap.sh <workload> [streams]   (kernel trace of a short bench run, analysed on the box;
# streams = K: the window analysed is the K-pipelines-in-one-process leg -- the longest run of kernels of the trace -- instead of the single stream's)
R=${GRAFT_REPO_ROOT:-$PWD}; wl=${1:-4k}; K=${2:-}
if [ -n "$K" ]; then multi="--streams-per-gpu $K --steps 2"; echo "== $wl, $K pipelines in one process"; else multi="--streams-per-gpu 0 --steps 3"; echo "== $wl, one pipeline"; fi
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ovl
KVAZZUP_BENCH_NOPROF=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ovl -o p -- python3 $R/bench.py --workload $wl --no-cpu-baseline --no-secondary --no-host-boundary --no-preset-line --repeats 1 --warmup 1 $multi > /tmp/ovl.log 2>&1
tail -c 300 /tmp/ovl.log | head -c 200; echo
f=$(find /tmp/ovl -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        n = r["Kernel_Name"]
        if "kvzx::" not in n: continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("kvzx::", "").replace("void ", ""), r.get("Queue_Id", "")))
rows.sort()
# the timed region: the longest run of dispatches without an idle gap above 3 ms (warm-up, flushes and the quality pass lie around it)
segs, cur, end = [], [], None
for r in rows:
    if end is not None and r[0] - end > 3_000_000: segs.append(cur); cur = []
    cur.append(r); end = r[1] if end is None else max(end, r[1])
segs.append(cur)
rows = max(segs, key=len)
t0, t1 = rows[0][0], max(r[1] for r in rows)
tot = sum(e - s for s, e, _, _ in rows)
ev = []
for s, e, _, _ in rows: ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = 0; depth = 0; last = t0; hist = collections.Counter()
for t, d in ev:
    if depth > 0: busy += t - last
    hist[depth] += t - last
    last = t; depth += d
wall = t1 - t0
print("window %.1f ms, %d dispatches; sum of kernel durations %.1f ms (%.2f x wall); some kernel running %.1f %% of the wall" % (wall / 1e6, len(rows), tot / 1e6, tot / wall, 100.0 * busy / wall))
print("time with k kernels in flight:", {k: "%.1f %%" % (100.0 * v / wall) for k, v in sorted(hist.items())})
for qid in sorted(set(r[3] for r in rows)):
    qr = [r for r in rows if r[3] == qid]
    busyq = sum(e - s for s, e, _, _ in qr)
    gaps = [b[0] - a[1] for a, b in zip(qr[:-1], qr[1:]) if b[0] > a[1]]
    small = [g for g in gaps if g < 50_000]
    names = collections.Counter(r[2] for r in qr).most_common(3)
    print("queue %s: %d kernels, busy %.1f %% of the wall, %d gaps below 50 us averaging %.1f us; mostly %s" % (qid, len(qr), 100.0 * busyq / wall, len(small), sum(small) / max(1, len(small)) / 1e3, ", ".join(n for n, _ in names)))
by = collections.defaultdict(lambda: [0, 0])
for s, e, n, _ in rows: by[n][0] += 1; by[n][1] += e - s
for n, (c, d) in sorted(by.items(), key=lambda x: -x[1][1])[:14]: print("  %-30s %6d x %8.1f us = %5.1f %% of wall" % (n[:30], c, d / c / 1e3, 100.0 * d / wall))
PY
rm -rf /tmp/ovl
