#!/bin/bash
# Per hardware queue: every kernel's average duration and the idle gap in front of it (start minus the end of the queue's previous kernel), from a kernel
# trace of a short bench run -- where a serial chain of short kernels loses its time.  tools/measure/chain_gaps.sh [default|headline] [extra bench args]
R=${GRAFT_REPO_ROOT:-$PWD}; mode=${1:-default}; shift
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/gaps
extra=""
[ "$mode" = "default" ] && extra="--custom preset=veryfast --custom bitrate=1000000 --custom rc-algorithm=lambda"
[ "$mode" = "intra" ] && extra="--custom period=1"
KVAZZUP_BENCH_NOPROF=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/gaps -o p -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-host-boundary --streams-per-gpu 0 --repeats 1 --steps 3 --warmup 1 $extra "$@" > /tmp/gaps.log 2>&1
tail -c 300 /tmp/gaps.log | head -c 250; echo
f=$(find /tmp/gaps -name "*kernel_trace.csv" | head -1); m=$(find /tmp/gaps -name "*memory_copy_trace.csv" | head -1)
python3 - "$f" "$m" <<'PY'
import csv, sys, collections
copies = []
try:
    for r in csv.DictReader(open(sys.argv[2])):
        copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?")[:20] + " %s B" % (r.get("Bytes", r.get("Size", "?"))), "copy"))
except Exception as e:
    print("no copy trace:", e)
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        n = r["Kernel_Name"]
        if "kvzx::" not in n: continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("kvzx::", "").replace("void ", "")[:34], r.get("Queue_Id", "")))
rows.sort()
segs, cur, end = [], [], None
for r in rows:
    if end is not None and r[0] - end > 3_000_000: segs.append(cur); cur = []
    cur.append(r); end = r[1] if end is None else max(end, r[1])
segs.append(cur)
rows = max(segs, key=len)
t0, t1 = rows[0][0], max(r[1] for r in rows)
print("window %.2f ms, %d kernels" % ((t1 - t0) / 1e6, len(rows)))
byq = collections.defaultdict(list)
for r in rows: byq[r[3]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    st = collections.defaultdict(lambda: [0, 0, 0, 0, []])
    busy = 0
    for i, (s, e, n, _) in enumerate(rs):
        a = st[n]; a[0] += 1; a[1] += e - s; busy += e - s
        if i: g = s - rs[i - 1][1]; a[2] += max(0, g); a[3] += 1; a[4].append(max(0, g))
    print("queue %s: %d kernels, busy %.1f %% of the window" % (q, len(rs), 100.0 * busy / (t1 - t0)))
    for n, a in sorted(st.items(), key=lambda kv: -kv[1][1]):
        gs = sorted(a[4]) or [0]
        print("   %-36s n %5d  avg %8.1f us  gap in front %7.1f us (median %.1f, 90 %% %.1f, max %.1f)" % (n, a[0], a[1] / a[0] / 1e3, a[2] / max(1, a[3]) / 1e3, gs[len(gs) // 2] / 1e3, gs[len(gs) * 9 // 10] / 1e3, gs[-1] / 1e3))
# a 1 ms excerpt from the last quarter of the window as a timeline (all queues and the copies) -- or, CHAIN_GAPS_AT_IDR=1, from 0.4 ms before the window's second intra picture's chain
import os
mid = t0 + (t1 - t0) * 3 // 4
if os.environ.get("CHAIN_GAPS_AT_IDR"):
    idr = [r for r in rows if r[2].startswith("k_intra_recon<false, false")]
    if len(idr) > 1: mid = idr[1][0] - 400_000
    span = 2_400_000
else:
    span = 1_000_000
print("--- timeline excerpt (us from its start; queue)")
for s, e, n, q in sorted(rows + [c for c in copies if t0 <= c[0] <= t1]):
    if mid <= s < mid + span: print("%8.1f .. %8.1f  q%-5s %s" % ((s - mid) / 1e3, (e - mid) / 1e3, q, n))
PY
