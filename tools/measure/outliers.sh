#!/bin/bash
# tools/measure/outliers.sh [4k|1080p] -- the slowest launches of every chain kernel in a kernel trace of the bench's headline leg, and WHAT SHARED THE CHIP with
# each of them (every kernel of any queue that overlaps it in time).  VERDICT r5 item 9: k_inter_recon max 938 us against 61 us on average at 4K.
R=${GRAFT_REPO_ROOT:-$PWD}; wl=${1:-4k}; shift
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/outl
KVAZZUP_BENCH_NOPROF=1 timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/outl -o p -- python3 $R/bench.py --workload $wl --no-cpu-baseline --no-secondary --no-host-boundary --no-preset-line --streams-per-gpu 0 --repeats 1 --steps 6 --warmup 1 "$@" > /tmp/outl.log 2>&1
tail -c 200 /tmp/outl.log | head -c 150; echo
f=$(find /tmp/outl -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections, bisect
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "kvzx::" not in n: continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("kvzx::", "").replace("void ", "")[:40], r.get("Queue_Id", "")))
rows.sort()
by = collections.defaultdict(list)
for r in rows: by[r[2]].append(r)
starts = [r[0] for r in rows]
print("%d kernels" % len(rows))
for name, rs in sorted(by.items(), key=lambda kv: -sum(e - s for s, e, _, _ in kv[1])):
    d = sorted(e - s for s, e, _, _ in rs); med = d[len(d) // 2]
    worst = sorted(rs, key=lambda r: r[0] - r[1])[:3]
    if (worst[0][1] - worst[0][0]) < 2.5 * med or len(rs) < 8: continue
    print("%-40s n %4d  median %7.1f us  p90 %7.1f  max %7.1f" % (name, len(rs), med / 1e3, d[len(d) * 9 // 10] / 1e3, d[-1] / 1e3))
    for s, e, _, q in worst:
        print("    %.1f us on queue %s; beside it:" % ((e - s) / 1e3, q))
        i = bisect.bisect_left(starts, s - 3_000_000)
        for s2, e2, n2, q2 in rows[i:]:
            if s2 > e: break
            if e2 < s or (s2, e2, n2, q2) == (s, e, name, q): continue
            print("        q%-3s %-40s %8.1f us (from %+8.1f to %+8.1f us of its %.1f)" % (q2, n2, (e2 - s2) / 1e3, (s2 - s) / 1e3, (e2 - s) / 1e3, (e - s) / 1e3))
PY
