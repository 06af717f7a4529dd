#!/bin/bash
# the host-boundary leg under the kernel + memory-copy trace: how long the PCIe copies take beside the kernels, and what each queue does
# tools/gpu_hosttrace.sh <workload | enconly | deconly>   (enconly / deconly: tools/measure/encoder_only.py / decoder_only.py, host pictures)
R=${GRAFT_REPO_ROOT:-$PWD}; wl=${1:-1080p}
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/htr
if [ "$wl" = "enconly" ]; then
HOST_ONLY=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/htr -o p -- python3 $R/tools/measure/encoder_only.py > /tmp/htr.log 2>&1
elif [ "$wl" = "deconly" ]; then
HOST_ONLY=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/htr -o p -- python3 $R/tools/measure/decoder_only.py > /tmp/htr.log 2>&1
else
KVAZZUP_BENCH_NOPROF=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/htr -o p -- python3 $R/bench.py --workload $wl --host-io --no-cpu-baseline --no-secondary --steps 3 --warmup 1 --repeats 1 > /tmp/htr.log 2>&1
fi
tail -c 400 /tmp/htr.log | head -c 300; echo
ls /tmp/htr/*/ 2>/dev/null | head
k=$(find /tmp/htr -name "*kernel_trace.csv" | head -1); m=$(find /tmp/htr -name "*memory_copy_trace.csv" | head -1)
python3 - "$k" "$m" <<'PY'
import csv, sys, collections
kr = list(csv.DictReader(open(sys.argv[1])))
mr = list(csv.DictReader(open(sys.argv[2])))
print("memory copy columns:", list(mr[0].keys()) if mr else None)
ev = []
for r in kr:
    n = r["Kernel_Name"]
    if "at::native" not in n: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + n.split("(")[0].replace("kvzx::", "").replace("void ", "")[:28], 0))
for r in mr:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "?")[:24], int(r.get("Bytes", r.get("Size", 0)) or 0)))
ev.sort()
# the timed region: last long run without a 3 ms gap
segs, cur, end = [], [], None
for e in ev:
    if end is not None and e[0] - end > 3_000_000: segs.append(cur); cur = []
    cur.append(e); end = e[1] if end is None else max(end, e[1])
segs.append(cur)
ev = max(segs, key=len)
t0, t1 = ev[0][0], max(e[1] for e in ev)
print("window %.2f ms, %d events" % ((t1 - t0) / 1e6, len(ev)))
st = collections.defaultdict(lambda: [0, 0, 0, 0])
for s, e, n, b in ev:
    key = n if not n.startswith("C ") else n + (" big" if b > 1000000 else " small")
    a = st[key]; a[0] += 1; a[1] += e - s; a[2] += b; a[3] = max(a[3], e - s)
for k, a in sorted(st.items(), key=lambda kv: -kv[1][1]):
    print("%-44s n %5d  avg %8.1f us  max %8.1f us  total %7.2f ms  %s" % (k, a[0], a[1] / a[0] / 1e3, a[3] / 1e3, a[1] / 1e6, ("%.1f GB/s" % (a[2] / a[1])) if a[2] else ""))
# do the copy engines run side by side?  sum of the copies' durations against the time at least one copy runs, per direction and together
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = None, None
    for s, e in iv:
        if ce is None or s > ce:
            if ce is not None: tot += ce - cs
            cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ((ce - cs) if ce is not None else 0)
h2d = [(s, e) for s, e, n, b in ev if n.startswith("C ") and "HOST_TO_DEV" in n]; d2h = [(s, e) for s, e, n, b in ev if n.startswith("C ") and "DEVICE_TO_HO" in n]
print("copies: H2D sum %.2f ms union %.2f ms | D2H sum %.2f ms union %.2f ms | both: sum %.2f ms union %.2f ms (window %.2f ms)"
      % (sum(e - s for s, e in h2d) / 1e6, union(h2d) / 1e6, sum(e - s for s, e in d2h) / 1e6, union(d2h) / 1e6, sum(e - s for s, e in h2d + d2h) / 1e6, union(h2d + d2h) / 1e6, (t1 - t0) / 1e6))
# a 600 us excerpt from the middle of the window, as a timeline
mid = t0 + (t1 - t0) * 3 // 4
print("--- timeline excerpt (us from the excerpt's start)")
for s, e, n, b in ev:
    if s >= mid and s < mid + 900_000: print("%8.1f .. %8.1f  %-40s %s" % ((s - mid) / 1e3, (e - mid) / 1e3, n, b or ""))
PY
