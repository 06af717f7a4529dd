#!/bin/bash
# what the OTHER codec's kernels do while an intra chain runs: tools/gpu_chain_shadow.sh [bench args]   (kernel trace of a short bench run, analysed on the box)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/shd
KVAZZUP_BENCH_NOPROF=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/shd -o p -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-host-boundary --repeats 1 --steps 4 --warmup 1 "$@" > /tmp/shd.log 2>&1
tail -c 300 /tmp/shd.log | head -c 200; echo
f=$(find /tmp/shd -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        n = r["Kernel_Name"]
        if "kvzx::" not in n: continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("kvzx::", "").replace("void ", "").split("<")[0]))
rows.sort()
def side(n): return "dec" if n.startswith("k_dec") else "enc"
for chain, other in (("k_intra_recon", "dec"), ("k_dec_intra", "enc")):
    spans = [(s, e) for s, e, n in rows if n == chain and e - s > 400_000]
    if not spans: continue
    inside = collections.defaultdict(list); outside = collections.defaultdict(list)
    started = []
    for s, e, n in rows:
        if side(n) != other or n in ("k_intra_recon", "k_dec_intra", "k_intra_analyse"): continue
        hit = any(a <= s < b for a, b in spans)
        (inside if hit else outside)[n].append(e - s)
    print("%d launches of %s, mean %.0f us.  Kernels of the %s side that START while one runs / otherwise:" % (len(spans), chain, sum(e - s for s, e in spans) / len(spans) / 1e3, "decoder's" if other == "dec" else "encoder's"))
    for n in sorted(set(inside) | set(outside)):
        a, b = inside.get(n, []), outside.get(n, [])
        print("  %-18s inside: %4d starts (%.1f per chain), mean %6.1f us   outside: %5d starts, mean %6.1f us" % (n, len(a), len(a) / len(spans), sum(a) / max(1, len(a)) / 1e3, len(b), sum(b) / max(1, len(b)) / 1e3))
    if chain == "k_dec_intra" or True:
        a, b = spans[len(spans) // 2]
        print("  --- every kernel around one %s (us from its start; it ends at %.0f):" % (chain, (b - a) / 1e3))
        for s2, e2, n2 in rows:
            if s2 >= a - 150_000 and s2 <= b + 250_000: print("    %8.1f .. %8.1f  %s" % ((s2 - a) / 1e3, (e2 - a) / 1e3, n2))
    tot = sum(e - s for s, e in spans); wall = rows[-1][1] - rows[0][0]
    print("  (the chains cover %.1f %% of the traced window)" % (100.0 * tot / wall))
PY
rm -rf /tmp/shd
