#!/usr/bin/env python3
"""tools/measure/wpp_critical_path.py -- what row-parallel (WPP) parsing of a picture can gain at best.  The product's parser (parse-only hook, ONE thread,
KVAZZUP_AMD_CTU_DUMP) times every coding tree unit of the headline clip's pictures; with unlimited threads a unit can start when the unit to its left and the one
above-right are done (the two-CTU lag of WPP: contexts, neighbours), so a picture takes as long as the heaviest path through that graph.  CPU only.

  python tools/measure/wpp_critical_path.py [--w 1920 --h 1080 --frames 8]"""
import argparse, collections, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser(); ap.add_argument("--w", type=int, default=1920); ap.add_argument("--h", type=int, default=1080); ap.add_argument("--frames", type=int, default=8)
a = ap.parse_args()
dump = tempfile.mktemp()
env = dict(os.environ, KVAZZUP_AMD_CTU_DUMP=dump)
r = subprocess.run([sys.executable, os.path.join(ROOT, "tools/measure/parse_rate.py"), "--w", str(a.w), "--h", str(a.h), "--frames", str(a.frames), "--threads", "1", "--reps", "1"], env=env, capture_output=True, text=True)
print(r.stdout.strip().splitlines()[-1])
t = collections.defaultdict(dict)
for l in open(dump):
    poc, cy, cx, ns = map(int, l.split()); t[poc][(cy, cx)] = ns / 1e3
os.unlink(dump)
for poc in sorted(t):
    g = t[poc]; R = max(r_ for r_, c in g) + 1; C = max(c for r_, c in g) + 1
    f = {}
    for r_ in range(R):
        for c in range(C):
            dep = f[(r_, c - 1)] if c else 0.0
            if r_: dep = max(dep, f[(r_ - 1, min(c + 1, C - 1))])
            f[(r_, c)] = dep + g[(r_, c)]
    tot = sum(g.values()); rows = [sum(g[(r_, c)] for c in range(C)) for r_ in range(R)]
    top = sorted(g.values(), reverse=True)
    print("picture %d: one thread %5.0f us; heaviest path %5.0f us = %.2fx at best; heaviest row %4.0f us; the 10 heaviest units of %d hold %2.0f %%" % (poc, tot, f[(R - 1, C - 1)], tot / f[(R - 1, C - 1)], max(rows), len(g), 100 * sum(top[:10]) / tot))
for th in (2, 4, 8):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools/measure/parse_rate.py"), "--w", str(a.w), "--h", str(a.h), "--frames", str(a.frames), "--threads", str(th), "--reps", "5"], capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1])
