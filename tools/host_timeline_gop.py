#!/usr/bin/env python3
"""tools/host_timeline_gop.py <timeline file> [first picture] [count]: encoder and decoder host events of consecutive pictures side by side
(us from the first one shown) -- where a GOP's time goes around its intra picture."""
import sys, collections
ev = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    ns, tid, what, pic = line.split()
    ev[int(pic)].setdefault(what, int(ns))
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 180
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
cols = ["feed0", "enq", "sub1", "gpudone", "col1", "dec0", "dlaunch1", "dcomplete", "out1"]
t0 = min(v for p in range(lo, lo + n) for v in ev.get(p, {}).values()) if any(p in ev for p in range(lo, lo + n)) else 0
print("pic  " + "".join("%10s" % c for c in cols))
prev = None
for p in range(lo, lo + n):
    if p not in ev: continue
    print("%4d " % p + "".join("%10.0f" % ((ev[p][c] - t0) / 1e3) if c in ev[p] else "%10s" % "-" for c in cols) + ("  IDR" if p % 64 == 0 else ""))
