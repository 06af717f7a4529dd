#!/usr/bin/env python3
"""tools/host_timeline.py <file written with KVAZZUP_AMD_TIMELINE=<file>>: per-picture host-side stage times of the encoder
(feed0 -> copied -> enq -> sub0 -> sub1 -> gpudone -> arith -> bg1 -> col1) and the gaps between consecutive pictures at every stage."""
import sys, collections
ev = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    ns, tid, what, pic = line.split()
    ev[int(pic)].setdefault(what, int(ns))
pics = sorted(p for p in ev if all(k in ev[p] for k in ("enq", "sub0", "sub1", "gpudone", "arith", "bg1")))
lo = int(sys.argv[2]) if len(sys.argv) > 2 else len(pics) // 2
sel = pics[lo:lo + (int(sys.argv[3]) if len(sys.argv) > 3 else 24)]
stages = ["feed0", "copied", "enq", "sub0", "sub1", "gpudone", "arith", "bg1", "col1"]
t0 = min(ev[p][s] for p in sel for s in stages if s in ev[p])
print("pic   " + "".join("%9s" % s for s in stages) + "   (us from the first event shown)")
for p in sel:
    print("%5d " % p + "".join("%9.1f" % ((ev[p][s] - t0) / 1e3) if s in ev[p] else "%9s" % "-" for s in stages))
print("median interval between consecutive pictures at each stage (us):")
for s in stages:
    d = sorted((ev[b][s] - ev[a][s]) / 1e3 for a, b in zip(pics[:-1], pics[1:]) if s in ev[a] and s in ev[b] and b == a + 1)
    if d: print("  %-8s median %7.1f  p90 %7.1f  max %8.1f" % (s, d[len(d) // 2], d[len(d) * 9 // 10], d[-1]))
print("median latency between stages (us):")
for a, b in zip(stages[:-1], stages[1:]):
    d = sorted((ev[p][b] - ev[p][a]) / 1e3 for p in pics if a in ev[p] and b in ev[p])
    if d: print("  %-8s -> %-8s median %7.1f  p90 %7.1f" % (a, b, d[len(d) // 2], d[len(d) * 9 // 10]))
