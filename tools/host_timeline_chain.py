#!/usr/bin/env python3
"""tools/host_timeline_chain.py <timeline file>: what the host threads of the OTHER codec do while an intra picture's chain runs on the GPU --
the encoder's submit calls (duration, spacing) inside / outside the windows [decoder launched an intra picture, that picture complete], and the
decoder's launches inside / outside [encoder submitted an intra picture, its GPU work done]."""
import sys, collections
ev = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    ns, tid, what, pic = line.split()
    ev[int(pic)].setdefault(what, int(ns))
period = int(sys.argv[2]) if len(sys.argv) > 2 else 64
def windows(a, b):
    return [(ev[p][a], ev[p][b]) for p in sorted(ev) if p % period == 0 and a in ev[p] and b in ev[p]]
def inside(t, ws): return any(x <= t < y for x, y in ws)
def report(title, ws, start, end):
    pics = sorted(p for p in ev if start in ev[p] and end in ev[p])
    rows = {True: [], False: []}; gaps = {True: [], False: []}
    for a, b in zip(pics[:-1], pics[1:]):
        k = inside(ev[b][start], ws)
        rows[k].append((ev[b][end] - ev[b][start]) / 1e3)
        if b == a + 1: gaps[k].append((ev[b][start] - ev[a][start]) / 1e3)
    print(title, "(%d windows, mean %.0f us)" % (len(ws), sum(y - x for x, y in ws) / max(1, len(ws)) / 1e3))
    for k in (True, False):
        d, g = sorted(rows[k]), sorted(gaps[k])
        if d: print("   %-8s n %5d   %s->%s median %6.1f us p90 %6.1f   spacing median %6.1f us p90 %6.1f mean %6.1f" % ("inside" if k else "outside", len(d), start, end, d[len(d) // 2], d[len(d) * 9 // 10], g[len(g) // 2] if g else 0, g[len(g) * 9 // 10] if g else 0, sum(g) / max(1, len(g))))
report("encoder submit calls while the DECODER has an intra picture on the GPU", windows("dlaunch1", "dcomplete"), "sub0", "sub1")
report("decoder launches while the ENCODER has an intra picture on the GPU", windows("sub1", "gpudone"), "dlaunch0", "dlaunch1")
report("encoder pictures finishing on the GPU while the DECODER has an intra picture there", windows("dlaunch1", "dcomplete"), "sub1", "gpudone")
