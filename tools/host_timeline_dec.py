#!/usr/bin/env python3
"""tools/host_timeline_dec.py <KVAZZUP_AMD_TIMELINE file>: what the decoder filter thread does between two NAL units (dec0 .. out2), medians."""
import sys, collections
rows = [l.split() for l in open(sys.argv[1])]
tags = ("dec0", "dlaunch0", "dlaunch1", "ddl", "dcomplete", "dec1", "out0", "alloc", "out1", "out2")
dec_tid = collections.Counter(r[1] for r in rows if r[2] == "dec0").most_common(1)[0][0]
seq = [(int(r[0]), r[2]) for r in rows if r[1] == dec_tid and r[2] in tags]
seq.sort()
d = collections.defaultdict(list)
for (ta, a), (tb, b) in zip(seq[:-1], seq[1:]):
    d[(a, b)].append((tb - ta) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("%-10s -> %-10s n %6d  median %7.1f us  p90 %7.1f  total %8.1f ms" % (k[0], k[1], len(v), v[len(v) // 2], v[len(v) * 9 // 10], sum(v) / 1e3))
