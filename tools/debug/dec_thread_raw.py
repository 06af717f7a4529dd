#!/usr/bin/env python3
"""tools/debug/dec_thread_raw.py <KVAZZUP_AMD_TIMELINE file> [first call] [calls]: the decoder filter thread's raw events for a few consecutive calls (us from the first shown)"""
import sys, collections
rows = [l.split() for l in open(sys.argv[1])]
dec_tid = collections.Counter(r[1] for r in rows if r[2] == "dec0").most_common(1)[0][0]
seq = sorted((int(r[0]), r[2], int(r[3])) for r in rows if r[1] == dec_tid)
starts = [i for i, e in enumerate(seq) if e[1] == "dec0"]
a = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
t0 = seq[starts[a]][0]
for e in seq[starts[a]:starts[a + n]]:
    print("%9.1f  %-10s %d" % ((e[0] - t0) / 1e3, e[1], e[2]))
