#!/usr/bin/env python3
"""tools/measure/parse_rate.py -- the decoder's HOST half alone (kvzx_decoder_set_parse_only: NAL units, headers, CABAC slice-data parser; no device
is touched, no picture comes out), timed on this machine's cores.  The stream is written by the CPU checker's encoder (tests/orc.py: test
infrastructure -- this is a measurement tool, not a product path) and cached under /tmp.

  python tools/measure/parse_rate.py [--w 1920 --h 1080 --frames 12 --qp 32 --kind 0 --period 64 --reps 5 --threads 1]

Prints pictures, bins (from the checker's encoder), parse time per picture and ns per bin, and the digest of everything the parser produced."""
import argparse
import ctypes as C
import os
import pickle
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def stream(w, h, frames, qp, kind, period, sao=0):
    key = "/tmp/parse_rate_%dx%d_%d_q%d_k%d_p%d_s%d.pkl" % (w, h, frames, qp, kind, period, sao)
    if os.path.exists(key):
        return pickle.load(open(key, "rb"))
    import orc
    oe = orc.OracleEncoder(w, h, qp=qp, period=period, me_range=16, sao=sao)
    aus, bins = [], []
    for t in range(frames):
        aus.append(oe.encode(orc.synth_frame(kind, 0x5EED0002, w, h, t)))
        bins.append(int(oe.debug()["bins"]))
    oe.close()
    nals = [orc.split_nals(a) for a in aus]
    pickle.dump((nals, bins), open(key, "wb"))
    return nals, bins


def parse_all(lib, nals, threads):
    h = lib.libOpenHevcInit(1, 2)
    assert lib.kvzx_decoder_set_parse_only(h, threads) == 1
    assert lib.libOpenHevcStartDecoder(h) == 0
    for t, au in enumerate(nals):
        for n in au:
            rc = lib.libOpenHevcDecode(h, n, len(n), t)
            assert rc >= 0, (t, rc, lib.kvzx_decoder_last_error(h))
    out, ms = (C.c_uint64 * 5)(), C.c_double()
    lib.kvzx_decoder_parse_probe_stats(h, out, C.byref(ms))
    lib.libOpenHevcClose(h)
    return list(out), ms.value


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--w", type=int, default=1920); ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=12); ap.add_argument("--qp", type=int, default=32)
    ap.add_argument("--kind", type=int, default=0); ap.add_argument("--period", type=int, default=64)
    ap.add_argument("--sao", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5); ap.add_argument("--threads", type=int, default=1)
    a = ap.parse_args()
    nals, bins = stream(a.w, a.h, a.frames, a.qp, a.kind, a.period, a.sao)
    from kvazzup_amd import _native
    lib = C.CDLL(_native.library_path())
    lib.libOpenHevcInit.restype = C.c_void_p
    lib.libOpenHevcInit.argtypes = [C.c_int, C.c_int]
    for f in (lib.libOpenHevcStartDecoder, lib.libOpenHevcClose, lib.kvzx_decoder_last_error):
        f.argtypes = [C.c_void_p]
    lib.libOpenHevcDecode.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int64]
    lib.kvzx_decoder_set_parse_only.argtypes = [C.c_void_p, C.c_int]
    lib.kvzx_decoder_parse_probe_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    best = None
    for _ in range(a.reps):
        st, ms = parse_all(lib, nals, a.threads)
        best = ms if best is None or ms < best else best
    nb = sum(bins)
    nb_p = sum(bins[1:]) if a.period > 1 else nb
    print("pictures %d  bins %d (P pictures: %.0f per picture)  tus %d  levels %d  digest %016x" % (st[0], nb, nb_p / max(1, len(bins) - 1), st[1], st[2], st[3]))
    print("parse: best of %d: %.3f ms total, %.3f ms per picture, %.2f ns per bin (threads %d)" % (a.reps, best, best / st[0], best * 1e6 / nb, a.threads))


if __name__ == "__main__":
    main()
