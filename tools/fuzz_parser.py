#!/usr/bin/env python3
"""tools/fuzz_parser.py -- the decoder's host half (NAL units, parameter sets, slice headers, CABAC slice-data parser, merge / AMVP derivation) under
mutated input, through the parse-only hook: no device is touched, so it runs anywhere -- and under AddressSanitizer (host code only):

  make -C kvazzup_amd/csrc asan                     # scratch/abi/libkvazzup_amd_asan.so: decoder.hip + openhevc_api.hip with -Xarch_host -fsanitize=address
  ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so) \
    KVAZZUP_AMD_LIBRARY=$PWD/scratch/abi/libkvazzup_amd_asan.so python tools/fuzz_parser.py --trials 4000

Every libOpenHevcDecode call must RETURN (a picture count or an error code); a crash, a sanitizer report or a hang (--call-timeout) ends the run with the
seed of the trial, which reproduces it (--seed S --trials 1).  Mutations: bit flips, byte runs overwritten, truncation, insertion, NAL units dropped /
duplicated / swapped, parameter sets and slice headers hit as often as slice data."""
import argparse
import os
import random
import signal
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parser_probe as PP          # noqa: E402


def mutate(rng, nals):
    out = [bytearray(n) for n in nals]
    for _ in range(rng.choice((1, 1, 1, 2, 3, 6))):
        kind = rng.random()
        i = rng.randrange(len(out))
        n = out[i]
        hdr = 6                                                    # start code + NAL header
        if kind < 0.35 and len(n) > hdr:                           # bit flips, anywhere behind the NAL header (the first bytes twice as often)
            for _ in range(rng.choice((1, 1, 2, 4, 16))):
                p = rng.randrange(hdr, len(n)) if rng.random() < 0.5 else rng.randrange(hdr, min(len(n), hdr + 24))
                n[p] ^= 1 << rng.randrange(8)
        elif kind < 0.5 and len(n) > hdr + 1:                      # a run overwritten
            p = rng.randrange(hdr, len(n)); k = min(len(n) - p, rng.choice((1, 2, 4, 8, 64)))
            n[p:p + k] = bytes(rng.randrange(256) for _ in range(k)) if rng.random() < 0.7 else bytes([rng.choice((0, 0xff))]) * k
        elif kind < 0.65 and len(n) > hdr + 1:                     # truncation
            del n[rng.randrange(hdr, len(n)):]
        elif kind < 0.72:                                          # insertion
            p = rng.randrange(hdr, len(n) + 1); n[p:p] = bytes(rng.randrange(256) for _ in range(rng.choice((1, 2, 3, 8))))
        elif kind < 0.8 and len(out) > 1:                          # a NAL unit dropped
            del out[i]
        elif kind < 0.87:                                          # ... duplicated
            out.insert(i, bytearray(n))
        elif kind < 0.94 and len(out) > 1:                         # ... swapped with another
            j = rng.randrange(len(out)); out[i], out[j] = out[j], out[i]
        elif kind < 0.97:                                          # the NAL header itself (type, layer, temporal id)
            if len(n) > 5:
                n[4 + rng.randrange(2)] = rng.randrange(256)
        elif len(n) > hdr + 2:                                     # a run of zero BITS spliced into the head (parameter sets, slice headers): an Exp-Golomb field of any size, wherever it lands
            head = min(len(n), hdr + 40)
            bits = "".join("{:08b}".format(b) for b in n[hdr:head])
            at = rng.randrange(len(bits))
            bits = bits[:at] + "0" * rng.choice((8, 16, 24, 30, 31, 32, 33, 48)) + "1" + bits[at:]
            bits += "0" * (-len(bits) % 8)
            n[hdr:head] = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
    return [bytes(n) for n in out if len(n) > 4]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=1000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--threads", default="1,4")
    ap.add_argument("--call-timeout", type=int, default=60, help="seconds one trial may take before it counts as a hang")
    ap.add_argument("--cases", default="", help="comma-separated case names (default: every golden stream and every stream the checker's encoder writes)")
    a = ap.parse_args()
    cases = [(n, v) for n, v in PP.all_cases() if not a.cases or n in a.cases.split(",")]
    lib = PP._lib()
    errors = pictures = 0
    cur = [None]

    def on_alarm(*_):
        print("HANG: trial %s" % (cur[0],), flush=True)
        os._exit(3)
    signal.signal(signal.SIGALRM, on_alarm)
    for t in range(a.trials):
        seed = a.seed + t
        rng = random.Random(seed)
        name, nals = cases[rng.randrange(len(cases))]
        threads = int(rng.choice(a.threads.split(",")))
        mut = mutate(rng, nals)
        cur[0] = (seed, name, threads)
        signal.alarm(a.call_timeout)
        h = lib.libOpenHevcInit(1, 2)
        assert lib.kvzx_decoder_set_parse_only(h, threads) == 1 and lib.libOpenHevcStartDecoder(h) == 0
        for k, n in enumerate(mut):
            rc = lib.libOpenHevcDecode(h, n, len(n), k)
            errors += rc < 0
            pictures += rc > 0
        lib.libOpenHevcClose(h)
        signal.alarm(0)
        if (t + 1) % 500 == 0:
            print("%d trials: %d calls returned an error, none crashed" % (t + 1, errors), flush=True)
    print("done: %d trials from seed %d over %d streams; %d calls returned an error code; no crash, no hang" % (a.trials, a.seed, len(cases), errors))


if __name__ == "__main__":
    main()
