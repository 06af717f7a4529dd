"""The committed bitstreams under tests/golden/streams/ (written by tests/golden/make_streams.py): Annex-B files a third party can check with
any HEVC decoder (tools/verify_external.sh).  Here: the checker decodes every one of them to the per-picture MD5 in index.json and finds every
decoded picture hash SEI (the HIP encoder's streams carry one per picture) correct -- on CPU; with a GPU, the HIP decoder does the same."""
import hashlib
import json
import os

import numpy as np
import pytest

import orc

DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "streams")
INDEX = json.load(open(os.path.join(DIR, "index.json")))


def access_units(stream):
    """split a stream into access units: a new one starts at a parameter set / at a slice segment with first_slice_segment_in_pic_flag that follows a VCL or
    suffix-SEI NAL unit"""
    nals = list(orc.split_nals(stream))
    aus, cur, seen_vcl = [], b"", False
    for n in nals:
        t = (n[4] >> 1) & 63
        first = t < 32 and (n[6] & 0x80) != 0
        if cur and ((t in (32, 33, 34, 35, 39) and seen_vcl) or (first and seen_vcl)):
            aus.append(cur); cur, seen_vcl = b"", False
        cur += n
        seen_vcl |= t < 32
    if cur:
        aus.append(cur)
    return aus


@pytest.mark.parametrize("name", sorted(INDEX))
def test_checker_decodes_golden_stream(name):
    meta = INDEX[name]
    stream = open(os.path.join(DIR, name + ".hevc"), "rb").read()
    assert len(stream) == meta["bytes"]
    od = orc.OracleDecoder()
    md5s = []
    for t, au in enumerate(access_units(stream)):
        for fr in od.decode_au(au, t):
            assert (fr["width"], fr["height"]) == (meta["width"], meta["height"])
            md5s.append(hashlib.md5(fr["i420"].tobytes()).hexdigest())
    for fr in od.flush():                                   # (a stream with reordering: the last pictures leave when it ends)
        md5s.append(hashlib.md5(fr["i420"].tobytes()).hexdigest())
    assert md5s == meta["frame_md5"]
    checked, bad = od.hash_stats()
    assert bad == 0 and checked == (meta["pictures"] if meta["hash_sei"] else 0)
    od.close()


def test_golden_streams_are_small_enough_to_live_in_the_repository():
    assert sum(m["bytes"] for m in INDEX.values()) <= 900 * 1024          # (thirty-one streams since round 6)
    assert sum(1 for m in INDEX.values() if m["hash_sei"] == "md5") >= 5


def test_a_corrupted_hash_is_noticed():
    name = sorted(n for n in INDEX if INDEX[n]["hash_sei"])[0]
    stream = bytearray(open(os.path.join(DIR, name + ".hevc"), "rb").read())
    i = stream.rfind(b"\x00\x00\x00\x01\x50\x01\x84")          # the last suffix SEI with payload type 132
    assert i > 0
    stream[i + 12] ^= 0x40                                         # one bit of the luma MD5
    od = orc.OracleDecoder()
    for t, au in enumerate(access_units(bytes(stream))):
        od.decode_au(au, t)
    checked, bad = od.hash_stats()
    assert checked == INDEX[name]["pictures"] and bad == 1
    od.close()


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("name", sorted(INDEX))
def test_hip_decoder_decodes_golden_stream(gpu, name, threads):
    """libOpenHevcSetCheckMD5(h, 1): the HIP decoder (synchronous and frame-threaded) compares the hash SEI messages with its own pictures"""
    import ctypes as C
    from kvazzup_amd.codec import Decoder
    meta = INDEX[name]
    stream = open(os.path.join(DIR, name + ".hevc"), "rb").read()
    gd = Decoder(threads=threads, frame_threads=threads > 1)
    gd.lib.libOpenHevcSetCheckMD5(gd.h, 1)
    md5s = []
    for t, au in enumerate(access_units(stream)):
        for fr in gd.decode_au(au, t):
            md5s.append(hashlib.md5(fr["i420"].tobytes()).hexdigest())
    for fr in gd.drain():
        md5s.append(hashlib.md5(fr["i420"].tobytes()).hexdigest())
    assert md5s == meta["frame_md5"]
    a, b = C.c_int(), C.c_int()
    gd.lib.kvzx_decoder_hash_stats(gd.h, C.byref(a), C.byref(b))
    assert b.value == 0 and a.value == (meta["pictures"] if meta["hash_sei"] else 0)
    gd.close()
