"""The decoder's host half alone (include/kvazzup_amd.h kvzx_decoder_set_parse_only): helper shared by tests/test_parser_probe.py and
tests/golden/make_parser_digests.py.  No device is touched; nothing is decoded."""
import ctypes as C
import json
import os

import orc

HERE = os.path.dirname(os.path.abspath(__file__))
DIR = os.path.join(HERE, "golden", "streams")
SEED = 0x5EED0000

# streams the checker's encoder writes here: (name, kwargs of OracleEncoder, width, height, pictures, clip kind)
ENCODED = [
    ("enc_plain_320x240_qp32", dict(qp=32, period=64, me_range=16), 320, 240, 5, 0),
    ("enc_noise_qp22", dict(qp=22, period=64, me_range=8), 320, 240, 4, 2),
    ("enc_lowqp_escape_codes", dict(qp=6, period=64, me_range=8), 192, 128, 3, 2),
    ("enc_flat_all_skip", dict(qp=32, period=64, me_range=8), 192, 128, 3, 1),
    ("enc_no_wpp", dict(qp=32, period=4, me_range=16, wpp=0), 416, 240, 5, 0),
    ("enc_tiles_wpp", dict(qp=30, period=3, me_range=16, tile_rows=2), 320, 256, 4, 0),
    ("enc_tiles_2x2_slices", dict(qp=30, period=3, me_range=16, tile_rows=2, tile_cols=2, wpp=0, slices=2), 384, 256, 4, 0),
    ("enc_sao_vaq_subme", dict(qp=30, period=64, me_range=8, sao=1, vaq=8, subme=2), 320, 256, 4, 0),
    ("enc_slices_wpp", dict(qp=32, period=64, me_range=8, slices=1), 320, 256, 3, 0),
    ("enc_720p", dict(qp=32, period=64, me_range=16), 1280, 720, 3, 0),
]


def _lib():
    from kvazzup_amd import _native
    lib = C.CDLL(_native.library_path())
    lib.libOpenHevcInit.restype = C.c_void_p
    lib.libOpenHevcInit.argtypes = [C.c_int, C.c_int]
    for f in (lib.libOpenHevcStartDecoder, lib.libOpenHevcClose, lib.kvzx_decoder_last_error):
        f.argtypes = [C.c_void_p]
    lib.libOpenHevcDecode.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int64]
    lib.kvzx_decoder_set_parse_only.argtypes = [C.c_void_p, C.c_int]
    lib.kvzx_decoder_parse_probe_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    return lib


def probe(nals, threads):
    """parse the NAL units; {'pictures', 'tus', 'levels', 'digest'} of what the parser produced"""
    lib = _lib()
    h = lib.libOpenHevcInit(1, 2)
    assert lib.kvzx_decoder_set_parse_only(h, threads) == 1
    assert lib.libOpenHevcStartDecoder(h) == 0
    try:
        for t, n in enumerate(list(nals) + [bytes([0, 0, 0, 1, 36 << 1, 1])]):      # (end of sequence: a picture of free slices is complete when its access unit has ended)
            rc = lib.libOpenHevcDecode(h, n, len(n), t)
            assert rc == 0, "NAL %d: libOpenHevcDecode returned %d (error %d)" % (t, rc, lib.kvzx_decoder_last_error(h))
        out = (C.c_uint64 * 5)()
        lib.kvzx_decoder_parse_probe_stats(h, out, None)
    finally:
        lib.libOpenHevcClose(h)
    return {"pictures": int(out[0]), "tus": int(out[1]), "levels": int(out[2]), "digest": "%016x" % out[3]}


def golden_cases():
    index = json.load(open(os.path.join(DIR, "index.json")))
    for name in sorted(index):
        yield "golden_" + name, list(orc.split_nals(open(os.path.join(DIR, name + ".hevc"), "rb").read()))


def encoded_case(name):
    for n, kw, w, h, frames, kind in ENCODED:
        if n == name:
            oe = orc.OracleEncoder(w, h, **kw)
            nals = []
            for t in range(frames):
                nals += orc.split_nals(oe.encode(orc.synth_frame(kind, SEED, w, h, t)))
            oe.close()
            return nals
    raise KeyError(name)


def all_cases():
    yield from golden_cases()
    for n, *_ in ENCODED:
        yield n, encoded_case(n)
