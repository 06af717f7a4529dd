"""GPU: kvazaar hash=md5 / checksum (decoded picture hash SEI, H.265 D.2.19) -- the HIP encoder's access units, SEI included, equal the checker
encoder's; the checker's decoder and the HIP decoder (libOpenHevcSetCheckMD5) both find the hashes correct, and a flipped bit is noticed."""
import ctypes as C

import numpy as np
import pytest

import orc

SEED = 0x5EED0002


@pytest.mark.gpu
@pytest.mark.parametrize("hash_name,hash_val,w,h,owf", [("md5", 2, 416, 240, 0), ("checksum", 1, 320, 192, 0), ("md5", 2, 640, 368, 3)])
def test_hash_sei_matches_oracle_and_verifies(gpu, hash_name, hash_val, w, h, owf):
    from kvazzup_amd.codec import Decoder, Encoder
    n = 8
    oe = orc.OracleEncoder(w, h, qp=31, period=4, me_range=8)
    oe.set_option("hash", hash_val)
    ge = Encoder(w, h, options=(("qp", 31), ("period", 4), ("me-range", 8), ("owf", owf)), fields={"hash": hash_val})
    od = orc.OracleDecoder()
    gd = Decoder()
    gd.lib.libOpenHevcSetCheckMD5(gd.h, 1)
    frames = [orc.synth_frame(0, SEED, w, h, t) for t in range(n)]
    got = [ge.encode(f) for f in frames]
    for _ in range(owf):
        got.append(ge.encode(None))
    got = [g for g in got if g[0] is not None]
    assert len(got) == n
    for t in range(n):
        want = oe.encode(frames[t])
        assert got[t][0] == want, t
        assert got[t][0].rfind(b"\x00\x00\x00\x01\x50\x01\x84") > 0            # suffix SEI NAL unit, payload type 132, last in the access unit
        assert len(od.decode_au(got[t][0], t)) == 1 and len(gd.decode_au(got[t][0], t)) == 1
    assert od.hash_stats() == (n, 0)
    a, b = C.c_int(), C.c_int()
    gd.lib.kvzx_decoder_hash_stats(gd.h, C.byref(a), C.byref(b))
    assert (a.value, b.value) == (n, 0)
    # a damaged hash: the synchronous HIP decoder answers the SEI NAL unit with an error and counts it
    au = bytearray(got[n - 1][0]); au[au.rfind(b"\x00\x00\x00\x01\x50\x01\x84") + 12] ^= 1
    gd2 = Decoder(); gd2.lib.libOpenHevcSetCheckMD5(gd2.h, 1)
    for t in range(n - 1):
        gd2.decode_au(got[t][0], t)
    with pytest.raises(RuntimeError):
        gd2.decode_au(bytes(au), n - 1)
    gd2.lib.kvzx_decoder_hash_stats(gd2.h, C.byref(a), C.byref(b))
    assert (a.value, b.value) == (n, 1)
    for x in (ge, gd, gd2, oe, od):
        x.close()
