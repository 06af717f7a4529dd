"""Access units that never arrive (the streams come over RTP, /root/reference/src/media/delivery/uvgrtpreceiver.cpp:86-112 hands on what it got): a picture the
reference picture set says the current one predicts from is not there.  What a decoder does then is not the standard's business.  This project's rule
("concealment v2"; libavcodec's -- OpenHEVC's -- generate_missing_ref puts a grey picture there): a copy of the nearest reference picture the DPB holds, with the
missing picture order count, stands in, without motion, never output, and decoding goes on; the next IDR picture cleans up.  The checker (oracle/hevc_dec.c missing_ref), the Python decoder and the
product (csrc/decoder.hip conceal_ref; GPU side: tests/test_gpu_lost_pictures.py) follow it -- the pictures behind a loss are wrong, and the same wrong everywhere."""
import numpy as np
import pytest

import orc
from test_random_access import python_pictures, vcl_type


def lossy(seed, n=26, w=64, h=64, every=5, **extra):
    """(access units with some lost, how many were lost): never the first picture, never an IDR or CRA picture"""
    kw = dict(intra_period=12, num_refs=1 + seed % 4, tmvp=1)
    if seed & 1:
        kw.update(gop=(2, 4, 8)[seed % 3], b_slices=50)
    else:
        kw.update(long_term=(seed >> 1) & 1)
    kw.update(extra)
    g = orc.OracleGen(w, h, seed=seed, **kw)
    aus = [g.picture() for _ in range(n)]
    g.close()
    types = [vcl_type(a) for a in aus]
    lose = [i for i in range(1, n) if types[i] not in (19, 21) and i % every == 2]
    return [(i, a) for i, a in enumerate(aus) if i not in lose], aus, lose


@pytest.mark.parametrize("seed", range(1, 13))
def test_a_grey_picture_stands_in_for_a_lost_reference_picture(seed):
    import parser_probe as PP
    cut, aus, lose = lossy(seed)
    d = orc.OracleDecoder()
    got = []
    for t, a in cut:
        got += d.decode_au(a, t)
    got += d.flush()
    concealed = d.concealed()
    d.close()
    assert len(got) == len(cut) and 0 < concealed <= len(lose) * 4
    py = python_pictures([a for _, a in cut])
    assert len(py) == len(got)
    for a, b in zip(got, py):
        assert np.array_equal(a["i420"], b["i420"])
    # the pictures up to the first loss, and from the first IDR picture behind the last loss on, are the clean stream's
    d = orc.OracleDecoder()
    clean = {}
    for t, a in enumerate(aus):
        for f in d.decode_au(a, t):
            clean[f["pts"]] = f["i420"]
    for f in d.flush():
        clean[f["pts"]] = f["i420"]
    d.close()
    types = [vcl_type(a) for a in aus]
    heal = next((i for i in range(lose[-1], len(aus)) if types[i] == 19), len(aus))
    ok = [f for f in got if f["pts"] < lose[0] or f["pts"] >= heal]      # (time stamps count access units in decoding order)
    assert ok
    for f in ok:
        assert np.array_equal(f["i420"], clean[f["pts"]]), f["pts"]
    for threads in (0, 3):
        assert PP.probe([n for _, a in cut for n in orc.split_nals(a)], threads)["pictures"] == len(cut)

