"""GPU: the HIP encoder reproduces the committed golden digests (tests/golden/oracle_streams.json, made by
tests/golden/make_golden.py from the CPU checker) and the HIP decoder turns those streams back into the
recorded reconstructions -- independent of a live checker build on the GPU box."""
import hashlib
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_golden_streams(gpu):
    from kvazzup_amd import synth
    from kvazzup_amd.codec import Decoder, Encoder
    with open(os.path.join(HERE, "golden", "oracle_streams.json")) as f:
        gold = json.load(f)
    for case in gold["cases"]:
        c = case["config"]
        ge = Encoder(c["w"], c["h"], options=(("qp", c["qp"]), ("period", c["period"]), ("me-range", c["me_range"]), ("wpp", c["wpp"]), ("deblock", c["deblock"]),
                                            ("tiles", "%dx%d" % (c.get("tile_cols", 1), c.get("tile_rows", 1))), ("sao", "full" if c.get("sao") else "off"), ("subme", c.get("subme", 0)),
                                            ("slices", ("none", "wpp", "tiles")[c.get("slices", 0)]))
                                            + ((("lossless", 1),) if c.get("lossless") else ()) + ((("scaling-list", "default"),) if c.get("scaling_list") else ()))
        gd = Decoder()
        for t, want in enumerate(case["frames"]):
            au, rec = ge.encode(synth.frame(c["kind"], c["seed"], c["w"], c["h"], t))
            assert len(au) == want["au_bytes"] and hashlib.md5(au).hexdigest() == want["au_md5"], (c, t)
            assert hashlib.md5(rec.tobytes()).hexdigest() == want["recon_md5"], (c, t)
            dec = gd.decode_au(au, t)
            assert len(dec) == 1 and hashlib.md5(dec[0]["i420"].tobytes()).hexdigest() == want["recon_md5"], (c, t)
        ge.close()
        gd.close()


@pytest.mark.gpu
def test_full_size_properties_1080p(gpu):
    """BASELINE size: properties that need no checker run -- decode(encode(x)) == reconstruction for every
    picture, the stream restarts cleanly at every IDR, flat content costs almost nothing."""
    import orc
    from kvazzup_amd import synth
    from kvazzup_amd.codec import Decoder, Encoder
    w, h = 1920, 1080
    ge = Encoder(w, h, options=(("qp", 32), ("period", 4), ("me-range", 16)))
    gd, gd2 = Decoder(), Decoder()
    sizes = []
    for t in range(6):
        fr = orc.synth_frame(0, 0x5EED0002, w, h, t)        # C twin of kvazzup_amd.synth (fast enough for 1080p)
        au, rec = ge.encode(fr)
        sizes.append(len(au))
        dec = gd.decode_au(au, t)
        assert len(dec) == 1 and np.array_equal(dec[0]["i420"], rec), t
        if t >= 4:                                   # a decoder joining at the second IDR gets the same pictures
            d2 = gd2.decode_au(au, t)
            assert len(d2) == 1 and np.array_equal(d2[0]["i420"], rec), t
        psnr = 10 * np.log10(255.0 ** 2 / max(1e-9, np.mean((fr[:w * h].astype(float) - rec[:w * h]) ** 2)))
        assert psnr > 30, (t, psnr)
    assert sizes[0] > 2 * sizes[1] and sizes[4] > 2 * sizes[5]      # inter pictures of this clip cost well under half an IDR
    ge.close(); gd.close(); gd2.close()
    fe = Encoder(w, h, options=(("qp", 32), ("period", 64)))
    flat = synth.frame(1, 0, w, h, 0)
    fe.encode(flat)
    au, rec = fe.encode(flat)
    assert len(au) < 400 and np.array_equal(rec, fe.encode(flat)[1])
    fe.close()
