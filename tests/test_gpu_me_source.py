"""GPU parity of "uvgx search pipelining v1" (kvazaar.h me-source; statement of record: build_refpad() in oracle/hevc_enc.c): the integer motion search of a P
picture looks at the previous INPUT picture, so k_me -- and k_intra_analyse<P> behind it -- of the pictures ahead run on the input stream beside the chain of
the picture in front.  Access units and reconstruction must be the checker's byte for byte, synchronously AND with pictures in flight (the search really runs
ahead then: per-working-set cost / candidate / progress arrays), with every tool that shares state with the search."""
import numpy as np
import pytest

import orc
from test_gpu_encoder import _diagnose

SEED = 0x5EED0000


def _pair(w, h, cfg, owf=0):
    from kvazzup_amd.codec import Encoder
    tiles = cfg.get("tiles", "1x1"); tc, tr = [int(v) for v in tiles.split("x")]
    br = cfg.get("bitrate", 0)
    oe = orc.OracleEncoder(w, h, qp=cfg.get("qp", 32), period=cfg.get("period", 64), me_range=cfg.get("me_range", 16), subme=cfg.get("subme", 0), sao=cfg.get("sao", 0),
                           tile_rows=tr, tile_cols=tc, bitrate=br, rc_bands=4 if br else 0, vaq=cfg.get("vaq", 0), me_early=cfg.get("me_early", 1), mv_frame=cfg.get("mv_frame", 0))
    oe.set_option("me-source", 1); oe.set_option("intra-in-p", cfg.get("intra_in_p", 0))
    if br and owf >= 3:
        oe.set_option("rc-delay", min(owf, 6) + 1)
    opts = (("qp", cfg.get("qp", 32)), ("period", cfg.get("period", 64)), ("me-range", cfg.get("me_range", 16)), ("me-source", 1), ("intra-in-p", cfg.get("intra_in_p", 0)),
            ("subme", cfg.get("subme", 0)), ("sao", "full" if cfg.get("sao") else "off"), ("owf", owf), ("me-early-termination", "on" if cfg.get("me_early", 1) else "off"),
            ("mv-constraint", ("none", "frame", "frametilemargin")[cfg.get("mv_frame", 0)]))
    opts += ((("tiles", tiles),) if tiles != "1x1" else ()) + ((("vaq", cfg["vaq"]),) if cfg.get("vaq") else ()) + ((("bitrate", br), ("rc-algorithm", "lambda")) if br else ())
    ge = Encoder(w, h, options=opts, fields={"target_bitrate": br})
    assert not ge.rejected, ge.rejected
    return oe, ge


def _frame(cfg, w, h, t):
    from kvazzup_amd import synth
    return synth.scene_cut_frame(SEED, w, h, t, cfg["cut"]) if cfg.get("cut") else orc.synth_frame(cfg.get("kind", 0), SEED, w, h, t)


CASES = [
    dict(w=256, h=192, frames=5),                                                            # the plain tool set: I + P, moving objects
    dict(w=416, h=240, frames=6, period=3, me_range=32),                                     # the picture behind an IDR searches the IDR's input picture
    dict(w=320, h=240, frames=4, qp=22, me_range=8, kind=2, me_early=0),                     # noise: every block searched, ties
    dict(w=192, h=128, frames=3, me_range=8, kind=1),                                        # flat: every block ends early (on the input pictures' difference)
    dict(w=130, h=70, frames=3, qp=20, me_range=4, me_early=0),                              # windows across the padded edges of the source planes
    dict(w=416, h=240, frames=6, subme=2, sao=1, intra_in_p=1, cut=3),                       # the fast presets' tool set with a scene cut: intra units priced on the input stream
    dict(w=416, h=240, frames=6, subme=4, intra_in_p=2, cut=2, qp=27),
    dict(w=640, h=368, frames=5, subme=2, tiles="2x2", intra_in_p=1, cut=2, qp=30),          # tile-constrained candidates
    dict(w=416, h=240, frames=4, me_range=32, subme=4, mv_frame=2, me_early=0, qp=30),
    dict(w=320, h=256, frames=4, sao=1, vaq=8, qp=30),                                       # per-CTU QPs: the head of the chain is a launch of its own
    dict(w=640, h=384, frames=10, bitrate=600000, sao=1, subme=2, intra_in_p=1, cut=5),      # uvgComm's default mode: rate control v2's state in picture order on the main stream
    dict(w=1920, h=1080, frames=3, subme=2, sao=1, intra_in_p=1),                            # BASELINE configs[1] size
]


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", CASES)
def test_search_on_the_input_picture_matches_oracle(gpu, cfg):
    from kvazzup_amd.codec import Decoder
    w, h = cfg["w"], cfg["h"]
    oe, ge = _pair(w, h, cfg)
    gd = Decoder()
    for t in range(cfg["frames"]):
        frame = _frame(cfg, w, h, t)
        au, rec = ge.encode(frame)
        want = oe.encode(frame)
        assert au == want, (t, len(au), len(want), _diagnose(oe.debug(), ge.debug_all()))
        assert np.array_equal(rec, oe.recon()), t
        got = gd.decode_au(au, t)
        assert len(got) == 1 and np.array_equal(got[0]["i420"], rec), t
    for x in (ge, gd, oe):
        x.close()


@pytest.mark.gpu
@pytest.mark.parametrize("owf", [2, 3, 6])
@pytest.mark.parametrize("cfg", [
    dict(w=640, h=384, frames=24, period=9, subme=2, sao=1, intra_in_p=1, cut=13),                         # searches of up to six pictures queued ahead of the chain, IDRs on the side stream
    dict(w=640, h=384, frames=24, period=64, bitrate=500000, subme=2, sao=1, intra_in_p=1, cut=7),         # ... under rate control v2
    dict(w=1280, h=720, frames=12, period=64),                                                             # the plain tool set
])
def test_search_running_ahead_of_the_chain_matches_oracle(gpu, owf, cfg):
    """pictures in flight: k_me / k_intra_analyse<P> of picture t + 1 .. t + owf are queued on the input stream while the main stream works on picture t -- the
    access units are still the synchronous checker's"""
    w, h, n = cfg["w"], cfg["h"], cfg["frames"]
    oe, ge = _pair(w, h, cfg, owf=owf)
    od = orc.OracleDecoder()
    frames = [_frame(cfg, w, h, t) for t in range(n)]
    aus = []
    for t in range(n + owf + 1):
        out = ge.encode(frames[t] if t < n else None, want_recon=False)
        if out[0]:
            aus.append(out[0])
    assert len(aus) == n, len(aus)
    for t in range(n):
        want = oe.encode(frames[t])
        assert aus[t] == want, (t, len(aus[t]), len(want))
        ref = od.decode_au(aus[t], t)
        assert len(ref) == 1 and np.array_equal(ref[0]["i420"], oe.recon()), t
    for x in (ge, oe, od):
        x.close()


@pytest.mark.gpu
def test_me_source_changes_the_search_and_presets_switch_it(gpu):
    """the option does something (vectors differ from the search on the reconstruction somewhere at QP 37) and the preset table sets it: superfast .. fast on, ultrafast and medium .. placebo off"""
    from kvazzup_amd.codec import Encoder
    for preset, on in (("ultrafast", 0), ("superfast", 1), ("veryfast", 1), ("faster", 1), ("fast", 1), ("medium", 0), ("placebo", 0)):
        e = Encoder(256, 128, options=(("preset", preset),))
        assert e.cfg.contents.me_source == on, preset
        e.close()
    w, h = 416, 240
    a = Encoder(w, h, options=(("qp", 37), ("me-source", 0))); b = Encoder(w, h, options=(("qp", 37), ("me-source", 1)))
    differ = False
    for t in range(4):
        f = orc.synth_frame(0, SEED, w, h, t)
        differ |= a.encode(f)[0] != b.encode(f)[0]
    assert differ
    a.close(); b.close()
