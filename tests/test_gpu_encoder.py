"""GPU parity: the HIP encoder behind kvz_api (include/kvazaar.h) against the CPU checker (oracle/),
bit-exact on the access unit bytes and on the reconstructed I420, with stage-level diagnostics."""
import numpy as np
import pytest

import orc

SEED = 0x5EED0000


def _diagnose(dbg_o, dbg_g):
    msgs = []
    for k in ("cu_log2", "cu_intra", "cu_intra_mode", "cu_mv", "cu_cbf", "cu_flags"):
        a, b = dbg_o[k], dbg_g[k]
        if k == "cu_mv":
            m = (dbg_o["cu_intra"] == 0)
            a, b = a[m], b[m]
        if k == "cu_intra_mode":
            m = dbg_o["cu_intra"] == 1
            a, b = a[m], b[m]
        if not np.array_equal(a, b):
            bad = np.argwhere(np.asarray(a != b))
            msgs.append("%s differs at %d entries, first %s (oracle %s gpu %s)" % (k, len(bad), bad[0].tolist(), a[tuple(bad[0])] if a.ndim == bad.shape[1] else "?", b[tuple(bad[0])] if b.ndim == bad.shape[1] else "?"))
    for c in range(3):
        a, b = dbg_o["rec%d" % c], dbg_g["rec%d" % c]
        if not np.array_equal(a, b):
            bad = np.argwhere(a != b)
            msgs.append("rec%d differs at %d samples, first (y,x)=%s" % (c, len(bad), bad[0].tolist()))
    return "; ".join(msgs) if msgs else "no stage-level difference found (entropy coding / assembly?)"


def run_clip(w, h, frames, qp, period, me_range, kind, wpp=1, deblock=1, via_picture=True, tile_rows=1, sao=0, mv_frame=0, vaq=0, gpu_entropy=0, subme=0, me_early=1, slices=0):
    from kvazzup_amd.codec import Encoder
    oe = orc.OracleEncoder(w, h, qp=qp, period=period, me_range=me_range, wpp=wpp, deblock=deblock, tile_rows=tile_rows, sao=sao, mv_frame=mv_frame, vaq=vaq, subme=subme, me_early=me_early, slices=slices)
    ge = Encoder(w, h, options=(("qp", qp), ("period", period), ("me-range", me_range), ("wpp", wpp), ("deblock", deblock), ("tiles", "1x%d" % tile_rows),
                                ("sao", "full" if sao else "off"), ("mv-constraint", ("none", "frame", "frametilemargin")[mv_frame]), ("gpu-entropy", gpu_entropy), ("slices", ("none", "wpp", "tiles")[slices]), ("subme", subme), ("me-early-termination", "on" if me_early else "off")) + ((('vaq', vaq),) if vaq else ()))
    assert not ge.rejected, ge.rejected
    try:
        for t in range(frames):
            frame = orc.synth_frame(kind, SEED, w, h, t)
            au_o = oe.encode(frame)
            au_g, rec_g = ge.encode(frame)
            dbg_o = oe.debug()
            if au_o != au_g or not np.array_equal(rec_g, oe.recon()):
                pytest.fail("frame %d (%s): AU %d vs %d bytes, equal=%s, recon equal=%s: %s" % (
                    t, "I" if dbg_o["is_intra"] else "P", len(au_o), len(au_g), au_o == au_g,
                    np.array_equal(rec_g, oe.recon()), _diagnose(dbg_o, ge.debug_all())))
            assert ge.last_bins() == dbg_o["bins"]
    finally:
        ge.close()
        oe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(w=128, h=64, frames=2, qp=32, period=1, me_range=8, kind=0),       # smallest legal size, all intra
    dict(w=256, h=192, frames=5, qp=32, period=64, me_range=16, kind=0),    # I + P, moving objects
    dict(w=320, h=240, frames=4, qp=22, period=64, me_range=8, kind=2),     # noise: dense coefficients
    dict(w=320, h=240, frames=4, qp=40, period=2, me_range=8, kind=2),
    dict(w=192, h=128, frames=3, qp=10, period=64, me_range=8, kind=2),     # low QP: large levels, escape codes
    dict(w=192, h=128, frames=3, qp=32, period=64, me_range=8, kind=1),     # flat: everything skipped
    dict(w=416, h=240, frames=6, qp=32, period=4, me_range=32, kind=0, wpp=0),
    dict(w=640, h=360, frames=4, qp=27, period=64, me_range=16, kind=0, deblock=0),
    dict(w=702, h=394, frames=3, qp=51, period=64, me_range=4, kind=0),     # odd-ish size (even), extreme QP
    dict(w=130, h=70, frames=3, qp=0, period=64, me_range=1, kind=2),
    dict(w=16, h=16, frames=3, qp=30, period=64, me_range=8, kind=0),        # smallest input the ABI accepts (coded 128x64)
    dict(w=3840, h=2160, frames=2, qp=32, period=64, me_range=16, kind=0),  # BASELINE configs[2] size: one IDR + one P against the checker
])
def test_encoder_matches_oracle(gpu, cfg):
    run_clip(**cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(w=256, h=192, frames=5, qp=32, period=64, me_range=16, kind=0),                          # I + P, WPP
    dict(w=320, h=240, frames=3, qp=10, period=2, me_range=8, kind=2),                            # dense coefficients, escape codes, long 0xff runs
    dict(w=192, h=128, frames=3, qp=32, period=64, me_range=8, kind=1),                           # everything skipped: substreams of a few bytes
    dict(w=416, h=240, frames=4, qp=32, period=4, me_range=16, kind=0, wpp=0),                    # one substream for the whole picture
    dict(w=320, h=256, frames=4, qp=30, period=3, me_range=16, kind=0, tile_rows=2),              # tiles + WPP: fresh contexts at tile starts
    dict(w=320, h=256, frames=3, qp=27, period=2, me_range=8, kind=2, tile_rows=2, wpp=0),        # one substream per tile
    dict(w=320, h=256, frames=3, qp=30, period=64, me_range=8, kind=0, sao=1, vaq=8),             # SAO syntax and cu_qp_delta bins in the token stream
    dict(w=1920, h=1080, frames=3, qp=32, period=64, me_range=16, kind=0),                        # BASELINE configs[1] size
])
def test_gpu_arithmetic_coder_matches_oracle(gpu, cfg):
    """gpu-entropy=1: the arithmetic coder proper on the GPU (k_cabac_rows, one wave per substream) instead of the host
    thread pool: same access units, byte for byte, and the same bin counts."""
    run_clip(gpu_entropy=1, **cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(w=320, h=256, frames=5, qp=30, period=64, me_range=16, kind=0, subme=4),                               # moving objects, both steps with diagonals
    dict(w=320, h=256, frames=4, qp=30, period=64, me_range=8, kind=0, subme=1, me_early=0),                    # half-sample, horizontal / vertical only; every block searched
    dict(w=320, h=256, frames=4, qp=27, period=64, me_range=8, kind=2, subme=2, me_early=0),                    # noise: ties and near-ties
    dict(w=320, h=256, frames=4, qp=32, period=64, me_range=8, kind=0, subme=3, wpp=0),
    dict(w=320, h=256, frames=5, qp=30, period=3, me_range=16, kind=0, subme=4, tile_rows=4, me_early=0),       # one CTU row per tile: most vertical candidates dropped
    dict(w=416, h=240, frames=4, qp=30, period=64, me_range=32, kind=0, subme=4, mv_frame=1, me_early=0),       # frame constraint, vectors at the picture edge
    dict(w=416, h=240, frames=4, qp=30, period=64, me_range=32, kind=2, subme=4, mv_frame=2, me_early=0),
    dict(w=130, h=70, frames=3, qp=20, period=64, me_range=4, kind=0, subme=4, me_early=0),                     # windows across the padded edges
    dict(w=320, h=256, frames=3, qp=30, period=64, me_range=8, kind=0, subme=4, sao=1, vaq=8),
    dict(w=1920, h=1080, frames=3, qp=32, period=64, me_range=16, kind=0, subme=4),                             # BASELINE configs[1] size
])
def test_subme_matches_oracle(gpu, cfg):
    """subme 1..4: k_subpel (fractional-sample refinement, SATD on the matrix cores) and the encoder's fractional-sample motion
    compensation against subme_refine() of the checker: same vectors, same access units, same reconstruction"""
    run_clip(**cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(w=320, h=256, frames=5, qp=30, period=3, me_range=16, kind=0, slices=1),                              # slices=wpp: a dependent slice segment per CTU row
    dict(w=320, h=256, frames=4, qp=30, period=64, me_range=8, kind=0, slices=1, tile_rows=2, sao=1),
    dict(w=320, h=256, frames=4, qp=27, period=2, me_range=8, kind=2, slices=2, tile_rows=2),                  # slices=tiles: an independent slice per tile, WPP rows inside
    dict(w=320, h=256, frames=4, qp=27, period=64, me_range=8, kind=0, slices=2, tile_rows=4, wpp=0, vaq=6),   # one substream per slice
    dict(w=320, h=256, frames=3, qp=30, period=64, me_range=8, kind=0, slices=1, gpu_entropy=1),               # the GPU arithmetic coder closes the rows the same way
    dict(w=1920, h=1080, frames=3, qp=32, period=64, me_range=16, kind=0, slices=1),                           # 17 NAL units per picture
])
def test_slices_option_matches_oracle(gpu, cfg):
    """uvgComm video/Slices (kvazaarfilter.cpp:205-215 -> kvazaar slices=wpp / tiles): one NAL unit per slice segment, the checker's
    access units byte for byte; the HIP decoder puts the pictures together again"""
    from kvazzup_amd.codec import Decoder, Encoder
    run_clip(**cfg)
    w, h = cfg["w"], cfg["h"]
    ge = Encoder(w, h, options=(("qp", cfg["qp"]), ("period", cfg["period"]), ("me-range", cfg["me_range"]), ("wpp", cfg.get("wpp", 1)), ("tiles", "1x%d" % cfg.get("tile_rows", 1)),
                                ("slices", ("none", "wpp", "tiles")[cfg["slices"]])))
    gd = Decoder()
    for t in range(3):
        au, rec = ge.encode(orc.synth_frame(cfg["kind"], SEED, w, h, t))
        assert au.count(b"\x00\x00\x00\x01") >= 2
        got = gd.decode_au(au, t)
        assert len(got) == 1 and np.array_equal(got[0]["i420"], rec), t
    ge.close(); gd.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(w=448, h=320, dims="2x2", cols=2, rows=2, frames=5, qp=30, period=4, kind=0),                              # uvgComm's first tile dimension default
    dict(w=448, h=320, dims="3x2", cols=3, rows=2, frames=4, qp=30, period=64, kind=0, wpp=0),                      # one substream per tile
    dict(w=448, h=320, dims="2x2", cols=2, rows=2, frames=4, qp=28, period=2, kind=2, slices=2, sao=1, vaq=6),      # a slice per tile, SAO merge and the QP chain stop at tile borders
    dict(w=448, h=320, dims="4x1", cols=4, rows=1, frames=4, qp=30, period=64, kind=0, subme=4, me_early=0),        # vectors confined in x: fractional candidates near the column borders
    dict(w=448, h=320, dims="16x16", cols=7, rows=5, frames=3, qp=32, period=2, kind=0),                            # finer than the CTU grid: one tile per CTU (7 x 5)
    dict(w=448, h=320, dims="2x3", cols=2, rows=3, frames=4, qp=30, period=1, kind=0, wpp=0, slices=2),             # all intra: the wavefront stops at tile borders
    dict(w=1920, h=1080, dims="4x4", cols=4, rows=4, frames=3, qp=32, period=64, kind=0),                           # BASELINE configs[1] size
    dict(w=640, h=384, dims="2x2", cols=2, rows=2, frames=24, qp=32, period=16, kind=0, bitrate=600000),            # rate control v2 over a tile grid
    dict(w=1920, h=1080, dims="16x16", cols=16, rows=16, frames=3, qp=32, period=64, kind=0),                       # uvgComm's largest tile dimension default at 1080p: a real 16 x 16 grid (30 x 17 CTUs), 256 tiles
    dict(w=3840, h=2160, dims="20x22", cols=20, rows=22, frames=2, qp=32, period=64, kind=0),                       # the level limit (A.4.2) at 4K: 440 tiles, more than a byte of tile ids
])
def test_tile_grid_matches_oracle(gpu, cfg):
    """kvazaar tiles=CxR with columns (uvgComm video/Tiles + video/tileDimensions, kvazaarfilter.cpp:196-202; the defaults "2x2" .. "16x16" all
    have columns): CTUs coded in tile-scan order, one substream per tile (per CTU row of a tile with WPP), contexts / WPP hand-over / availability /
    motion vectors / the QP chain per tile -- the checker's access units byte for byte, and the HIP decoder decodes them"""
    from kvazzup_amd.codec import Decoder, Encoder
    w, h = cfg["w"], cfg["h"]
    kw = dict(wpp=cfg.get("wpp", 1), sao=cfg.get("sao", 0), vaq=cfg.get("vaq", 0), subme=cfg.get("subme", 0), me_early=cfg.get("me_early", 1), slices=cfg.get("slices", 0))
    br = cfg.get("bitrate", 0)
    oe = orc.OracleEncoder(w, h, qp=cfg["qp"], period=cfg["period"], me_range=8, tile_rows=cfg["rows"], tile_cols=cfg["cols"], bitrate=br, rc_bands=4 if br else 0, **kw)
    opts = (("qp", cfg["qp"]), ("period", cfg["period"]), ("me-range", 8), ("tiles", cfg["dims"]), ("wpp", kw["wpp"]), ("sao", "full" if kw["sao"] else "off"), ("subme", kw["subme"]),
            ("me-early-termination", "on" if kw["me_early"] else "off"), ("slices", ("none", "wpp", "tiles")[kw["slices"]])) + ((("vaq", kw["vaq"]),) if kw["vaq"] else ()) + \
           ((("bitrate", br), ("rc-algorithm", "lambda")) if br else ())
    ge = Encoder(w, h, options=opts, fields={"target_bitrate": br})
    assert not ge.rejected, ge.rejected
    gd = Decoder()
    for t in range(cfg["frames"]):
        frame = orc.synth_frame(cfg["kind"], SEED, w, h, t)
        au, rec = ge.encode(frame)
        want = oe.encode(frame)
        assert au == want, (t, len(au), len(want), _diagnose(oe.debug(), ge.debug_all()))
        assert np.array_equal(rec, oe.recon()), t
        got = gd.decode_au(au, t)
        assert len(got) == 1 and np.array_equal(got[0]["i420"], rec), t
    ge.close(); gd.close(); oe.close()


@pytest.mark.gpu
def test_gpu_arithmetic_coder_pipelined(gpu):
    """gpu-entropy=1 with owf 6: six pictures in flight, every slot's coder on its own stream; output lags six pictures."""
    from kvazzup_amd.codec import Encoder
    w, h, frames, owf = 640, 384, 20, 6
    clip = [orc.synth_frame(0, SEED, w, h, t) for t in range(frames)]
    opts = (("qp", 30), ("period", 8), ("me-range", 8))
    e0 = Encoder(w, h, options=opts)
    want = [e0.encode(f) for f in clip]
    e0.close()
    e1 = Encoder(w, h, options=opts + (("owf", owf), ("gpu-entropy", 1)))
    got = [e1.encode(f) for f in clip]
    assert got[:owf] == [(None, None)] * owf
    for _ in range(owf):
        got.append(e1.encode(None))
    assert e1.encode(None) == (None, None)
    e1.close()
    for t in range(frames):
        assert got[t + owf][0] == want[t][0], t
        assert np.array_equal(got[t + owf][1], want[t][1]), t


@pytest.mark.gpu
@pytest.mark.parametrize("owf,period", [(1, 4), (2, 4), (3, 4), (4, 4), (8, 4), (12, 4), (16, 4), (16, 1), (6, 2), (20, 3)])
def test_owf_lags_output_and_flushes(gpu, owf, period):
    """video/OWF = n (kvazaarfilter.cpp:193): the access unit returned by call t is picture t - n (n >= 2: the
    host coding stage runs on background threads; n = 3 .. 16 keep that many pictures in flight, more are taken as 16), NULL pictures flush the rest; the bytes and
    reconstructions are those of the synchronous encoder.  With a short intra period several intra pictures are in flight at once, each started on the input
    stream beside the P pictures in front of it (period 1: nothing but)."""
    from kvazzup_amd.codec import Encoder
    w, h, frames = 320, 192, 7 + min(owf, 16) + (12 if owf >= 12 else 0)
    clip = [orc.synth_frame(0, SEED, w, h, t) for t in range(frames)]
    opts = (("qp", 30), ("period", period), ("me-range", 8))
    e0 = Encoder(w, h, options=opts)
    want = [e0.encode(f) for f in clip]
    e0.close()
    e1 = Encoder(w, h, options=opts + (("owf", owf),))
    owf = min(owf, 16)
    got = [e1.encode(f) for f in clip]
    assert got[:owf] == [(None, None)] * owf
    for _ in range(owf):
        got.append(e1.encode(None))
    assert e1.encode(None) == (None, None)
    e1.close()
    for t in range(frames):
        assert got[t + owf][0] == want[t][0], t
        assert np.array_equal(got[t + owf][1], want[t][1]), t


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,bitrate,owf,opts", [(320, 192, 200000, 0, {}), (640, 384, 900000, 2, {}), (640, 384, 400000, 1, dict(sao=1, subme=2)), (448, 320, 600000, 2, dict(vaq=6, tile_rows=2)),
                                                  (1920, 1080, 3000000, 2, {}),
                                                  (640, 384, 900000, 6, {}), (640, 384, 500000, 4, dict(sao=1, subme=2)), (1920, 1080, 3000000, 6, dict(sao=1, subme=2)),
                                                  (3840, 2160, 12000000, 2, {})])     # (4K: a group's workgroups alone are more than the GPU holds at once); owf > 2: the delay follows the pictures in flight (rc-delay)
def test_rate_control_v2_matches_the_checker(gpu, w, h, bitrate, owf, opts):
    """rc-algorithm lambda (what uvgComm sets with its bitrate, kvazaarfilter.cpp:223-228): "uvgx rate control v2" -- the picture-level
    controller plus feedback inside the picture: a P picture's CTU rows are reconstructed in four groups and the QP of the next group is
    decided ON THE DEVICE from the levels of the groups before (inside the one launch of k_inter_recon: rc_group_done), travelling as cu_qp_delta.  Access units identical to the
    checker's (rc_band_decide() in oracle/hevc_enc.c), picture for picture; the streams decode; the rate lands near the target."""
    from kvazzup_amd.codec import Decoder, Encoder
    frames = 12 if w >= 3840 else (24 if w >= 1920 else 48)
    clip = [orc.synth_frame(0, SEED, w, h, t) for t in range(frames)]
    oe = orc.OracleEncoder(w, h, qp=32, period=16, me_range=8, bitrate=bitrate, rc_bands=4, **opts)
    if owf > 2:
        oe.set_option("rc-delay", min(owf, 6) + 1)        # the controller books picture t - (pictures in flight + 1): the synchronous checker with the same delay decides alike
    want = [oe.encode(f) for f in clip]
    o = (("qp", 32), ("period", 16), ("me-range", 8), ("owf", owf), ("bitrate", bitrate), ("rc-algorithm", "lambda"), ("sao", "full" if opts.get("sao") else "off"),
         ("subme", opts.get("subme", 0)), ("tiles", "1x%d" % opts.get("tile_rows", 1))) + ((("vaq", opts["vaq"]),) if opts.get("vaq") else ())
    ge = Encoder(w, h, options=o, fields={"target_bitrate": bitrate})
    assert not ge.rejected, ge.rejected
    got = [ge.encode(f, want_recon=False)[0] for f in clip]
    for _ in range(owf):
        got.append(ge.encode(None, want_recon=False)[0])
    got = got[owf:]
    for t in range(frames):
        assert got[t] == want[t], (t, len(got[t]), len(want[t]))
    gd = Decoder()
    for t, au in enumerate(got):
        assert len(gd.decode_au(au, t)) == 1
    kbps = sum(len(a) for a in got) * 8 * 30 / frames / 1000
    assert frames < 24 or 0.6 * bitrate / 1000 < kbps < 1.5 * bitrate / 1000, kbps        # (a dozen pictures are mostly their intra picture)
    ge.close(); gd.close(); oe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("bitrate,owf", [(200000, 0), (1000000, 2)])
def test_rate_control_matches_the_checker_picture_for_picture(gpu, bitrate, owf):
    """video/bitrate != 0 (kvazaarfilter.cpp:223-228): picture-level rate control.  Its decisions depend on the sizes of
    earlier access units, with a fixed three-picture delay, so the pipelined encoder (owf 2) and the checker agree on
    every QP; one differing byte anywhere would make the streams diverge."""
    from kvazzup_amd.codec import Decoder, Encoder
    w, h, frames = 320, 192, 40
    clip = [orc.synth_frame(0, SEED, w, h, t) for t in range(frames)]
    oe = orc.OracleEncoder(w, h, qp=32, period=16, me_range=8, bitrate=bitrate)
    want = [oe.encode(f) for f in clip]
    ge = Encoder(w, h, options=(("qp", 32), ("period", 16), ("me-range", 8), ("owf", owf)), fields={"target_bitrate": bitrate})
    got = [ge.encode(f, want_recon=False)[0] for f in clip]
    for _ in range(owf):
        got.append(ge.encode(None, want_recon=False)[0])
    got = got[owf:]
    assert got == want
    qps = set()
    gd = Decoder()
    for t, au in enumerate(got):
        assert len(gd.decode_au(au, t)) == 1
    kbps = sum(len(a) for a in got) * 8 * 30 / frames / 1000
    assert 0.7 * bitrate / 1000 < kbps < 1.4 * bitrate / 1000, kbps
    ge.close(); gd.close(); oe.close()


@pytest.mark.gpu
def test_largest_size_8k_intra_and_inter_picture(gpu):
    """7680x4320 (BASELINE configs[4] size, 120 x 68 CTUs): one IDR and one P picture, access units and
    reconstruction bit-exact against the checker"""
    run_clip(w=7680, h=4320, frames=2, qp=35, period=64, me_range=8, kind=0)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(w=320, h=256, frames=5, qp=30, period=3, me_range=16, kind=0, tile_rows=2),            # tiles + WPP
    dict(w=320, h=256, frames=4, qp=30, period=64, me_range=16, kind=0, tile_rows=4),           # one CTU row per tile
    dict(w=320, h=256, frames=4, qp=27, period=2, me_range=8, kind=2, tile_rows=2, wpp=0),      # tiles without WPP: one substream per tile
    dict(w=256, h=448, frames=5, qp=32, period=4, me_range=32, kind=0, tile_rows=3),            # uneven rows (2, 2, 3), long vectors against the constraint
    dict(w=1920, h=1080, frames=3, qp=32, period=64, me_range=16, kind=0, tile_rows=4),
])
def test_tile_rows_match_oracle(gpu, cfg):
    """kvazaar "tiles" 1xN (kvazaarfilter.cpp:196-202): full-width tile rows with uniform spacing, prediction and
    entropy contexts confined to the tile, motion vectors constrained to it, deblocking across the boundaries"""
    run_clip(**cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,wpp,tile_rows", [(320, 256, 1, 1), (320, 256, 0, 1), (448, 320, 1, 2), (1280, 720, 1, 1)])
def test_roi_delta_qp_map_matches_oracle(gpu, w, h, wpp, tile_rows):
    """kvz_picture.roi (kvazaarfilter.cpp:423-431) with set-qp-in-cu: every CTU quantised with its own QP, cu_qp_delta coded with
    the CTU's first residual, QpY prediction along the CTU row, deblocking with the averaged QpY -- the map changes, is clamped
    and is removed again during the clip"""
    import ctypes as C
    from kvazzup_amd.codec import Encoder
    rng = np.random.default_rng(3)
    oe = orc.OracleEncoder(w, h, qp=30, period=4, me_range=16, wpp=wpp, tile_rows=tile_rows, qp_in_cu=1)
    ge = Encoder(w, h, options=(("qp", 30), ("period", 4), ("me-range", 16), ("wpp", wpp), ("tiles", "1x%d" % tile_rows), ("set-qp-in-cu", 1)))
    assert not ge.rejected, ge.rejected
    maps = {1: (4, 3, rng.integers(-12, 13, 12)), 3: (7, 5, rng.integers(-30, 31, 35)), 5: (0, 0, None)}
    cur = None
    for t in range(7):
        if t in maps:
            rw, rh, m = maps[t]
            oe.set_roi(rw, rh, m)
            cur = None if not rw else (rw, rh, np.ascontiguousarray(m, dtype=np.int8))
        p = ge.pic.contents
        if cur:
            p.roi.width, p.roi.height = cur[0], cur[1]
            p.roi.roi_array = cur[2].ctypes.data_as(C.POINTER(C.c_int8))
        else:
            p.roi.width = p.roi.height = 0
            p.roi.roi_array = None
        frame = orc.synth_frame(0 if t < 6 else 2, 7, w, h, t)
        au_o = oe.encode(frame)
        au_g, rec_g = ge.encode(frame)
        assert au_g == au_o, "picture %d: %d vs %d bytes" % (t, len(au_g), len(au_o))
        assert np.array_equal(rec_g, oe.recon()), t
    ge.close(); oe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(w=320, h=256, frames=6, qp=32, period=4, me_range=16, kind=0, sao=1),
    dict(w=320, h=256, frames=4, qp=37, period=2, me_range=8, kind=2, sao=1, wpp=0),            # noise: band offsets, one substream
    dict(w=448, h=320, frames=5, qp=22, period=3, me_range=16, kind=0, sao=1, tile_rows=2),     # no merge-up across the tile boundary
    dict(w=130, h=70, frames=3, qp=27, period=64, me_range=8, kind=0, sao=1),                   # conformance window
    dict(w=640, h=360, frames=3, qp=30, period=64, me_range=16, kind=0, sao=1, deblock=0),      # SAO straight on the reconstruction
    dict(w=1920, h=1080, frames=3, qp=32, period=64, me_range=16, kind=0, sao=1),
])
def test_sao_matches_oracle(gpu, cfg):
    """kvazaar "sao" (off at the ultrafast preset, on for the slower ones): statistics, "uvgx SAO decision v1", the filter
    (H.265 8.7.3) and sao() of every CTU, bit-exact against the checker -- access units and filtered reconstruction"""
    run_clip(**cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("owf,bitrate", [(2, 600000), (4, 600000), (7, 600000), (7, 0), (8, 0), (12, 0)])
def test_sao_pipelined_at_every_depth_matches_oracle(gpu, owf, bitrate):
    """SAO with pictures in flight: owf 2..7 start a picture's tokenizer from the launcher thread once its chain is done (encoder.h tok_deferred_), deeper
    pipelines queue it behind the chain's event; intra pictures every 5 (the side stream), a bit rate (the row groups), 20 pictures so that every set and slot
    is used more than once (with a bit rate at most six pictures are in flight, without one as many as owf says) -- access units and reconstructions against
    the checker, in order"""
    from kvazzup_amd.codec import Encoder
    w, h, frames = 448, 320, 20
    oe = orc.OracleEncoder(w, h, qp=30, period=5, me_range=8, sao=1, subme=2, bitrate=bitrate)
    if bitrate and owf >= 3:
        oe.set_option("rc-delay", min(owf, 6) + 1)
    ge = Encoder(w, h, options=(("qp", 30), ("period", 5), ("me-range", 8), ("sao", "full"), ("subme", 2), ("owf", owf)), fields={"target_bitrate": bitrate})
    got = []
    for t in range(frames + owf):
        out = ge.encode(orc.synth_frame(0, SEED, w, h, t) if t < frames else None)
        if out[0] is not None:
            got.append(out)
    assert len(got) == frames
    for t in range(frames):
        want = oe.encode(orc.synth_frame(0, SEED, w, h, t))
        assert got[t][0] == want, (owf, t, len(got[t][0]), len(want))
        assert np.array_equal(got[t][1], oe.recon()), (owf, t)
    ge.close(); oe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("KVZ_SWEEP_SEEDS", "16")))))
def test_random_tool_combinations_match_oracle(gpu, seed):
    """a seeded sweep over tool combinations (size, QP, period, range, WPP, tile rows, SAO, deblocking, delta-QP map, output
    lag): encoder and decoder against the checker, every picture -- catches interactions the single-feature tests cannot"""
    import ctypes as C
    from kvazzup_amd.codec import Decoder, Encoder
    rng = np.random.default_rng(1000 + seed)
    w, h = int(rng.integers(8, 60)) * 8, int(rng.integers(8, 48)) * 8
    hc = (h + 63) // 64
    cfg = dict(qp=int(rng.integers(8, 46)), period=int(rng.choice([1, 2, 3, 5, 64])), me_range=int(rng.choice([1, 4, 8, 16, 32])),
               wpp=int(rng.integers(0, 2)), deblock=int(rng.integers(0, 2)), tile_rows=int(rng.integers(1, min(hc, 3) + 1)),
               sao=int(rng.integers(0, 2)), qp_in_cu=int(rng.integers(0, 2)), bitrate=int(rng.choice([0, 0, 0, 150000, 2000000])), mv_frame=int(rng.choice([0, 0, 1, 2])), vaq=int(rng.choice([0, 0, 3, 12])), me_early=int(rng.integers(0, 2)))
    owf = int(rng.choice([0, 1, 2, 3, 5]))
    kind = int(rng.choice([0, 2]))
    # (drawn after everything the earlier rounds' sweeps drew, so that their configurations stay what they were)
    extra = dict(subme=int(rng.choice([0, 0, 2, 4])), tile_cols=int(rng.choice([1, 1, 2])) if w >= 256 else 1)
    tools = dict(intra_in_p=int(rng.integers(0, 3)), rdoq=int(rng.integers(0, 2)), signhide=int(rng.integers(0, 2)))
    cfg.update(extra)
    lossless = int(rng.integers(0, 4) == 0)          # (round 4: a quarter of the seeds with uvgComm's "lossless" box on top of whatever else they drew)
    me_source = int(rng.integers(0, 2))               # (round 6, drawn last: half of the seeds with the search on the input picture -- k_me / k_intra_analyse<P> ahead on the input stream -- on top of whatever else they drew)
    frames = (9 if owf < 3 else 12) if cfg["bitrate"] else 5              # (the rate controller starts moving the QP behind its delay)
    oe = orc.OracleEncoder(w, h, **cfg)
    if cfg["bitrate"] and owf >= 3:
        oe.set_option("rc-delay", owf + 1)             # the feedback delay follows the pictures in flight (encoder.hip rc_delay_)
    oe.set_option("intra-in-p", tools["intra_in_p"]); oe.set_option("rdoq", tools["rdoq"]); oe.set_option("signhide", tools["signhide"])
    if lossless:
        oe.set_option("lossless", 1)
    oe.set_option("me-source", me_source)
    od = orc.OracleDecoder()
    ge = Encoder(w, h, options=(("qp", cfg["qp"]), ("period", cfg["period"]), ("me-range", cfg["me_range"]), ("wpp", cfg["wpp"]),
                                ("deblock", cfg["deblock"]), ("tiles", "%dx%d" % (cfg["tile_cols"], cfg["tile_rows"])), ("sao", "full" if cfg["sao"] else "off"),
                                ("subme", cfg["subme"]), ("intra-in-p", tools["intra_in_p"]), ("rdoq", tools["rdoq"]), ("signhide", tools["signhide"]),
                                ("set-qp-in-cu", cfg["qp_in_cu"]), ("owf", owf), ("me-source", me_source),
                                ("mv-constraint", ("none", "frame", "frametilemargin")[cfg["mv_frame"]])) + ((("vaq", cfg["vaq"]),) if cfg["vaq"] else ()) + (("me-early-termination", "on" if cfg["me_early"] else "off"),) + ((("lossless", 1),) if lossless else ()), fields={"target_bitrate": cfg["bitrate"]})
    assert not ge.rejected, (cfg, ge.rejected)
    gd = Decoder()
    roi = None
    if cfg["qp_in_cu"] or cfg["vaq"]:
        rw, rh = int(rng.integers(1, 6)), int(rng.integers(1, 5))
        roi = (rw, rh, np.ascontiguousarray(rng.integers(-14, 15, rw * rh), dtype=np.int8))
        oe.set_roi(*roi)
    want, got = [], []
    for t in range(frames + owf):
        if t < frames:
            fr = orc.synth_frame(kind, 77 + seed, w, h, t)
            want.append((oe.encode(fr), oe.recon()))
            p = ge.pic.contents
            if roi:
                p.roi.width, p.roi.height = roi[0], roi[1]
                p.roi.roi_array = roi[2].ctypes.data_as(C.POINTER(C.c_int8))
            out = ge.encode(fr)
        else:
            out = ge.encode(None)
        if out[0] is not None:
            got.append(out)
    assert len(got) == frames, (cfg, owf, len(got))
    for t in range(frames):
        assert got[t][0] == want[t][0], (cfg, tools, lossless, me_source, owf, t, len(got[t][0]), len(want[t][0]))
        assert np.array_equal(got[t][1], want[t][1]), (cfg, tools, owf, t)
        dec = gd.decode_au(got[t][0], t)
        ref = od.decode_au(want[t][0], t)
        assert len(dec) == 1 and len(ref) == 1 and np.array_equal(dec[0]["i420"], ref[0]["i420"]) and np.array_equal(dec[0]["i420"], want[t][1]), (cfg, t)
    ge.close(); gd.close(); oe.close(); od.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2])
def test_mv_constraint_frame_matches_oracle_and_keeps_blocks_inside(gpu, mode):
    """uvgComm's mv-constraint setting (kvazaarfilter.cpp:246-276) frame / frametile (1) and frametilemargin (2): candidates
    whose displaced block leaves the picture are not searched; long vectors at the picture edges are what it removes"""
    from kvazzup_amd.codec import Encoder
    w, h = 320, 192
    run_clip(w=w, h=h, frames=5, qp=30, period=64, me_range=32, kind=2, mv_frame=mode)
    run_clip(w=448, h=320, frames=4, qp=30, period=64, me_range=16, kind=0, mv_frame=mode, tile_rows=2)
    ge = Encoder(w, h, options=(("qp", 30), ("period", 64), ("me-range", 32), ("mv-constraint", ("frame", "frametilemargin")[mode - 1])))
    for t in range(3):
        ge.encode(orc.synth_frame(2, SEED, w, h, t))
    d = ge.debug_all()
    cw, ch = d["coded_w"], d["coded_h"]
    mv, l2 = d["cu_mv"].astype(int) // 4, d["cu_log2"]
    for by in range(ch // 8):
        for bx in range(cw // 8):
            n = 1 << int(l2[by, bx]); x0, y0 = (bx * 8) & ~31, (by * 8) & ~31       # the constraint is applied to the 32x32 search block
            assert 0 <= x0 + mv[by, bx, 0] and x0 + mv[by, bx, 0] + 32 <= cw and 0 <= y0 + mv[by, bx, 1] and y0 + mv[by, bx, 1] + 32 <= ch
    ge.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(w=448, h=320, frames=6, qp=30, period=4, me_range=16, kind=0, vaq=5),
    dict(w=320, h=256, frames=4, qp=36, period=64, me_range=8, kind=2, vaq=20, wpp=0),
    dict(w=640, h=384, frames=4, qp=26, period=3, me_range=16, kind=0, vaq=10, tile_rows=2, sao=1),
    dict(w=1920, h=1080, frames=3, qp=32, period=64, me_range=16, kind=0, vaq=8),
])
def test_vaq_matches_oracle(gpu, cfg):
    """uvgComm video/VAQ (kvazaar "vaq" 1..20, kvazaarfilter.cpp:280-284): per-CTU QP from the CTU's luma variance against the
    picture's average ("uvgx VAQ v1"), carried as cu_qp_delta; k_vaq_stats / k_vaq_apply against the checker"""
    run_clip(**cfg)


@pytest.mark.gpu
def test_reopened_instances_reuse_streams_and_code_identically(gpu):
    """uvgComm closes and re-opens its encoder and decoder on every settings change (kvazaarfilter.cpp:91-119): the library hands the HIP
    streams of a closed instance to the next one (csrc/stream_pool.h); access units and decoded pictures stay those of the first instance"""
    from kvazzup_amd.codec import Encoder, Decoder
    w, h, frames = 640, 384, 5
    clip = [orc.synth_frame(0, SEED, w, h, t) for t in range(frames)]
    first = None
    for cycle in range(4):
        e = Encoder(w, h, options=(("qp", 30), ("period", 4), ("me-range", 8), ("owf", 2)))
        d = Decoder()
        aus = []
        for f in clip + [None, None]:
            au, _ = e.encode(f)
            if au is not None:
                aus.append(au)
        pics = [p["i420"].copy() for au in aus for p in d.decode_au(au)]
        e.close(); d.close()
        assert len(aus) == frames and len(pics) == frames
        if first is None:
            first = (aus, pics)
        else:
            assert aus == first[0]
            assert all(np.array_equal(a, b) for a, b in zip(pics, first[1]))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(w=416, h=240, qp=30, period=4, rdoq=1, signhide=1, frames=8, kind=0),
    dict(w=416, h=240, qp=24, period=64, rdoq=0, signhide=1, frames=6, kind=0),                 # sign hiding alone, mostly P pictures
    dict(w=416, h=240, qp=34, period=1, rdoq=1, signhide=0, frames=3, kind=0),                  # the zero-out alone, all intra (horizontal / vertical scans in 8x8 blocks)
    dict(w=320, h=192, qp=22, period=2, rdoq=1, signhide=1, frames=4, kind=2),                  # noise: every coefficient group busy, big levels
    dict(w=640, h=368, qp=30, period=8, rdoq=1, signhide=1, frames=6, kind=0, subme=4, sao=1, tiles="2x2"),
    dict(w=1920, h=1080, qp=32, period=64, rdoq=1, signhide=1, frames=3, kind=0),               # BASELINE configs[1] size: 32x32 transforms on the matrix cores
])
def test_rdoq_and_sign_hiding_match_oracle(gpu, cfg):
    """kvazaar rdoq / signhide (row f4): "uvgx RDOQ v1" and sign data hiding in the quantiser epilogue of k_inter_recon / k_intra_recon, the hidden
    sign left out by the tokenizer, sign_data_hiding_enabled_flag in the PPS -- access units and reconstruction equal the checker's, and both
    decoders (which derive the hidden signs from the parities) return the reconstruction"""
    from kvazzup_amd.codec import Decoder, Encoder
    w, h = cfg["w"], cfg["h"]
    tiles = cfg.get("tiles", "1x1"); tc, tr = [int(v) for v in tiles.split("x")]
    oe = orc.OracleEncoder(w, h, qp=cfg["qp"], period=cfg["period"], me_range=8, subme=cfg.get("subme", 0), sao=cfg.get("sao", 0), tile_rows=tr, tile_cols=tc)
    oe.set_option("rdoq", cfg["rdoq"]); oe.set_option("signhide", cfg["signhide"])
    ge = Encoder(w, h, options=(("qp", cfg["qp"]), ("period", cfg["period"]), ("me-range", 8), ("rdoq", cfg["rdoq"]), ("signhide", cfg["signhide"]),
                                ("subme", cfg.get("subme", 0)), ("sao", "full" if cfg.get("sao") else "off")) + ((("tiles", tiles),) if tiles != "1x1" else ()))
    assert not ge.rejected, ge.rejected
    gd = Decoder(); od = orc.OracleDecoder()
    for t in range(cfg["frames"]):
        frame = orc.synth_frame(cfg["kind"], SEED, w, h, t)
        au, rec = ge.encode(frame)
        want = oe.encode(frame)
        assert au == want, (t, len(au), len(want), _diagnose(oe.debug(), ge.debug_all()))
        assert np.array_equal(rec, oe.recon()), t
        got = gd.decode_au(au, t); ref = od.decode_au(au, t)
        assert len(got) == 1 and np.array_equal(got[0]["i420"], rec), t
        assert len(ref) == 1 and np.array_equal(ref[0]["i420"], rec), t
    for x in (ge, gd, oe, od):
        x.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(w=416, h=240, qp=32, frames=8, cut=3, level=2),
    dict(w=416, h=240, qp=32, frames=6, cut=3),                    # level 1: 16x16 intra units only (the fast presets)
    dict(w=416, h=240, qp=27, frames=7, cut=2, subme=4, sao=1, rdoq=1, signhide=1, level=2),
    dict(w=640, h=368, qp=30, frames=6, cut=2, subme=2, tiles="2x2", level=2),
    dict(w=1920, h=1080, qp=32, frames=4, cut=2, level=2),          # BASELINE configs[1] size
    dict(w=1920, h=1080, qp=32, frames=3, cut=1),
    dict(w=640, h=384, qp=32, frames=12, cut=6, bitrate=600000, sao=1, subme=2),      # rate control v2 (uvgComm's default mode sets a bitrate): the row groups are priced without the intra units
])
def test_intra_units_in_p_pictures_match_oracle(gpu, cfg):
    """intra-in-p (row f4): a scene cut inside a GOP -- k_me's 16x16 costs, k_intra_analyse<P> decisions, k_intra_recon<.., P> behind k_inter_recon;
    access units and reconstruction equal the checker's, both decoders return the reconstruction, and the P picture at the cut does carry intra units"""
    from kvazzup_amd import synth
    from kvazzup_amd.codec import Decoder, Encoder
    w, h = cfg["w"], cfg["h"]
    tiles = cfg.get("tiles", "1x1"); tc, tr = [int(v) for v in tiles.split("x")]
    br = cfg.get("bitrate", 0)
    oe = orc.OracleEncoder(w, h, qp=cfg["qp"], period=64, me_range=8, subme=cfg.get("subme", 0), sao=cfg.get("sao", 0), tile_rows=tr, tile_cols=tc, bitrate=br, rc_bands=4 if br else 0)
    lvl = cfg.get("level", 1)
    oe.set_option("intra-in-p", lvl); oe.set_option("rdoq", cfg.get("rdoq", 0)); oe.set_option("signhide", cfg.get("signhide", 0))
    ge = Encoder(w, h, options=(("qp", cfg["qp"]), ("period", 64), ("me-range", 8), ("intra-in-p", lvl), ("rdoq", cfg.get("rdoq", 0)), ("signhide", cfg.get("signhide", 0)),
                                ("subme", cfg.get("subme", 0)), ("sao", "full" if cfg.get("sao") else "off")) + ((("tiles", tiles),) if tiles != "1x1" else ())
                 + ((("bitrate", br), ("rc-algorithm", "lambda")) if br else ()), fields={"target_bitrate": br})
    assert not ge.rejected, ge.rejected
    gd = Decoder(); od = orc.OracleDecoder()
    intra_units = []
    for t in range(cfg["frames"]):
        frame = synth.scene_cut_frame(SEED, w, h, t, cfg["cut"])
        au, rec = ge.encode(frame)
        want = oe.encode(frame)
        assert au == want, (t, len(au), len(want), _diagnose(oe.debug(), ge.debug_all()))
        assert np.array_equal(rec, oe.recon()), t
        got = gd.decode_au(au, t); ref = od.decode_au(au, t)
        assert len(got) == 1 and np.array_equal(got[0]["i420"], rec), t
        assert len(ref) == 1 and np.array_equal(ref[0]["i420"], rec), t
        intra_units.append(int(np.count_nonzero(ge.debug_all()["cu_intra"])))
    assert intra_units[cfg["cut"]] > 0 and intra_units[0] > 0, intra_units           # (picture 0 is the IDR picture)
    for x in (ge, gd, oe, od):
        x.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(w=416, h=240, qp=32, frames=5, period=4),
    dict(w=416, h=240, qp=27, frames=5, period=64, subme=4, sao=1, rdoq=1, signhide=1, intra_in_p=2, cut=3),
    dict(w=640, h=368, qp=30, frames=4, period=64, tiles="2x2", subme=2),
    dict(w=640, h=384, qp=32, frames=8, period=64, bitrate=500000, sao=1, subme=2, intra_in_p=1, cut=4),      # uvgComm's default mode + the scaling-list box
    dict(w=1920, h=1080, qp=32, frames=3, period=64),                # BASELINE configs[1] size
])
def test_scaling_list_default_matches_oracle(gpu, cfg):
    """`scaling-list default` (uvgComm's "scaling list" checkbox, kvazaarfilter.cpp:235-242): SPS scaling_list_enabled_flag with the default lists, per-position
    quantiser and dequantiser in k_inter_recon and k_intra_recon<.., SCAL>; access units and reconstruction equal the checker's, both decoders return it"""
    from kvazzup_amd import synth
    from kvazzup_amd.codec import Decoder, Encoder
    w, h = cfg["w"], cfg["h"]
    tiles = cfg.get("tiles", "1x1"); tc, tr = [int(v) for v in tiles.split("x")]
    br = cfg.get("bitrate", 0)
    oe = orc.OracleEncoder(w, h, qp=cfg["qp"], period=cfg["period"], me_range=8, subme=cfg.get("subme", 0), sao=cfg.get("sao", 0), tile_rows=tr, tile_cols=tc, bitrate=br, rc_bands=4 if br else 0)
    oe.set_option("scaling-list", 1); oe.set_option("intra-in-p", cfg.get("intra_in_p", 0)); oe.set_option("rdoq", cfg.get("rdoq", 0)); oe.set_option("signhide", cfg.get("signhide", 0))
    ge = Encoder(w, h, options=(("qp", cfg["qp"]), ("period", cfg["period"]), ("me-range", 8), ("scaling-list", "default"), ("intra-in-p", cfg.get("intra_in_p", 0)),
                                ("rdoq", cfg.get("rdoq", 0)), ("signhide", cfg.get("signhide", 0)), ("subme", cfg.get("subme", 0)), ("sao", "full" if cfg.get("sao") else "off"))
                 + ((("tiles", tiles),) if tiles != "1x1" else ()) + ((("bitrate", br), ("rc-algorithm", "lambda")) if br else ()), fields={"target_bitrate": br})
    assert not ge.rejected, ge.rejected
    gd = Decoder(); od = orc.OracleDecoder()
    for t in range(cfg["frames"]):
        frame = synth.scene_cut_frame(SEED, w, h, t, cfg["cut"]) if "cut" in cfg else orc.synth_frame(0, SEED, w, h, t)
        au, rec = ge.encode(frame)
        want = oe.encode(frame)
        assert au == want, (t, len(au), len(want), _diagnose(oe.debug(), ge.debug_all()))
        assert np.array_equal(rec, oe.recon()), t
        got = gd.decode_au(au, t); ref = od.decode_au(au, t)
        assert len(got) == 1 and np.array_equal(got[0]["i420"], rec), t
        assert len(ref) == 1 and np.array_equal(ref[0]["i420"], rec), t
    for x in (ge, gd, oe, od):
        x.close()


LOSSLESS_CASES = [
    dict(w=256, h=128, qp=32, frames=3, period=64, how="field"),                                        # uvgComm's way: kvz_config.lossless written directly
    dict(w=192, h=128, qp=27, frames=3, period=1, how="option"),                                        # all intra
    dict(w=416, h=240, qp=32, frames=6, period=64, subme=4, intra_in_p=2, cut=3, how="option"),         # quarter-sample vectors, intra units in P pictures, a scene cut
    dict(w=640, h=384, qp=37, frames=4, period=64, subme=2, sao=1, bitrate=500000, how="field"),        # the boxes that lossless switches off again (SAO, rate control)
    dict(w=1920, h=1088, qp=32, frames=3, period=64, how="field"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", LOSSLESS_CASES, ids=lambda c: "%dx%d_p%d_%s" % (c["w"], c["h"], c["period"], c["how"]))
def test_lossless_is_lossless_and_matches_the_checker(gpu, cfg):
    """uvgComm's "lossless" box (kvz_config.lossless, kvazaarfilter.cpp:244): every coding unit with cu_transquant_bypass_flag, the residual coded sample
    by sample (k_intra_recon<.., LL>, k_inter_recon<.., LL>).  The access units equal the checker's byte for byte, and what the three decoders make of them
    IS the source picture."""
    from kvazzup_amd import synth
    from kvazzup_amd.codec import Decoder, Encoder
    w, h = cfg["w"], cfg["h"]
    opts = [("qp", cfg["qp"]), ("period", cfg["period"]), ("me-range", 8), ("subme", cfg.get("subme", 0)), ("intra-in-p", cfg.get("intra_in_p", 0)),
            ("sao", cfg.get("sao", 0)), ("bitrate", cfg.get("bitrate", 0))]
    if cfg["how"] == "option":
        ge = Encoder(w, h, options=tuple(opts + [("lossless", 1)]))
    else:
        ge = Encoder(w, h, options=tuple(opts), fields={"lossless": 1})
    oe = orc.OracleEncoder(w, h, qp=cfg["qp"], period=cfg["period"], me_range=8, subme=cfg.get("subme", 0), sao=cfg.get("sao", 0), bitrate=cfg.get("bitrate", 0))
    oe.set_option("intra-in-p", cfg.get("intra_in_p", 0)); oe.set_option("lossless", 1)
    gd = Decoder(); od = orc.OracleDecoder()
    for t in range(cfg["frames"]):
        frame = synth.scene_cut_frame(SEED, w, h, t, cfg["cut"]) if "cut" in cfg else orc.synth_frame(0, SEED, w, h, t)
        au, rec = ge.encode(frame)
        want = oe.encode(frame)
        assert au == want, (t, len(au), len(want), _diagnose(oe.debug(), ge.debug_all()))
        assert np.array_equal(rec, frame), t
        got = gd.decode_au(au, t); ref = od.decode_au(au, t)
        assert len(got) == 1 and np.array_equal(got[0]["i420"], frame), t
        assert len(ref) == 1 and np.array_equal(ref[0]["i420"], frame), t
    for x in (ge, gd, oe, od):
        x.close()


@pytest.mark.gpu
def test_presets_switch_rdoq_and_sign_hiding_on(gpu):
    """preset medium and above: rdoq; slow and above: signhide too; superfast and above: intra units in P pictures (config_parse)"""
    from kvazzup_amd.codec import Encoder
    for preset, rd, sh, ip in (("ultrafast", 0, 0, 0), ("superfast", 0, 0, 1), ("fast", 0, 0, 1), ("medium", 1, 0, 2), ("slow", 1, 1, 2), ("placebo", 1, 1, 2)):
        e = Encoder(256, 128, options=(("preset", preset),))
        assert (e.cfg.contents.rdoq_enable, e.cfg.contents.signhide_enable, e.cfg.contents.intra_in_p) == (rd, sh, ip), preset
        e.close()
    # tools that do not exist are refused when switched on, accepted when switched off (kvazaarfilter.cpp:363-367 logs the refusal)
    e = Encoder(256, 128, options=(("amp", 1), ("smp", 0), ("bipred", 1), ("tmvp", 0), ("rd", 2), ("ref", 3), ("mv-rdo", 1), ("full-intra-search", 0)))
    assert sorted(e.rejected) == ["amp", "bipred", "mv-rdo", "rd", "ref"], e.rejected
    e.close()
