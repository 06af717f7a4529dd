#!/usr/bin/env python3
"""tests/golden/make_streams.py -- writes the self-verifying Annex-B streams under tests/golden/streams/ (run on a GPU box:
`gpurun -- python tests/golden/make_streams.py gpurun_out/streams`, then copy the directory here).

Every stream of the HIP encoder carries a decoded picture hash SEI (hash=md5, H.265 D.2.19) behind each picture, so ANY conforming decoder
verifies it without this repository: `ffmpeg -i x.hevc -f null -` complains on a mismatch, tools/verify_external.sh also compares the
per-picture MD5 of the cropped output (`-f framemd5`) with index.json.  The synthesiser's streams (oracle/hevc_gen.c: tools this encoder
does not use) carry no SEI; index.json holds their pictures' MD5 as the checker decodes them."""
import hashlib, json, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import orc

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(R, "tests", "golden", "streams")
os.makedirs(out, exist_ok=True)
W, H, N, SEED = 416, 240, 10, 0x5EED0002
HIP = {   # name: kvz_api options (all with hash=md5)
    "hip_plain_qp32_p8": (("qp", 32), ("period", 8), ("me-range", 16)),
    "hip_tiles2x2_qp30": (("qp", 30), ("period", 8), ("me-range", 16), ("tiles", "2x2")),
    "hip_slices_wpp_qp32": (("qp", 32), ("period", 8), ("me-range", 16), ("slices", "wpp")),
    "hip_subme4_sao_qp28": (("qp", 28), ("period", 8), ("me-range", 16), ("subme", 4), ("sao", "full")),
    "hip_rc_400k": (("qp", 32), ("period", 8), ("me-range", 16), ("bitrate", 400000), ("rc-algorithm", "lambda")),
    "hip_vaq_roi_qp32": (("qp", 32), ("period", 8), ("me-range", 16), ("vaq", 8)),
    # a scene cut at picture 4 of the first GOP: intra units in P pictures, rdoq, sign data hiding (preset slow's tool set)
    "hip_scenecut_slow_qp30": (("qp", 30), ("period", 8), ("me-range", 16), ("preset", "slow")),
    # round 4: uvgComm's "scaling list" checkbox (kvazaarfilter.cpp:235-242): scaling_list_enabled_flag with the default lists
    "hip_scaling_list_default_qp30": (("qp", 30), ("period", 8), ("me-range", 16), ("scaling-list", "default"), ("subme", 2)),
    # ... and its "lossless" checkbox (:244): cu_transquant_bypass_flag everywhere; a smaller, shorter clip -- lossless pictures are large
    "hip_lossless": (("qp", 32), ("period", 8), ("me-range", 16), ("lossless", 1), ("subme", 2)),
}
SHAPE = {"hip_lossless": (192, 128, 4)}      # (width, height, pictures) where not W, H, N
index = {}
only = [a[len("--only="):] for a in sys.argv if a.startswith("--only=")]      # --only=<name>: just these streams of the HIP encoder (the others, and the synthesiser's, stay)
if (only or "--no-hip" in sys.argv) and os.path.exists(os.path.join(out, "index.json")):
    index = json.load(open(os.path.join(out, "index.json")))      # (CPU only: the HIP encoder's streams stay as they are)
if "--no-hip" not in sys.argv:
    from kvazzup_amd import synth
    from kvazzup_amd.codec import Encoder
    for name, opts in HIP.items():
        br = dict(opts).get("bitrate", 0)
        w_, h_, n_ = SHAPE.get(name, (W, H, N))
        if only and name not in only:
            continue
        e = Encoder(w_, h_, options=opts, fields={"hash": 2, "target_bitrate": br})
        assert not e.rejected, (name, e.rejected)
        od = orc.OracleDecoder()
        stream, md5s = b"", []
        for t in range(n_):
            src = synth.scene_cut_frame(SEED, w_, h_, t, 4) if "scenecut" in name else orc.synth_frame(0, SEED, w_, h_, t)
            au, rec = e.encode(src)
            assert not dict(opts).get("lossless") or np.array_equal(rec, src), (name, t)
            fr = od.decode_au(au, t)
            assert len(fr) == 1 and np.array_equal(fr[0]["i420"], rec), (name, t)      # the checker decodes it to the encoder's reconstruction
            stream += au; md5s.append(hashlib.md5(rec.tobytes()).hexdigest())
        assert od.hash_stats() == (n_, 0), (name, od.hash_stats())                      # ... and finds every hash SEI correct
        e.close(); od.close()
        open(os.path.join(out, name + ".hevc"), "wb").write(stream)
        index[name] = {"width": w_, "height": h_, "pictures": n_, "bytes": len(stream), "hash_sei": "md5", "source": "HIP encoder (kvz_api), options %s" % dict(opts),
                       "frame_md5": md5s}
# the synthesiser (CPU): a Kvazaar-shaped stream (lp-g4d3t1-like references, intra CUs in P pictures) and one with every tool switched on
GEN = {
    "gen_kvazaar_shaped": dict(seed=9, density=30, intra_period=8, num_refs=3, tmvp=0, amp=0, sao=0, sign_hiding=1, transform_skip=0, wpp=1, tile_rows=1, tile_cols=1,
                               qp_delta=0, deblock_mode=0, intra_in_p=10, all_part_modes=0, nxn_intra=1, max_cu_log2=5, min_cu_log2=3, slices=0, big_mvd=0),
    "gen_everything_on": dict(seed=23, density=30, intra_period=8, num_refs=4, tmvp=1, amp=1, sao=1, sign_hiding=1, transform_skip=1, cabac_init=1, wpp=1, tile_rows=1, tile_cols=1,
                              qp_delta=2, chroma_qp_offsets=1, deblock_mode=3, intra_in_p=20, all_part_modes=1, chroma_modes=1, nxn_intra=1, slices=0, big_mvd=0),
    # round 4: what a peer with uvgComm's "scaling list" / "lossless" boxes ticked sends (kvazaarfilter.cpp:235-244), and the general form of both tools
    "gen_scaling_list_default": dict(seed=31, density=30, intra_period=8, num_refs=3, tmvp=0, amp=0, sao=0, sign_hiding=1, transform_skip=0, wpp=1, tile_rows=1, tile_cols=1,
                                     qp_delta=0, deblock_mode=0, intra_in_p=10, all_part_modes=0, nxn_intra=1, max_cu_log2=5, min_cu_log2=3, slices=0, big_mvd=0, scaling_lists=1),
    "gen_scaling_lists_sps_pps": dict(seed=32, density=30, intra_period=8, num_refs=2, tmvp=1, sao=1, sign_hiding=1, transform_skip=1, wpp=1, tile_rows=1, tile_cols=1,
                                      qp_delta=2, intra_in_p=20, nxn_intra=1, th_depth_inter=2, th_depth_intra=2, max_cu_log2=6, slices=0, big_mvd=0, scaling_lists=4),
    "gen_lossless": dict(seed=33, density=30, intra_period=8, num_refs=3, tmvp=0, amp=0, sao=1, sign_hiding=1, transform_skip=0, wpp=1, tile_rows=1, tile_cols=1,
                         qp_delta=0, deblock_mode=0, intra_in_p=10, all_part_modes=0, nxn_intra=1, max_cu_log2=5, min_cu_log2=3, slices=0, big_mvd=0, tq_bypass=100),
    "gen_transquant_bypass_mixed": dict(seed=34, density=30, intra_period=8, num_refs=2, tmvp=1, sao=1, sign_hiding=1, transform_skip=1, wpp=1, tile_rows=1, tile_cols=1,
                                        qp_delta=2, deblock_mode=2, intra_in_p=25, nxn_intra=1, th_depth_inter=1, th_depth_intra=1, slices=0, big_mvd=0, tq_bypass=35),
    # round 4: B slices, what `bipred=1` / `gop=8` in a peer's custom-parameter list make Kvazaar write (kvazaarfilter.cpp:351-371).  frame_md5 is in OUTPUT order
    "gen_b_lowdelay": dict(seed=35, density=25, intra_period=8, num_refs=3, tmvp=1, amp=0, sao=0, sign_hiding=1, transform_skip=0, wpp=1, tile_rows=1, tile_cols=1,
                           qp_delta=0, deblock_mode=0, intra_in_p=10, all_part_modes=1, nxn_intra=1, max_cu_log2=5, min_cu_log2=3, slices=0, big_mvd=0, b_slices=80),
    # explicit weighted prediction (pred_weight_table()): what an x265 peer writes by default (weightp), in P and B slices
    "gen_weighted": dict(seed=37, density=25, intra_period=8, num_refs=3, tmvp=1, amp=0, sao=1, sign_hiding=1, transform_skip=0, wpp=1, tile_rows=1, tile_cols=1,
                         qp_delta=0, deblock_mode=0, intra_in_p=10, all_part_modes=1, nxn_intra=1, max_cu_log2=5, min_cu_log2=3, slices=0, big_mvd=0, b_slices=40, weighted=60),
    "gen_list_modification": dict(seed=38, density=25, intra_period=8, num_refs=4, tmvp=1, amp=0, sao=0, sign_hiding=1, transform_skip=0, wpp=1, tile_rows=1, tile_cols=1,
                                  qp_delta=0, deblock_mode=0, intra_in_p=10, all_part_modes=1, nxn_intra=1, max_cu_log2=5, min_cu_log2=3, slices=0, big_mvd=0, b_slices=40, list_mod=70),
    # round 6: coding tree blocks of 32 and 16 samples -- what encoders other than Kvazaar write (hardware encoders) --, tiles whose boundaries fall inside a 64x64
    # area, SAO, intra blocks in P pictures, temporal prediction
    "gen_ctb32": dict(seed=41, density=30, intra_period=8, num_refs=2, tmvp=1, amp=1, sao=1, sign_hiding=1, transform_skip=0, wpp=0, tile_rows=2, tile_cols=2,
                      qp_delta=2, deblock_mode=0, intra_in_p=20, all_part_modes=1, nxn_intra=1, max_cu_log2=5, min_cu_log2=3, slices=0, big_mvd=0, ctb_log2=5),
    "gen_ctb16": dict(seed=42, density=30, intra_period=8, num_refs=2, tmvp=1, amp=0, sao=1, sign_hiding=0, transform_skip=1, wpp=1, tile_rows=1, tile_cols=1,
                      qp_delta=1, deblock_mode=2, intra_in_p=20, all_part_modes=1, nxn_intra=1, max_cu_log2=4, min_cu_log2=3, slices=0, big_mvd=0, ctb_log2=4),
    # round 6: FREE slices -- slice segments that begin at any coding tree block, independent slices (own slice_qp_delta) and dependent segments mixed: what an encoder
    # that cuts its slices by bytes or block counts sends; without and with WPP (the second one with 32-sample coding tree blocks)
    "gen_free_slices": dict(seed=51, density=30, intra_period=8, num_refs=2, tmvp=1, amp=1, sao=1, sign_hiding=1, transform_skip=0, wpp=0, tile_rows=1, tile_cols=1,
                            qp_delta=2, deblock_mode=0, intra_in_p=25, all_part_modes=1, nxn_intra=1, max_cu_log2=5, min_cu_log2=3, slices=3, big_mvd=0),
    "gen_free_slices_wpp": dict(seed=53, density=30, intra_period=8, num_refs=2, tmvp=1, amp=0, sao=1, sign_hiding=0, transform_skip=1, wpp=1, tile_rows=1, tile_cols=1,
                                qp_delta=1, deblock_mode=2, intra_in_p=25, all_part_modes=1, nxn_intra=1, max_cu_log2=5, min_cu_log2=3, slices=3, big_mvd=0, ctb_log2=5),
    # round 6: a minimum coding block of 16 samples (MinCbLog2SizeY 4): inter units cut into four (PART_NxN), intra NxN with 8x8 prediction blocks
    "gen_min_cb16": dict(seed=57, density=30, intra_period=8, num_refs=2, tmvp=1, amp=1, sao=1, sign_hiding=1, transform_skip=0, wpp=1, tile_rows=1, tile_cols=1,
                         qp_delta=1, deblock_mode=0, intra_in_p=25, all_part_modes=1, nxn_intra=1, max_cu_log2=6, min_cu_log2=4, slices=0, big_mvd=0, min_cb_log2=4),
    # round 6: what a Kvazaar peer with uvgComm's tiles + slices=tiles settings sends as far as the loop filters go: loop_filter_across_tiles_enabled_flag = 0,
    # pps_loop_filter_across_slices_enabled_flag = 0 (Kvazaar filters its tiles one by one), a slice per tile; SAO and deblocking on
    "gen_closed_tiles": dict(seed=59, density=30, intra_period=8, num_refs=2, tmvp=1, amp=0, sao=1, sign_hiding=1, transform_skip=0, wpp=0, tile_rows=2, tile_cols=2,
                             qp_delta=0, deblock_mode=0, intra_in_p=20, all_part_modes=1, nxn_intra=1, max_cu_log2=6, min_cu_log2=3, slices=2, big_mvd=0, lf_across=2),
    # round 6: PCM coding units (raw samples at 1 .. 8 bits inside the arithmetic codeword, which ends in front of them and starts again behind them)
    "gen_pcm": dict(seed=61, density=30, intra_period=8, num_refs=2, tmvp=1, amp=0, sao=1, sign_hiding=1, transform_skip=0, wpp=1, tile_rows=1, tile_cols=1,
                    qp_delta=1, deblock_mode=0, intra_in_p=35, all_part_modes=1, nxn_intra=1, max_cu_log2=4, min_cu_log2=3, slices=0, big_mvd=0, pcm=5),
    # round 6: long-term reference pictures (the sequence's first picture kept as one: SPS candidates and explicit entries, LSBs only and with the MSB cycles)
    "gen_long_term": dict(seed=63, density=20, intra_period=16, num_refs=3, tmvp=1, amp=0, sao=1, sign_hiding=1, transform_skip=0, wpp=1, tile_rows=1, tile_cols=1,
                          qp_delta=0, deblock_mode=0, intra_in_p=10, all_part_modes=1, nxn_intra=0, max_cu_log2=5, min_cu_log2=3, slices=0, big_mvd=0, long_term=1),
    # round 6: constrained intra prediction (intra blocks among inter ones take no reference samples from them)
    "gen_cip": dict(seed=67, density=25, intra_period=8, num_refs=2, tmvp=1, amp=1, sao=1, sign_hiding=1, transform_skip=0, wpp=1, tile_rows=1, tile_cols=1,
                    qp_delta=0, deblock_mode=0, intra_in_p=45, all_part_modes=1, nxn_intra=1, max_cu_log2=6, min_cu_log2=3, slices=0, big_mvd=0, cip=1),
    # round 6: random access points that are no IDR pictures -- CRA pictures with RADL and RASL leading pictures, the parameter sets repeated there (a decoder may
    # start at any of them and must drop the RASL pictures; tests cut this stream there), and pictures with pic_output_flag = 0
    "gen_open_gop": dict(seed=74, density=20, intra_period=32, num_refs=3, tmvp=1, amp=1, sao=1, sign_hiding=1, transform_skip=0, wpp=1, tile_rows=1, tile_cols=1,
                         qp_delta=0, deblock_mode=0, intra_in_p=10, all_part_modes=1, nxn_intra=0, max_cu_log2=5, min_cu_log2=3, slices=0, big_mvd=0, b_slices=60, gop=4,
                         open_gop=1, hidden_pics=12),
    # round 6: temporal sub-layers (TemporalId 0 .. 3 over a group of eight, parameter sets with sub-layer syntax, sub-layer non-reference pictures): what Kvazaar's gop=8 sends
    "gen_sub_layers": dict(seed=81, density=20, intra_period=32, num_refs=3, tmvp=1, amp=1, sao=1, sign_hiding=1, transform_skip=0, wpp=1, tile_rows=1, tile_cols=1,
                           qp_delta=0, deblock_mode=0, intra_in_p=10, all_part_modes=1, nxn_intra=0, max_cu_log2=5, min_cu_log2=3, slices=0, big_mvd=0, b_slices=60, gop=8,
                           temporal_layers=1),
    # round 6: the reference picture sets written every way 7.3.7 allows (SPS candidates explicit and predicted; slices that name a candidate, predict from one, write
    # their own), the VUI's optional parts in front of the timing information
    "gen_rps_forms": dict(seed=85, density=15, intra_period=32, num_refs=3, tmvp=1, amp=0, sao=1, sign_hiding=1, transform_skip=0, wpp=1, tile_rows=1, tile_cols=1,
                          qp_delta=0, deblock_mode=0, intra_in_p=10, all_part_modes=1, nxn_intra=0, max_cu_log2=5, min_cu_log2=3, slices=0, big_mvd=0, b_slices=60, gop=4,
                          rps_forms=1, vui_extras=1),
    "gen_b_gop8": dict(seed=36, density=25, intra_period=16, num_refs=4, tmvp=1, amp=1, sao=1, sign_hiding=1, transform_skip=0, wpp=1, tile_rows=1, tile_cols=1,
                       qp_delta=0, deblock_mode=0, intra_in_p=10, all_part_modes=1, nxn_intra=1, max_cu_log2=5, min_cu_log2=3, slices=0, big_mvd=0, b_slices=70, gop=8),
}
for name, cfg in ({} if only else GEN).items():
    if "--new-only" in sys.argv and name in index and os.path.exists(os.path.join(out, name + ".hevc")):
        continue                                              # (--new-only: streams that exist stay byte for byte what they are)
    g = orc.OracleGen(W, H, **cfg)
    od = orc.OracleDecoder()
    stream, md5s = b"", []
    npic = 16 if cfg.get("open_gop") else 12 if cfg.get("gop") or cfg.get("long_term") else 6
    for t in range(npic):
        au = g.picture()
        stream += au
        for fr in od.decode_au(au, t):
            md5s.append(hashlib.md5(fr["i420"].tobytes()).hexdigest())
    for fr in od.flush():                                  # (pictures held back for reordering)
        md5s.append(hashlib.md5(fr["i420"].tobytes()).hexdigest())
    g.close(); od.close()
    assert len(md5s) == npic or (cfg.get("hidden_pics") and npic - 6 < len(md5s) < npic), (name, len(md5s))      # (hidden_pics: some pictures are not output)
    open(os.path.join(out, name + ".hevc"), "wb").write(stream)
    index[name] = {"width": W, "height": H, "pictures": len(md5s), "bytes": len(stream), "hash_sei": None, "source": "oracle/hevc_gen.c (stream synthesiser), %s" % cfg, "frame_md5": md5s}
json.dump(index, open(os.path.join(out, "index.json"), "w"), indent=1, sort_keys=True)
print({k: v["bytes"] for k, v in index.items()}, "total", sum(v["bytes"] for v in index.values()))
