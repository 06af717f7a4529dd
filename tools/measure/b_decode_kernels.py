"""tools/measure/b_decode_kernels.py: the B-picture streams of b_decode_rate.py with the decoder's kernel timing on -- which kernel the general paths cost (GPU box)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import orc
from kvazzup_amd.codec import Decoder
w, h, n = 1920, 1080, 33
for name, kw in (("P", dict(b_slices=0)), ("B gop 8", dict(b_slices=80, gop=8))):
    g = orc.OracleGen(w, h, seed=21, intra_period=32, num_refs=4, tmvp=1, strong_intra=0, sign_hiding=1, wpp=1, tile_rows=1, intra_in_p=5, all_part_modes=0, amp=0, sao=1, qp_delta=0,
                      deblock_mode=0, th_depth_inter=1, th_depth_intra=1, max_cu_log2=6, min_cu_log2=3, nxn_intra=0, chroma_modes=0, transform_skip=0, cabac_init=0, chroma_qp_offsets=0,
                      par_mrg_level=2, big_mvd=0, uniform_tiles=1, density=10, slices=0, **kw)
    aus = [g.picture() for _ in range(n)]
    g.close()
    d = Decoder(threads=8, frame_threads=True, download=False)
    d.set_profiling(True)
    t0 = time.perf_counter(); got = 0
    for t, au in enumerate(aus):
        got += len(d.decode_au(au, t))
    got += len(d.drain())
    dt = time.perf_counter() - t0
    kt = d.kernel_times()
    d.close()
    print("%-10s %6.0f frames/s; per picture: " % (name, n / dt) + ", ".join("%s %.0f us" % (k, 1e3 * ms / max(1, c)) for k, (ms, c) in kt.items() if c))
