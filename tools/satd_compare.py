"""All-intra quality / rate of the intra mode search with SATD against SAD (intra-satd=1 / 0), 1080p, QP 32.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kvazzup_amd import synth
from kvazzup_amd.codec import Encoder

w, h = 1920, 1080
for kind in (synth.MOVING, 1, 2):
    for satd in (1, 0):
        e = Encoder(w, h, options=(("qp", 32), ("period", 1), ("intra-satd", str(satd))))
        bits, ps = 0, []
        for t in range(4):
            f = synth.frame(kind, 0x5EED0002, w, h, t)
            au, rec = e.encode(f)
            bits += len(au) * 8
            cw, ch = e.coded_size()
            y0 = np.asarray(f, dtype=np.uint8)[:w * h].reshape(h, w).astype(float)
            y1 = rec[:cw * ch].reshape(ch, cw)[:h, :w].astype(float)
            ps.append(10 * np.log10(255 ** 2 / np.mean((y0 - y1) ** 2)))
        print("clip %d  intra-satd=%d  bits/picture %.0f  psnr_y %.3f dB" % (kind, satd, bits / 4, float(np.mean(ps))))
        e.close()
