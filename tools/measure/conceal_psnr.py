#!/usr/bin/env python3
"""tools/measure/conceal_psnr.py: what a viewer gets behind lost access units -- luma PSNR of the concealed decode against the clean decode of the same stream (the
checker's encoder and decoder, CPU), with the stand-in a copy of the nearest kept reference picture (concealment v2) and with a grey one (ORC_CONCEAL_GREY=1: what
libavcodec's rule shows)."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import orc
    w, h, n, period = 416, 240, 49, 16
    e = orc.OracleEncoder(w, h, qp=30, period=period, me_range=16)
    aus = [e.encode(orc.synth_frame(0, 0x5EED0007, w, h, t)) for t in range(n)]
    e.close()
    def dec(keep):
        d = orc.OracleDecoder(); out = {}
        for t in keep:
            for f in d.decode_au(aus[t], t): out[f["pts"]] = f["i420"][:w * h].astype(np.float64)
        d.close(); return out
    clean = dec(range(n))
    lossy = dec([t for t in range(n) if t % period == 0 or t % 5 != 2])
    ps = [10 * np.log10(255 * 255 / max(np.mean((lossy[t] - clean[t]) ** 2), 1e-9)) for t in sorted(lossy) if not np.array_equal(lossy[t], clean[t])]
    print("%d of %d pictures shown differ from the clean decode: luma PSNR against it mean %.2f dB, worst %.2f dB" % (len(ps), len(lossy), np.mean(ps), np.min(ps)))
else:
    for name, env in (("copy of the nearest reference picture held (v2)", {}), ("grey (libavcodec's rule)", {"ORC_CONCEAL_GREY": "1"})):
        out = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, **env), capture_output=True, text=True).stdout.strip()
        print("%-50s %s" % (name, out))
