#!/usr/bin/env python3
"""tools/measure/soak_random_access.py [first seed] [last seed]: streams with CRA / RASL / RADL pictures, hidden pictures and (every other pair of seeds) temporal sub-layers on top of tests/test_gpu_everything.py's
draw of every other option, read whole, from their first CRA picture on, with that picture called BLA and with an end of sequence NAL unit before it (GPU box);
prints the seeds where the HIP decoder and the checker disagree."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import orc
from test_gpu_everything import drawn
from test_gpu_random_access import both
from test_random_access import EOS, discard_prior, rename, vcl_type
a, b = int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad, cuts = [], 0
for seed in range(a, b + 1):
    w, h, kw = drawn(seed)
    for k in ("long_term", "gop", "b_slices", "intra_period"):
        kw.pop(k, None)
    kw.update(gop=(2, 4, 8)[seed % 3], open_gop=1, b_slices=(0, 50)[seed & 1], intra_period=48, hidden_pics=(0, 0, 12)[seed % 3], tmvp=1, temporal_layers=(seed >> 1) & 1, rps_forms=(seed >> 2) & 1, vui_extras=(seed >> 3) & 1, hdr_extras=(seed >> 1) & (seed >> 2) & 1)
    g = orc.OracleGen(w, h, seed=seed, **kw)
    aus = [g.picture() for _ in range(20)]
    g.close()
    types = [vcl_type(x) for x in aus]
    cras = [i for i, t in enumerate(types) if t == 21]
    th = 1 + 2 * (seed % 3)
    try:
        both(aus, range(len(aus)), th, th > 1)
        if seed % 4 == 0:
            kw2 = dict(kw, intra_period=7 + seed % 5)
            g = orc.OracleGen(w, h, seed=seed, **kw2)
            short = [discard_prior(g.picture()) for _ in range(20)]      # (IDR pictures in the middle of groups, no_output_of_prior_pics_flag = 1)
            g.close()
            both(short, range(len(short)), th, th > 1)
        for k in cras[:1]:
            cuts += 1
            both(aus[k:], range(k, len(aus)), th, th > 1)
            both([rename(x, 21, 16) if i == k else x for i, x in enumerate(aus)], range(len(aus)), th, th > 1)
            both(aus[:k] + [EOS + aus[k]] + aus[k + 1:], range(len(aus)), th, th > 1)
            both([discard_prior(rename(x, 21, 16)) if i == k else x for i, x in enumerate(aus)], range(len(aus)), th, th > 1)
    except BaseException as e:      # (pytest.fail raises an outcome exception)
        bad.append(seed); print("seed", seed, (w, h), kw, cras, str(e)[:300], flush=True)
print("%d streams (%d with a CRA picture: cut, renamed, behind an end of sequence), %d differ %s" % (b - a + 1, cuts, len(bad), bad))
