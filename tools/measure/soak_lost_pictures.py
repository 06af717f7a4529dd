#!/usr/bin/env python3
"""tools/measure/soak_lost_pictures.py [first seed] [last seed]: access units lost from streams of tests/test_gpu_everything.py's draw (every option at once) -- the
HIP decoder and the checker must conceal alike (GPU box); prints the seeds where they disagree."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import random
import orc
from test_gpu_everything import drawn
from test_gpu_lost_pictures import run
from test_random_access import vcl_type
a, b = int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad, lost = [], 0
for seed in range(a, b + 1):
    w, h, kw = drawn(seed)
    kw.pop("intra_period", None)
    if not kw.get("long_term"):
        kw.update(gop=(0, 2, 4, 8)[seed % 4], b_slices=(0, 50)[seed & 1], open_gop=(seed >> 2) & 1, temporal_layers=(seed >> 3) & 1, rps_forms=(seed >> 1) & 1)
    g = orc.OracleGen(w, h, seed=seed, intra_period=16, tmvp=1, **kw)
    aus = [g.picture() for _ in range(24)]
    g.close()
    types = [vcl_type(x) for x in aus]
    r = random.Random(seed)
    lose = [i for i in range(1, len(aus)) if types[i] not in (19, 21) and r.random() < 0.15]
    if not lose:
        continue
    lost += len(lose)
    th = 1 + 2 * (seed % 3)
    try:
        run([(i, x) for i, x in enumerate(aus) if i not in lose], th, must_conceal=False)
    except BaseException as e:      # (pytest.fail raises an outcome exception)
        bad.append(seed); print("seed", seed, (w, h), kw, lose, repr(e)[:300], flush=True)
print("%d streams, %d access units lost, %d differ %s" % (b - a + 1, lost, len(bad), bad))
