#!/usr/bin/env python3
"""tools/measure/soak_everything.py [first seed] [last seed]: tests/test_gpu_everything.py's draw over many more seeds (GPU box); prints the seeds that differ."""
import os, sys, traceback
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_gpu_everything import drawn
from test_gpu_foreign import run_stream
a, b = int(sys.argv[1]) if len(sys.argv) > 1 else 49, int(sys.argv[2]) if len(sys.argv) > 2 else 400
bad = []
for seed in range(a, b + 1):
    w, h, kw = drawn(seed)
    try:
        run_stream(w, h, 12 if kw.get("long_term") else 6, seed=seed, threads=3 if seed & 1 else 1, frame_threads=bool(seed & 1), **kw)
    except BaseException as e:      # (pytest.fail raises an outcome exception)
        bad.append(seed); print("seed", seed, (w, h), kw, str(e)[:300], flush=True)
print("%d streams, %d differ %s" % (b - a + 1, len(bad), bad))
