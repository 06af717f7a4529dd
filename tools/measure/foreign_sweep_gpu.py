"""one-off sweep: random synthesiser streams (every switch from the seed, scaling lists / transquant bypass / B slices / explicit weights / modified reference lists mixed in) through the HIP decoder against the
checker.  GPU box only: python tools/measure/foreign_sweep_gpu.py [first seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_foreign as T
first, count = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3000, 200)
bad = 0
sizes = [(416, 240), (352, 288), (200, 136), (648, 360), (64, 64), (24, 16), (1280, 720)]
for seed in range(first, first + count):
    w, h = sizes[seed % len(sizes)]
    kw = dict(seed=seed, threads=4 if seed % 3 == 0 else 1, frame_threads=seed % 3 == 0)
    if seed % 4 == 1: kw.update(scaling_lists=seed % 5, tq_bypass=(0, 25, 100)[seed % 3])
    if seed % 4 == 2: kw.update(b_slices=(40, 100)[seed % 2], gop=(0, 4, 8)[seed % 3])
    if seed % 4 == 3: kw.update(b_slices=(0, 50, 100)[seed % 3], gop=(0, 4, 8)[(seed // 3) % 3], weighted=(0, 40, 100)[(seed // 2) % 3], list_mod=(0, 60, 100)[(seed // 5) % 3], num_refs=4)
    try:
        T.run_stream(w, h, 8, **kw)
    except BaseException as e:      # noqa: BLE001
        bad += 1
        print(seed, "FAIL", repr(e)[:300], flush=True)
print("seeds", first, first + count, "bad", bad)
