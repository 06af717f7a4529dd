"""one-off sweep: random B streams (reordered groups, every other switch from the seed) through the HIP decoder against the checker.  GPU box only:
python tools/measure/b_sweep_gpu.py [first seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_foreign as T
first, count = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2000, 120)
bad = 0
sizes = [(416, 240), (352, 288), (200, 136), (648, 360), (64, 64), (136, 72)]
for seed in range(first, first + count):
    w, h = sizes[seed % len(sizes)]
    try:
        T.run_stream(w, h, 14, seed=seed, b_slices=(40, 70, 100)[seed % 3], gop=(0, 2, 4, 8)[(seed // 3) % 4], intra_period=6 + seed % 7,
                     threads=3 if seed & 1 else 1, frame_threads=bool(seed & 1))
    except BaseException as e:      # noqa: BLE001 (pytest.fail raises an exception of its own)
        bad += 1
        print(seed, "FAIL", repr(e)[:300], flush=True)
print("seeds", first, first + count, "bad", bad)
