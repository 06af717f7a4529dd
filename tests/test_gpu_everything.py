"""GPU parity, everything at once (round 6): every option the synthesiser has gained -- coding tree block and minimum coding block sizes, free slices, closed filter
boundaries, PCM units, long-term reference pictures, constrained intra prediction -- drawn together from the seed on top of the older ones, so that the features meet
each other (a PCM unit as the only intra neighbour under constrained intra prediction, a slice border through a long-term picture's list, ...).  The HIP decoder must
reproduce the checker's decoder bit for bit, synchronous and with frame threads."""
import random

import pytest

from test_gpu_foreign import run_stream


def drawn(seed):
    r = random.Random(1000 + seed)
    ctb = r.choice((6, 6, 5, 4))
    min_cb = r.choice([3, 3, 4] + ([5] if ctb >= 5 else []))
    sizes = [(416, 240), (352, 288), (192, 128), (640, 352), (128, 128)] if min_cb > 3 else [(416, 240), (352, 288), (200, 136), (648, 360), (64, 64)]
    w, h = r.choice(sizes)
    if min_cb == 5 and (w % 32 or h % 32):
        w, h = 192, 128
    layout = r.choice(("one", "free", "free", "tiles", "tile_slices"))
    kw = dict(ctb_log2=ctb, min_cb_log2=min_cb, cip=r.choice((0, 0, 1)), pcm=r.choice((0, 0, 15)), lf_across=r.choice((0, 1, 2)), intra_in_p=r.choice((10, 30, 50)))
    if layout == "free":
        kw["slices"] = 3
    elif layout == "tiles":
        kw.update(slices=0, tile_rows=r.choice((2, 3)), tile_cols=r.choice((1, 2, 3)))
    elif layout == "tile_slices":
        kw.update(slices=2, tile_rows=2, tile_cols=r.choice((1, 2)), wpp=0)
    else:
        kw["slices"] = 0
    if r.random() < 0.4 and layout in ("one", "free"):
        kw.update(long_term=1, gop=0, b_slices=0, intra_period=20)
    return w, h, kw


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(1, 49))
def test_everything_at_once(gpu, seed):
    w, h, kw = drawn(seed)
    run_stream(w, h, 12 if kw.get("long_term") else 6, seed=seed, threads=3 if seed & 1 else 1, frame_threads=bool(seed & 1), **kw)
