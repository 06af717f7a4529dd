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


@pytest.mark.gpu
@pytest.mark.parametrize("frame", [False, True])
@pytest.mark.parametrize("kw", [
    dict(slices=3, wpp=1, cip=1, pcm=15, lf_across=1, intra_in_p=40, long_term=1, num_refs=3, tmvp=1),
    dict(slices=3, wpp=0, cip=1, pcm=15, lf_across=2, intra_in_p=40, ctb_log2=4, min_cb_log2=4),
    dict(slices=2, tile_rows=2, tile_cols=2, wpp=0, lf_across=2, cip=1, intra_in_p=40, ctb_log2=5),
])
def test_decoder_survives_corrupted_streams_of_the_new_kinds(gpu, frame, kw):
    """bit flips, truncations, garbage, dropped and swapped slice segments in streams with free slices, closed filter boundaries, PCM units, long-term references and
    constrained intra prediction: every call returns (a picture, nothing, or an error code), nothing hangs, no kernel reads what a damaged header promised -- and from
    the next clean IDR picture on the output is the checker's again"""
    import os
    import numpy as np
    import orc
    from kvazzup_amd.codec import Decoder, split_nals
    w, h, period = 208, 144, 8
    g = orc.OracleGen(w, h, seed=93, intra_period=period, density=30, sao=1, all_part_modes=1, **kw)
    aus = [g.picture() for _ in range(3 * period)]
    g.close()
    od = orc.OracleDecoder()
    want = []
    for t in range(2 * period, 3 * period):
        want += [f["i420"] for f in od.decode_au(aus[t], t)]
    want += [f["i420"] for f in od.flush()]
    od.close()
    assert len(want) == period
    rng = np.random.default_rng(777)
    gd = Decoder(threads=4, frame_threads=True) if frame else Decoder()
    errors = 0
    for trial in range(int(os.environ.get("KVZ_FUZZ_TRIALS", "120"))):
        t = int(rng.integers(0, 2 * period))
        nals = [bytearray(n) for n in split_nals(aus[t])]
        kind = trial % 6
        i = int(rng.integers(0, len(nals)))
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                nals[i][min(5 + int(rng.integers(0, max(len(nals[i]) - 5, 1))), len(nals[i]) - 1)] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            nals[i] = nals[i][:max(6, int(rng.integers(6, len(nals[i]) + 1)))]
        elif kind == 2:
            p = int(rng.integers(6, max(7, len(nals[i]))))
            nals[i][p:p + 16] = bytes(rng.integers(0, 256, 16, dtype=np.uint8))
        elif kind == 3 and len(nals) > 1:
            del nals[i]                                   # a lost slice segment (or parameter set)
        elif kind == 4 and len(nals) > 1:
            j = int(rng.integers(0, len(nals))); nals[i], nals[j] = nals[j], nals[i]
        else:
            nals.insert(i, bytearray(nals[i]))            # a duplicated one
        for nal in nals:
            try:
                gd.decode_nal(bytes(nal), t)
            except RuntimeError:
                errors += 1
    assert errors > 0
    got = []
    for t in range(2 * period, 3 * period):
        try:
            got += gd.decode_au(aus[t], t)
        except RuntimeError:
            pass                                          # (what the damage left behind may fail once more before the IDR takes over)
    eos, quiet = bytes([0, 0, 0, 1, 36 << 1, 1]), 0
    for _ in range(80):                                   # (a damaged picture still in the frame threads' ring fails when its turn comes: one error per such picture)
        try:
            f = gd.decode_nal(eos)
        except RuntimeError:
            continue
        quiet = 0 if f is not None else quiet + 1
        if f is not None:
            got.append(f)
        elif quiet > 8:
            break
    gd.close()
    assert len(got) >= period and all(np.array_equal(a["i420"], b) for a, b in zip(got[-period:], want)), len(got)
