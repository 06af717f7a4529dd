"""Generates tests/golden/oracle_streams.json from the CPU checker (oracle/): per picture the MD5 of the
access unit and of the cropped reconstruction, for a handful of small synthetic clips.  The reference
tree holds no vectors for this path, so these pin the checker against itself over time and give the
GPU tests (test_gpu_golden.py) data that does not depend on a live oracle build."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402

CASES = [
    dict(w=128, h=64, qp=32, period=1, me_range=8, kind=0, seed=0x5EED0001, wpp=1, deblock=1, frames=2),
    dict(w=320, h=240, qp=32, period=64, me_range=16, kind=0, seed=0x5EED0002, wpp=1, deblock=1, frames=6),
    dict(w=320, h=240, qp=22, period=64, me_range=8, kind=2, seed=0x5EED0003, wpp=1, deblock=1, frames=3),
    dict(w=416, h=240, qp=27, period=4, me_range=32, kind=0, seed=0x5EED0004, wpp=0, deblock=1, frames=6),
    dict(w=192, h=128, qp=40, period=64, me_range=8, kind=1, seed=0x5EED0005, wpp=1, deblock=0, frames=3),
    dict(w=320, h=256, qp=30, period=3, me_range=16, kind=0, seed=0x5EED0006, wpp=1, deblock=1, frames=5, tile_rows=2, sao=1),
    dict(w=256, h=192, qp=35, period=64, me_range=8, kind=2, seed=0x5EED0007, wpp=0, deblock=1, frames=3, sao=1),
    dict(w=320, h=256, qp=30, period=64, me_range=16, kind=0, seed=0x5EED0008, wpp=1, deblock=1, frames=5, subme=4),
    dict(w=320, h=256, qp=27, period=4, me_range=8, kind=0, seed=0x5EED0009, wpp=1, deblock=1, frames=5, tile_rows=2, subme=2),
    dict(w=448, h=320, qp=30, period=4, me_range=8, kind=0, seed=0x5EED000A, wpp=1, deblock=1, frames=5, tile_rows=2, tile_cols=2, slices=2, sao=1),
    dict(w=320, h=256, qp=30, period=64, me_range=8, kind=0, seed=0x5EED000B, wpp=1, deblock=1, frames=4, slices=1),
    # round 4: uvgComm's "lossless" and "scaling list" boxes (kvazaarfilter.cpp:235-244)
    dict(w=320, h=240, qp=32, period=64, me_range=8, kind=0, seed=0x5EED000C, wpp=1, deblock=1, frames=4, subme=2, lossless=1),
    dict(w=192, h=128, qp=27, period=1, me_range=8, kind=2, seed=0x5EED000D, wpp=1, deblock=1, frames=2, tile_rows=2, lossless=1),
    dict(w=320, h=240, qp=30, period=64, me_range=8, kind=0, seed=0x5EED000E, wpp=1, deblock=1, frames=4, sao=1, scaling_list=1),
]

out = {"generator": "tests/golden/make_golden.py", "source": "oracle/ (CPU checker)", "cases": []}
for c in CASES:
    e = orc.OracleEncoder(c["w"], c["h"], qp=c["qp"], period=c["period"], me_range=c["me_range"], wpp=c["wpp"], deblock=c["deblock"],
                           tile_rows=c.get("tile_rows", 1), sao=c.get("sao", 0), subme=c.get("subme", 0), tile_cols=c.get("tile_cols", 1), slices=c.get("slices", 0))
    if c.get("scaling_list"):
        e.set_option("scaling-list", 1)
    if c.get("lossless"):
        e.set_option("lossless", 1)
    frames = []
    for t in range(c["frames"]):
        au = e.encode(orc.synth_frame(c["kind"], c["seed"], c["w"], c["h"], t))
        frames.append({"au_bytes": len(au), "au_md5": hashlib.md5(au).hexdigest(), "recon_md5": hashlib.md5(e.recon().tobytes()).hexdigest()})
    e.close()
    out["cases"].append({"config": {k: v for k, v in c.items() if k != "frames"}, "frames": frames})
with open(os.path.join(HERE, "oracle_streams.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote", len(out["cases"]), "cases")
