"""k_tokenize's list form (P pictures: a fixed number of waves works off k_inter_signal's list of (unit, role) pairs, enc_kernels.hip) is what pictures of more than
16 384 units run (2160p); the smaller test pictures run the grid form.  Here the list form is FORCED (KVAZZUP_AMD_TOK_LIST=1, read once per process: a child
process) for small pictures with every kind of unit -- skipped blocks, 32x32 and 16x16 units, intra units in P pictures, tiles, SAO syntax, per-CTU QPs -- against
the checker, byte for byte."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, orc
from kvazzup_amd import synth
from kvazzup_amd.codec import Encoder
SEED = 0x5EED0000
CASES = [
    dict(w=416, h=240, frames=5),
    dict(w=320, h=240, frames=4, qp=22, kind=2),                                        # noise: every unit has residual of every component
    dict(w=192, h=128, frames=3, kind=1),                                               # flat: headers only
    dict(w=640, h=368, frames=5, subme=2, sao=1, intra_in_p=2, cut=2, tiles="2x2"),      # 8x8 intra units in P pictures (four coding units per unit), tiles, SAO syntax
    dict(w=320, h=256, frames=4, vaq=8, sao=1, qp=30),                                   # cu_qp_delta in the headers
    dict(w=1280, h=720, frames=4, subme=2, intra_in_p=1, cut=2, me_source=1),
]
for cfg in CASES:
    w, h = cfg["w"], cfg["h"]
    tiles = cfg.get("tiles", "1x1"); tc, tr = [int(v) for v in tiles.split("x")]
    oe = orc.OracleEncoder(w, h, qp=cfg.get("qp", 32), period=64, me_range=8, subme=cfg.get("subme", 0), sao=cfg.get("sao", 0), tile_rows=tr, tile_cols=tc, vaq=cfg.get("vaq", 0))
    oe.set_option("intra-in-p", cfg.get("intra_in_p", 0)); oe.set_option("me-source", cfg.get("me_source", 0))
    ge = Encoder(w, h, options=(("qp", cfg.get("qp", 32)), ("period", 64), ("me-range", 8), ("subme", cfg.get("subme", 0)), ("sao", "full" if cfg.get("sao") else "off"),
                                ("intra-in-p", cfg.get("intra_in_p", 0)), ("me-source", cfg.get("me_source", 0))) + ((("tiles", tiles),) if tiles != "1x1" else ()) + ((("vaq", cfg["vaq"]),) if cfg.get("vaq") else ()))
    assert not ge.rejected, ge.rejected
    for t in range(cfg["frames"]):
        f = synth.scene_cut_frame(SEED, w, h, t, cfg["cut"]) if cfg.get("cut") else orc.synth_frame(cfg.get("kind", 0), SEED, w, h, t)
        au, rec = ge.encode(f)
        want = oe.encode(f)
        assert au == want, (cfg, t, len(au), len(want))
        assert np.array_equal(rec, oe.recon()), (cfg, t)
        assert ge.last_bins() == oe.debug()["bins"], (cfg, t)
    ge.close(); oe.close()
print("OK")
'''


@pytest.mark.gpu
def test_list_form_of_the_tokenizer_forced_for_small_pictures(gpu, tmp_path):
    script = tmp_path / "tok_list_child.py"
    script.write_text(CHILD % (ROOT, ROOT))
    r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, KVAZZUP_AMD_TOK_LIST="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
