import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc
from kvazzup_amd.codec import Decoder
seed, ctb, w, h = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
g = orc.OracleGen(w, h, seed=seed, ctb_log2=ctb, slices=3)
print({k: g.config[k] for k in ("wpp", "tmvp", "num_refs", "b_slices", "gop")})
gd = Decoder(threads=3, frame_threads=True)
for t in range(8):
    au = g.picture()
    for i, nal in enumerate(orc.split_nals(au)):
        ty = (nal[4 if nal[2] == 0 else 3] >> 1) & 63
        try:
            r = gd.decode_nal(nal, t)
        except RuntimeError as e:
            print("picture", t, "nal", i, "type", ty, "first bits", "{:08b}".format(nal[6 if nal[2] == 0 else 5]), e, "last_error", gd.lib.kvzx_decoder_last_error(gd.h)); sys.exit(0)
        print("picture", t, "nal", i, "type", ty, "->", "picture" if r else "-")
