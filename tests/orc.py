"""ctypes bindings of the CPU checker (oracle/build/liboracle.so).  Test infrastructure."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def build():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "build", "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_api_enc_open.restype = C.c_void_p
        L.orc_api_enc_open.argtypes = [C.c_int] * 10
        L.orc_api_enc_encode.restype = C.c_long
        L.orc_api_enc_encode.argtypes = [C.c_void_p] * 5 + [C.c_long]
        L.orc_enc_close.argtypes = [C.c_void_p]
        L.orc_enc_get_debug.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_enc_get_recon.argtypes = [C.c_void_p] * 4
        L.orc_dec_open.restype = C.c_void_p
        L.orc_dec_close.argtypes = [C.c_void_p]
        L.orc_dec_decode_nal.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64]
        L.orc_dec_get_frame.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_dec_predeblock_plane.restype = C.c_void_p
        L.orc_dec_predeblock_plane.argtypes = [C.c_void_p, C.c_int]
        L.orc_api_split_nals.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_int]
        L.orc_synth_frame.argtypes = [C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_api_closed_loop.argtypes = [C.c_int] * 7 + [C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


class EncDebug(C.Structure):
    _fields_ = [("coded_w", C.c_int), ("coded_h", C.c_int), ("is_intra", C.c_int), ("poc", C.c_int),
                ("cu_log2", C.c_void_p), ("cu_intra", C.c_void_p), ("cu_flags", C.c_void_p), ("cu_merge_idx", C.c_void_p),
                ("cu_mvp_idx", C.c_void_p), ("cu_intra_mode", C.c_void_p), ("cu_cbf", C.c_void_p), ("cu_mv", C.c_void_p),
                ("coef", C.c_void_p * 3), ("predeblock", C.c_void_p * 3), ("recon", C.c_void_p * 3),
                ("bs_v", C.c_void_p), ("bs_h", C.c_void_p), ("bins", C.c_uint64)]


class DecFrame(C.Structure):
    _fields_ = [("plane", C.c_void_p * 3), ("stride", C.c_int * 3), ("width", C.c_int), ("height", C.c_int),
                ("coded_width", C.c_int), ("coded_height", C.c_int), ("poc", C.c_int), ("pts", C.c_int64),
                ("fps_num", C.c_uint32), ("fps_den", C.c_uint32), ("slice_type", C.c_int)]


def _arr(ptr, shape, dtype):
    n = int(np.prod(shape))
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()


def synth_frame(kind, seed, w, h, t):
    out = np.empty(w * h * 3 // 2, dtype=np.uint8)
    lib().orc_synth_frame(kind, seed, w, h, t, out.ctypes.data)
    return out


def split_nals(au):
    au = np.ascontiguousarray(np.frombuffer(bytes(au), dtype=np.uint8))
    offs = (C.c_long * 64)()
    n = lib().orc_api_split_nals(au.ctypes.data, len(au), offs, 64)
    o = list(offs[:n]) + [len(au)]
    return [bytes(au[o[i]:o[i + 1]]) for i in range(n)]


class OracleEncoder:
    def __init__(self, w, h, qp=32, period=64, vps_period=1, me_range=16, fps=(30, 1), wpp=1, deblock=1, bitrate=0, tile_rows=1, qp_in_cu=0, sao=0, mv_jitter=0, mv_frame=0, vaq=0, me_early=1, satd=1, subme=0, rc_bands=0, slices=0, tile_cols=1):
        self.w, self.h = w, h
        lib().orc_api_enc_open_ex2.restype = C.c_void_p
        self.p = lib().orc_api_enc_open_ex2(w, h, qp, period, vps_period, me_range, fps[0], fps[1], wpp, deblock, bitrate, (tile_rows & 0xff) | ((int(rc_bands) & 15) << 8) | ((int(slices) & 3) << 12) | (int(bool(qp_in_cu)) << 16) | (int(bool(sao)) << 17) | (int(bool(mv_jitter)) << 18) | ((int(mv_frame) & 3) << 19) | ((int(vaq) & 31) << 21) | ((0 if me_early else 1) << 26) | ((0 if satd else 1) << 27) | ((int(subme) & 7) << 28), int(tile_cols))
        if not self.p:
            raise RuntimeError("orc_enc_open failed")
        self.buf = np.empty(w * h * 3 + (1 << 20), dtype=np.uint8)

    def set_option(self, name, value):
        """options by name (oracle/hevc_enc.h orc_enc_set_option): "hash" 0 none / 1 checksum / 2 md5, ..."""
        lib().orc_enc_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        if not lib().orc_enc_set_option(self.p, name.encode(), int(value)):
            raise ValueError("oracle encoder: unknown option %s" % name)

    def set_roi(self, rw, rh, deltas):
        """delta-QP map (kvz_picture.roi): rw x rh int8 cells over the picture; rw = 0 removes it"""
        a = np.ascontiguousarray(deltas, dtype=np.int8) if rw else np.zeros(1, np.int8)
        lib().orc_api_enc_set_roi.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib().orc_api_enc_set_roi(self.p, rw, rh, a.ctypes.data)

    def encode(self, i420):
        i420 = np.ascontiguousarray(i420, dtype=np.uint8)
        ny = self.w * self.h
        base = i420.ctypes.data
        n = lib().orc_api_enc_encode(self.p, base, base + ny, base + ny + ny // 4, self.buf.ctypes.data, len(self.buf))
        assert n <= len(self.buf)
        return bytes(self.buf[:n])

    def debug(self):
        d = EncDebug()
        lib().orc_enc_get_debug(self.p, C.byref(d))
        cw, ch = d.coded_w, d.coded_h
        b8 = (ch // 8, cw // 8)
        out = {"coded_w": cw, "coded_h": ch, "is_intra": d.is_intra, "poc": d.poc, "bins": d.bins}
        for k in ("cu_log2", "cu_intra", "cu_flags", "cu_merge_idx", "cu_mvp_idx", "cu_intra_mode", "cu_cbf"):
            out[k] = _arr(getattr(d, k), b8, np.uint8)
        out["cu_mv"] = _arr(d.cu_mv, b8 + (2,), np.int16)
        for c in range(3):
            shp = (ch, cw) if c == 0 else (ch // 2, cw // 2)
            out["coef%d" % c] = _arr(d.coef[c], shp, np.int16)
            out["predeblock%d" % c] = _arr(d.predeblock[c], shp, np.uint8)
            out["rec%d" % c] = _arr(d.recon[c], shp, np.uint8)
        out["bs_v"] = _arr(d.bs_v, (ch // 4, cw // 8), np.uint8)
        out["bs_h"] = _arr(d.bs_h, (ch // 8, cw // 4), np.uint8)
        return out

    def recon(self):
        out = np.empty(self.w * self.h * 3 // 2, dtype=np.uint8)
        ny = self.w * self.h
        lib().orc_enc_get_recon(self.p, out.ctypes.data, out.ctypes.data + ny, out.ctypes.data + ny + ny // 4)
        return out

    def close(self):
        if self.p:
            lib().orc_enc_close(self.p)
            self.p = None

    def __del__(self):
        self.close()


class OracleDecoder:
    def __init__(self):
        self.p = lib().orc_dec_open()

    def decode_nal(self, nal, pts=0):
        b = np.frombuffer(bytes(nal), dtype=np.uint8)
        return lib().orc_dec_decode_nal(self.p, b.ctypes.data, len(b), pts)

    def get_frame(self):
        f = DecFrame()
        if not lib().orc_dec_get_frame(self.p, C.byref(f)):
            return None
        planes = []
        for c in range(3):
            w = f.width if c == 0 else f.width // 2
            h = f.height if c == 0 else f.height // 2
            full = _arr(f.plane[c], (h, f.stride[c]), np.uint8) if w == f.stride[c] else None
            if full is None:
                rows = [np.frombuffer((C.c_char * w).from_address(f.plane[c] + y * f.stride[c]), dtype=np.uint8).copy() for y in range(h)]
                full = np.stack(rows)
            planes.append(full[:, :w])
        i420 = np.concatenate([p.reshape(-1) for p in planes])
        return {"i420": i420, "width": f.width, "height": f.height, "poc": f.poc, "pts": f.pts,
                "fps": (f.fps_num, f.fps_den), "slice_type": f.slice_type}

    def hash_stats(self):
        """(decoded picture hash SEI messages checked, of which mismatching)"""
        a, b = C.c_int(), C.c_int()
        lib().orc_dec_hash_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib().orc_dec_hash_stats(self.p, C.byref(a), C.byref(b))
        return a.value, b.value

    def concealed(self):
        """reference pictures that were missing and replaced by grey ones so far"""
        lib().orc_dec_concealed.argtypes = [C.c_void_p]
        return lib().orc_dec_concealed(self.p)

    def decode_au(self, au, pts=0):
        """feed every NAL of an access unit; returns list of decoded frames"""
        frames = []
        for nal in split_nals(au):
            r = self.decode_nal(nal, pts)
            if r < 0:
                raise RuntimeError("oracle decoder error %d" % r)
            while r > 0:                      # (pictures come in output order: one NAL unit may release several, or none)
                fr = self.get_frame()
                if fr is None:
                    break
                frames.append(fr)
        return frames

    def flush(self):
        """end of stream: the pictures still held back for reordering, in output order"""
        lib().orc_dec_flush.argtypes = [C.c_void_p]
        lib().orc_dec_flush(self.p)
        frames = []
        while True:
            fr = self.get_frame()
            if fr is None:
                return frames
            frames.append(fr)

    def close(self):
        if self.p:
            lib().orc_dec_close(self.p)
            self.p = None

    def __del__(self):
        self.close()


GEN_FIELDS = ("width", "height", "seed", "intra_period", "qp", "density", "num_refs", "tmvp", "amp", "sao", "strong_intra", "sign_hiding",
              "transform_skip", "cabac_init", "wpp", "tile_rows", "uniform_tiles", "th_depth_inter", "th_depth_intra", "qp_delta",
              "chroma_qp_offsets", "deblock_mode", "par_mrg_level", "intra_in_p", "all_part_modes", "chroma_modes", "nxn_intra",
              "max_cu_log2", "min_cu_log2", "big_mvd", "slices", "tile_cols", "tq_bypass", "scaling_lists", "b_slices", "gop", "weighted", "list_mod", "ctb_log2", "cip", "long_term", "pcm", "lf_across", "min_cb_log2", "open_gop", "hidden_pics", "temporal_layers", "vui_extras", "rps_forms", "hdr_extras")


class OracleGen:
    """conformance-style stream synthesiser (oracle/hevc_gen.c): random but valid Main-profile syntax covering the tools a foreign
    encoder uses.  Keyword arguments are the orc_gen_config fields; anything not given is drawn from the seed."""

    def __init__(self, width, height, seed=1, **kw):
        cfg = {k: -1 for k in GEN_FIELDS}
        cfg.update(width=width, height=height, seed=seed, intra_period=8, qp=30, density=30)
        for k, v in kw.items():
            if k not in cfg:
                raise KeyError(k)
            cfg[k] = v
        arr = (C.c_int * len(GEN_FIELDS))(*[int(cfg[k]) for k in GEN_FIELDS])
        L = lib()
        L.orc_api_gen_open.restype = C.c_void_p
        L.orc_api_gen_open.argtypes = [C.c_void_p, C.c_int]
        L.orc_api_gen_picture.restype = C.c_long
        L.orc_api_gen_picture.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.orc_api_gen_config.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_gen_close.argtypes = [C.c_void_p]
        self.p = L.orc_api_gen_open(arr, len(GEN_FIELDS))
        if not self.p:
            raise RuntimeError("orc_gen_open failed")
        self.buf = np.empty(width * height * 4 + (1 << 20), dtype=np.uint8)
        out = (C.c_int * len(GEN_FIELDS))()
        L.orc_api_gen_config(self.p, out, len(GEN_FIELDS))
        self.config = dict(zip(GEN_FIELDS, [int(v) for v in out]))

    def picture(self):
        n = lib().orc_api_gen_picture(self.p, self.buf.ctypes.data, len(self.buf))
        assert n <= len(self.buf)
        return bytes(self.buf[:n])

    def close(self):
        if self.p:
            lib().orc_gen_close(self.p)
            self.p = None

    def __del__(self):
        self.close()
