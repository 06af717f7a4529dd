"""ctypes bindings of tests/hostcheck (host build of the product's serial code).  Test infrastructure."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


class HcFrame(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("cw", "ch", "width", "height", "qp", "is_intra", "poc", "wpp", "deblock", "fps_num", "fps_den", "write_ps")] + \
               [(n, C.c_void_p) for n in ("cu_log2", "cu_intra", "cu_flags", "cu_merge_idx", "cu_mvp_idx", "cu_intra_mode", "cu_cbf", "cu_mv", "cu_mvd")] + \
               [("coef", C.c_void_p * 3)]


def lib():
    global _LIB
    if _LIB is None:
        d = os.path.join(ROOT, "tests", "hostcheck")
        import fcntl
        with open(os.path.join(d, ".build.lock"), "w") as lk:      # (pytest -n: several workers reach this at once, and a half-linked library must not be loaded)
            fcntl.flock(lk, fcntl.LOCK_EX)
            subprocess.run(["make", "-s", "-C", d], check=True, stdout=subprocess.DEVNULL)
        L = C.CDLL(os.path.join(d, "build", "libhostcheck.so"))
        L.hc_encode_au.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.hc_encode_au_tokens.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.hc_inter_signal.argtypes = [C.c_void_p]
        L.hc_deblock.argtypes = [C.c_void_p] * 4
        L.hc_intra_predict.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.hc_table.argtypes = [C.c_int, C.c_void_p]
        _LIB = L
    return _LIB


def make_frame(dbg, width, height, qp, wpp=1, deblock=1, fps=(30, 1), write_ps=1):
    """dbg: dict of numpy arrays as produced by orc.OracleEncoder.debug(); arrays are kept alive in the returned holder"""
    hold = {k: np.ascontiguousarray(v).copy() for k, v in dbg.items() if isinstance(v, np.ndarray)}
    if "cu_mvd" not in hold:
        hold["cu_mvd"] = np.zeros_like(hold["cu_mv"])
    f = HcFrame()
    f.cw, f.ch, f.width, f.height, f.qp = dbg["coded_w"], dbg["coded_h"], width, height, qp
    f.is_intra, f.poc, f.wpp, f.deblock, f.fps_num, f.fps_den, f.write_ps = dbg["is_intra"], dbg["poc"], wpp, deblock, fps[0], fps[1], write_ps
    for n in ("cu_log2", "cu_intra", "cu_flags", "cu_merge_idx", "cu_mvp_idx", "cu_intra_mode", "cu_cbf", "cu_mv", "cu_mvd"):
        setattr(f, n, hold[n].ctypes.data)
    for c in range(3):
        f.coef[c] = hold["coef%d" % c].ctypes.data
    return f, hold


def encode_au(f):
    out = np.empty(f.cw * f.ch * 3 + (1 << 16), dtype=np.uint8)
    bins = C.c_ulonglong()
    n = lib().hc_encode_au(C.byref(f), out.ctypes.data, len(out), C.byref(bins))
    assert n > 0
    return bytes(out[:n]), bins.value


def encode_au_tokens(f):
    out = np.empty(f.cw * f.ch * 3 + (1 << 16), dtype=np.uint8)
    nt = C.c_ulonglong()
    n = lib().hc_encode_au_tokens(C.byref(f), out.ctypes.data, len(out), C.byref(nt))
    assert n > 0, n
    return bytes(out[:n]), nt.value
