"""Python handle on the C++ filter harness (csrc/filters.hip): source -> KvazaarFilter ->
[WireAdapter -> OpenHEVCFilter] -> sink, each filter on its own thread as in uvgComm's FilterGraph
(/root/reference/src/media/processing/filtergraph.cpp:347-351,576-577)."""
import ctypes as C
import numpy as np
from . import _native as N


def _bind(lib):
    if getattr(lib, "_uvgx_bound", False):
        return
    lib.uvgx_pipeline_create.restype = C.c_void_p
    lib.uvgx_pipeline_create.argtypes = [C.c_char_p, C.c_int, C.c_int]
    lib.uvgx_pipeline_push_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64]
    lib.uvgx_pipeline_push_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64]
    lib.uvgx_pipeline_wait.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
    lib.uvgx_pipeline_push_device_paced.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_uint32, C.c_int]
    lib.uvgx_pipeline_push_host_paced.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_uint32, C.c_int, C.c_int]
    lib.uvgx_pipeline_flush.argtypes = [C.c_void_p]
    lib.uvgx_pipeline_push_encoded.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_int64, C.c_uint32, C.c_int]
    lib.uvgx_pipeline_encoder_backlog.restype = C.c_uint32
    lib.uvgx_pipeline_encoder_backlog.argtypes = [C.c_void_p]
    lib.uvgx_pipeline_pop_encoded.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_int64)]
    lib.uvgx_pipeline_pop_decoded.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_int),
                                              C.POINTER(C.c_int), C.POINTER(C.c_int64)]
    lib.uvgx_pipeline_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.uvgx_pipeline_busy_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.uvgx_pipeline_latency_us.restype = C.c_uint32
    lib.uvgx_pipeline_latency_us.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.c_uint32, C.c_int]
    lib.uvgx_pipeline_delay_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.uvgx_pipeline_encoder.restype = C.c_void_p
    lib.uvgx_pipeline_encoder.argtypes = [C.c_void_p]
    lib.uvgx_pipeline_decoder.restype = C.c_void_p
    lib.uvgx_pipeline_decoder.argtypes = [C.c_void_p]
    lib.uvgx_pipeline_destroy.argtypes = [C.c_void_p]
    lib._uvgx_bound = True


DEFAULT_SETTINGS = {            # uvgComm defaults with rate control off (defaultsettings.cpp:266-281)
    "video/Preset": "ultrafast", "video/QP": "32", "video/Intra": "64", "video/VPS": "1", "video/WPP": "1", "video/OWF": "0",
    "video/Tiles": "0", "video/Slices": "0", "video/bitrate": "0", "video/scalingList": "0", "video/lossless": "0",
    "video/mvConstraint": "none", "video/vaq": "0", "video/kvzThreads": "auto", "video/OPENHEVC_threads": "1",
    "video/OH_parallelization": "Slice",
}


class Pipeline:
    def __init__(self, width, height, fps=(30, 1), settings=None, custom=(), loopback=True, keep_outputs=True):
        self.lib = N.load_library()
        _bind(self.lib)
        s = dict(DEFAULT_SETTINGS)
        s.update({"video/ResolutionWidth": str(width), "video/ResolutionHeight": str(height),
                  "video/FramerateNumerator": str(fps[0]), "video/FramerateDenominator": str(fps[1])})
        s.update({k: str(v) for k, v in (settings or {}).items()})
        s["parameters/size"] = str(len(custom))
        for i, (k, v) in enumerate(custom, 1):
            s["parameters/%d/Name" % i] = k
            s["parameters/%d/Value" % i] = str(v)
        text = "\n".join("%s=%s" % kv for kv in s.items())
        self.w, self.h, self.fps = width, height, fps
        self.p = self.lib.uvgx_pipeline_create(text.encode(), int(loopback), int(keep_outputs))
        if not self.p:
            raise RuntimeError("uvgx_pipeline_create failed (no usable HIP device? there is no CPU fallback)")
        self.loopback = loopback
        self.pushed = 0
        self._buf = np.empty(width * height * 3 + (1 << 20), dtype=np.uint8)

    def push(self, i420, pts=None):
        i420 = np.ascontiguousarray(i420, dtype=np.uint8)
        self.lib.uvgx_pipeline_push_host(self.p, i420.ctypes.data, self.w, self.h, self.fps[0], self.fps[1], self.pushed if pts is None else pts)
        self.pushed += 1

    def push_device(self, dptr, pts=None):
        self.lib.uvgx_pipeline_push_device(self.p, dptr, self.w, self.h, self.fps[0], self.fps[1], self.pushed if pts is None else pts)
        self.pushed += 1

    def push_device_paced(self, dptr, max_backlog=6, timeout_ms=60000, pts=None):
        """push_device that sleeps (in C) until the encoder filter buffers fewer than max_backlog pictures"""
        ok = self.lib.uvgx_pipeline_push_device_paced(self.p, dptr, self.w, self.h, self.fps[0], self.fps[1], self.pushed if pts is None else pts, max_backlog, timeout_ms)
        self.pushed += 1 if ok else 0
        return bool(ok)

    def push_host_paced(self, i420, max_backlog=6, timeout_ms=60000, pts=None, borrow=True):
        """a HOST picture (contiguous uint8 numpy array, w*h*3/2 bytes) through the reference's own boundary; with borrow the caller keeps
        the array alive and unchanged until the picture has been encoded"""
        ok = self.lib.uvgx_pipeline_push_host_paced(self.p, i420.ctypes.data, self.w, self.h, self.fps[0], self.fps[1], self.pushed if pts is None else pts, max_backlog, timeout_ms, int(borrow))
        self.pushed += 1 if ok else 0
        return bool(ok)

    def push_encoded(self, au, pts=0, max_backlog=40, timeout_ms=60000):
        """an access unit straight into the receiving side (WireAdapter -> OpenHEVCFilter), as a peer's stream arrives; None = flush marker"""
        if au is None:
            return bool(self.lib.uvgx_pipeline_push_encoded(self.p, None, 0, 0, 0, 0))
        au = bytes(au)
        return bool(self.lib.uvgx_pipeline_push_encoded(self.p, au, len(au), pts, max_backlog, timeout_ms))

    def flush(self):
        """everything pushed so far comes out without further input (harness only); the next picture pushed should be an IDR"""
        self.lib.uvgx_pipeline_flush(self.p)

    def wait(self, n, timeout_ms=60000):
        return bool(self.lib.uvgx_pipeline_wait(self.p, n, timeout_ms))

    def backlog(self):
        return self.lib.uvgx_pipeline_encoder_backlog(self.p)

    def pop_encoded(self):
        n, pts = C.c_uint32(), C.c_int64()
        r = self.lib.uvgx_pipeline_pop_encoded(self.p, self._buf.ctypes.data, len(self._buf), C.byref(n), C.byref(pts))
        return (bytes(self._buf[:n.value]), pts.value) if r == 1 else None

    def pop_decoded(self):
        n, pts, w, h = C.c_uint32(), C.c_int64(), C.c_int(), C.c_int()
        r = self.lib.uvgx_pipeline_pop_decoded(self.p, self._buf.ctypes.data, len(self._buf), C.byref(n), C.byref(w), C.byref(h), C.byref(pts))
        return {"i420": self._buf[:n.value].copy(), "width": w.value, "height": h.value, "pts": pts.value} if r == 1 else None

    def stats(self):
        out = (C.c_uint64 * 8)()
        self.lib.uvgx_pipeline_stats(self.p, out)
        keys = ("encoded_pictures", "encoded_bytes", "received_nals", "received_bytes", "dropped", "decoded_pictures", "encoding_delay_ms_sum", "encoder_inputs_discarded")
        return dict(zip(keys, [int(v) for v in out]))

    def busy_ms(self):
        """milliseconds each filter thread has spent inside process(): (encoder, wire adapter, decoder)"""
        out = (C.c_double * 3)()
        self.lib.uvgx_pipeline_busy_ms(self.p, out)
        return tuple(float(v) for v in out)

    def latency_us(self, which, reset=True):
        """per-picture delays in microseconds since the last reset, in output order: which 0 = encoding delay (pushed -> access unit out of the
        encoder filter, kvazaarfilter.cpp:478-479), 1 = total delay (-> decoded picture out of the last filter, displayfilter.cpp:113-115)"""
        n = self.lib.uvgx_pipeline_latency_us(self.p, which, None, 0, 0)
        buf = (C.c_uint32 * max(1, n))()
        n = min(n, self.lib.uvgx_pipeline_latency_us(self.p, which, buf, n, int(reset)))
        return np.frombuffer(buf, dtype=np.uint32, count=n).copy()

    def delay_stats(self):
        """the filters' own delay histograms (uvgx::Stats): {'encoding': (count, mean, p50, p99), 'total': (...)} in microseconds"""
        out = (C.c_double * 8)()
        self.lib.uvgx_pipeline_delay_stats(self.p, out)
        return {"encoding": tuple(out[0:4]), "total": tuple(out[4:8])}

    def encoder_handle(self):
        return self.lib.uvgx_pipeline_encoder(self.p)

    def decoder_handle(self):
        return self.lib.uvgx_pipeline_decoder(self.p)

    def close(self):
        if getattr(self, "p", None):
            self.lib.uvgx_pipeline_destroy(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
