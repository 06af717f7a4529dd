"""Thin Python drivers over the C ABI, used by tests/ and bench.py.  They perform exactly the
call sequences of uvgComm's filters (kvazaarfilter.cpp:145-299,407-476; openhevcfilter.cpp:36-56,
112-172,194-237); the C++ mirrors of those filters live in csrc/filters.hip."""
import ctypes as C
import numpy as np
from . import _native as N

DEFAULT_OPTIONS = (("preset", "ultrafast"), ("threads", "0"), ("owf", "0"), ("wpp", "1"), ("qp", "32"), ("period", "64"),
                   ("vps-period", "1"), ("intra-bits", ""), ("gop", "lp-g4d3t1"), ("scaling-list", "off"), ("mv-constraint", "none"))


class Encoder:
    """kvz_api driven the way KvazaarFilter::init / feedInput drive it."""

    def __init__(self, width, height, fps=(30, 1), options=(), fields=None):
        self.lib = N.load_library()
        self.api = self.lib.kvz_api_get(8).contents
        self.w, self.h = width, height
        self.cfg = self.api.config_alloc()
        if not self.cfg:
            raise RuntimeError("config_alloc failed")
        self.api.config_init(self.cfg)
        opts = dict(DEFAULT_OPTIONS)
        opts["input-res"] = "%dx%d" % (width, height)
        opts["input-fps"] = "%d/%d" % fps
        for k, v in options:
            opts[k] = str(v)
        self.rejected = []
        for k, v in opts.items():
            if self.api.config_parse(self.cfg, k.encode(), v.encode()) != 1:
                self.rejected.append(k)
        self.cfg.contents.target_bitrate = 0
        self.cfg.contents.hash = 0
        for k, v in (fields or {}).items():
            setattr(self.cfg.contents, k, v)
        self.enc = self.api.encoder_open(self.cfg)
        if not self.enc:
            self.api.config_destroy(self.cfg)
            self.cfg = None
            raise RuntimeError("kvz_api.encoder_open failed (no usable HIP device? there is no CPU fallback)")
        # the ring of input pictures KvazaarFilter keeps (createInputVector(owf + 1), getNextPic: kvazaarfilter.cpp:32-42,76-88): the encoder
        # reads a picture where it lies until its access unit has come back, so a picture is not written again before that
        self.pics = [self.api.picture_alloc(width, height) for _ in range(int(self.cfg.contents.owf) + 1)]
        self.next_pic = 0
        self.pts = 0
        self._au = np.empty(width * height * 3 + (1 << 20), dtype=np.uint8)

    @property
    def pic(self):
        """the kvz_picture the next encode() call fills and hands over (tests set its roi fields, as KvazaarFilter::feedInput does)"""
        return self.pics[self.next_pic]

    # -- the reference call sequence: memcpy into kvz_picture, encoder_encode, drain chunks
    def encode(self, i420, want_recon=True):
        """i420 = None flushes (pic_in == NULL).  Returns (None, None) when the call produced no output
        (owf >= 1: the first call, or a flush with nothing in flight)."""
        ny = self.w * self.h
        if i420 is not None:
            i420 = np.ascontiguousarray(i420, dtype=np.uint8)
            pic = self.pics[self.next_pic]
            self.next_pic = (self.next_pic + 1) % len(self.pics)
            p = pic.contents
            C.memmove(p.y, i420.ctypes.data, ny)
            C.memmove(p.u, i420.ctypes.data + ny, ny // 4)
            C.memmove(p.v, i420.ctypes.data + ny + ny // 4, ny // 4)
            p.pts = self.pts
            self.pts += 1
        chunks = C.POINTER(N.KvzDataChunk)()
        length = C.c_uint32(0)
        recon = C.POINTER(N.KvzPicture)()
        info = N.KvzFrameInfo()
        ok = self.api.encoder_encode(self.enc, pic if i420 is not None else None, C.byref(chunks), C.byref(length),
                                     C.byref(recon) if want_recon else None, None, C.byref(info))
        if not ok:
            raise RuntimeError("encoder_encode failed")
        if not chunks:
            return None, None
        parts = []
        c = chunks
        while c:
            parts.append(bytes(c.contents.data[:c.contents.len]))
            c = c.contents.next
        self.api.chunk_free(chunks)
        au = b"".join(parts)
        assert len(au) == length.value
        rec = None
        if want_recon and recon:
            r = recon.contents
            rec = np.concatenate([np.frombuffer((C.c_char * n).from_address(ptr), dtype=np.uint8).copy()
                                  for ptr, n in ((r.y, ny), (r.u, ny // 4), (r.v, ny // 4))])
            self.api.picture_free(recon)
        self.info = {"poc": info.poc, "qp": info.qp, "nal_unit_type": info.nal_unit_type, "slice_type": info.slice_type}
        return au, rec

    # -- extension entry points
    def encode_device(self, dptr):
        n = C.c_uint32(0)
        info = N.KvzFrameInfo()
        ok = self.lib.kvzx_encoder_encode_device(self.enc, dptr, self._au.ctypes.data, len(self._au), C.byref(n), C.byref(info))
        if not ok:
            raise RuntimeError("kvzx_encoder_encode_device failed")
        return bytes(self._au[:n.value])

    def coded_size(self):
        a, b = C.c_int(), C.c_int()
        self.lib.kvzx_encoder_coded_size(self.enc, C.byref(a), C.byref(b))
        return a.value, b.value

    def debug(self, what, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        if not self.lib.kvzx_encoder_debug_copy(self.enc, what.encode(), out.ctypes.data, out.nbytes):
            raise RuntimeError("debug_copy(%s) failed" % what)
        return out

    def debug_all(self):
        cw, ch = self.coded_size()
        b8 = (ch // 8, cw // 8)
        d = {"coded_w": cw, "coded_h": ch}
        for k in ("cu_log2", "cu_intra", "cu_flags", "cu_merge_idx", "cu_mvp_idx", "cu_intra_mode", "cu_cbf"):
            d[k] = self.debug(k, np.uint8, b8)
        d["cu_mv"] = self.debug("cu_mv", np.int16, b8 + (2,))
        for c in range(3):
            shp = (ch, cw) if c == 0 else (ch // 2, cw // 2)
            d["coef%d" % c] = self.debug("coef%d" % c, np.int16, shp)
            d["rec%d" % c] = self.debug("rec%d" % c, np.uint8, shp)
            d["src%d" % c] = self.debug("src%d" % c, np.uint8, shp)
        return d

    def set_profiling(self, on):
        self.lib.kvzx_encoder_set_profiling(self.enc, int(on))

    def kernel_times(self, reset=True):
        ms = (C.c_double * 16)()
        n = (C.c_uint64 * 16)()
        k = self.lib.kvzx_encoder_kernel_times(self.enc, ms, n, int(reset))
        return {self.lib.kvzx_encoder_kernel_name(i).decode(): (ms[i], n[i]) for i in range(k)}

    def last_bins(self):
        return self.lib.kvzx_encoder_last_bins(self.enc)

    def close(self):
        if getattr(self, "enc", None):
            self.api.encoder_close(self.enc)
            self.enc = None
        for pic in getattr(self, "pics", []):
            self.api.picture_free(pic)
        self.pics = []
        if getattr(self, "cfg", None):
            self.api.config_destroy(self.cfg)
            self.cfg = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def split_nals(au):
    """Annex-B access unit -> list of NAL units, each with its 4-byte start code (what uvgRTP hands to
    OpenHEVCFilter one by one, /root/reference/src/media/delivery/uvgrtpreceiver.cpp:86-112)."""
    au = bytes(au)
    starts = []
    i = 0
    while True:
        j = au.find(b"\x00\x00\x00\x01", i)
        if j < 0:
            break
        starts.append(j)
        i = j + 4
    starts.append(len(au))
    return [au[starts[k]:starts[k + 1]] for k in range(len(starts) - 1)]


class Decoder:
    """libOpenHevc* driven the way OpenHEVCFilter::init / process / sendDecodedOutput drive it."""

    OH_THREAD_FRAME, OH_THREAD_SLICE = 1, 2

    def __init__(self, threads=1, download=True, device=None, frame_threads=False, temporal_layer=7, no_cropping=False):
        self.lib = N.load_library()
        self.threads, self.frame_threads = threads, bool(frame_threads) and threads > 1
        self.h = self.lib.libOpenHevcInit(threads, self.OH_THREAD_FRAME if frame_threads else self.OH_THREAD_SLICE)
        if device is not None:
            self.lib.kvzx_decoder_set_device(self.h, device)
        if self.lib.libOpenHevcStartDecoder(self.h) == -1:
            self.lib.libOpenHevcClose(self.h)
            self.h = None
            raise RuntimeError("libOpenHevcStartDecoder failed (no usable HIP device? there is no CPU fallback)")
        self.lib.libOpenHevcSetTemporalLayer_id(self.h, int(temporal_layer))      # (the highest sub-layer decoded: OpenHEVC's default 7 = all; uvgComm's filter passes 0, openhevcfilter.cpp:54)
        self.lib.libOpenHevcSetActiveDecoders(self.h, 0)
        if no_cropping:
            self.lib.libOpenHevcSetNoCropping(self.h, 1)                # (pictures at their coded size: the conformance window is not applied)
        self.lib.libOpenHevcSetViewLayers(self.h, 0)
        self.download = download
        if not download:
            self.lib.kvzx_decoder_set_download(self.h, 0)
        self.vps = self.sps = self.pps = False

    def decode_nal(self, nal, pts=0):
        """returns None or a dict with the packed I420 picture (the row-wise copy of sendDecodedOutput)"""
        nal = bytes(nal)
        if len(nal) < 6:                     # not a NAL unit (csrc/filters.hip OpenHEVCFilter::process drops it the same way)
            return None
        t = nal[4] >> 1
        self.vps |= t == 32
        self.sps |= t == 33
        self.pps |= t == 34
        vcl = t <= 31
        if not ((self.vps and self.sps and self.pps) or not vcl):
            return None
        buf = (C.c_ubyte * len(nal)).from_buffer_copy(nal)
        got = self.lib.libOpenHevcDecode(self.h, buf, len(nal), pts)
        if got < 0:
            raise RuntimeError("libOpenHevcDecode error %d" % got)
        if got == 0:
            return None
        fr = N.OpenHevcFrame()
        if self.lib.libOpenHevcGetOutput(self.h, got, C.byref(fr)) <= 0:
            return None
        self.lib.libOpenHevcGetPictureInfo(self.h, C.byref(fr.frameInfo))
        info = fr.frameInfo
        w, h = info.nWidth, info.nHeight
        out = {"width": w, "height": h, "fps": (info.frameRate.num, info.frameRate.den), "pts": info.nTimeStamp}
        if self.download:
            ys, qs = info.nYPitch, info.nUPitch // 2
            y = np.empty((h, w), np.uint8)
            u = np.empty((h // 2, w // 2), np.uint8)
            v = np.empty((h // 2, w // 2), np.uint8)
            for i in range(h):                              # openhevcfilter.cpp:218-229
                y[i] = np.frombuffer((C.c_char * w).from_address(fr.pvY + i * ys), dtype=np.uint8)
                if i % 2 == 0:
                    u[i // 2] = np.frombuffer((C.c_char * (w // 2)).from_address(fr.pvU + i * qs), dtype=np.uint8)
                    v[i // 2] = np.frombuffer((C.c_char * (w // 2)).from_address(fr.pvV + i * qs), dtype=np.uint8)
            out["i420"] = np.concatenate([y.reshape(-1), u.reshape(-1), v.reshape(-1)])
        return out

    def decode_au(self, au, pts=0):
        return [f for f in (self.decode_nal(n, pts) for n in split_nals(au)) if f is not None]

    def drain(self):
        """end-of-sequence NAL units hand out the pictures still held back: by the frame threads' ring, and -- a stream whose SPS allows
        reordering (B pictures in groups) -- by the output process, which lets go of them in POC order"""
        out = []
        ring = (self.threads + 1) if self.frame_threads else 1
        eos = bytes([0, 0, 0, 1, 36 << 1, 1])
        misses = 0
        for _ in range(ring + 20):                  # (ring + sps_max_num_reorder_pics <= 15 + slack)
            f = self.decode_nal(eos)
            if f is not None:
                out.append(f)
                misses = 0
            else:
                misses += 1
                if misses > ring:
                    break
        return out

    def output_device(self):
        planes = (C.c_void_p * 3)()
        pitches = (C.c_int * 3)()
        if not self.lib.kvzx_decoder_output_device(self.h, planes, pitches):
            return None
        return list(planes), list(pitches)

    def set_profiling(self, on):
        self.lib.kvzx_decoder_set_profiling(self.h, int(on))

    def kernel_times(self, reset=True):
        ms = (C.c_double * 16)()
        n = (C.c_uint64 * 16)()
        k = self.lib.kvzx_decoder_kernel_times(self.h, ms, n, int(reset))
        return {self.lib.kvzx_decoder_kernel_name(i).decode(): (ms[i], n[i]) for i in range(k)}

    def close(self):
        if getattr(self, "h", None):
            self.lib.libOpenHevcFlush(self.h)
            self.lib.libOpenHevcClose(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
