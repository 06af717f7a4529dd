"""Tile-row split of ONE picture stream over several processes, one GPU each (BASELINE configs[4]).

Every rank opens an encoder on its own device with the same configuration, `tiles = 1xN` and its band of CTU rows
(whole tile rows).  Per picture: band_phase1 on every rank; the two halo blocks (4 luma + 2 x 2 chroma rows with their
vertical edges filtered, and the CU records of one 8x8 row) go to rank - 1 / rank + 1 -- the one exchange step on the
data path, ~7 bytes per luma column and boundary; band_phase2; the substreams are gathered on rank 0, which assembles
the access unit.  torch.distributed is the transport: backend "nccl" (= RCCL over xGMI) moves the device blocks
directly; with "gloo" (CPU tests, several ranks sharing one GPU) they are staged through host memory.
"""
import ctypes as C

import numpy as np

from . import _native as N


def band_partition(ctu_rows, tile_rows, world):
    """contiguous whole tile rows per rank: [(first CTU row, CTU rows)] * world; tiles follow H.265 6.5.1 uniform spacing"""
    if tile_rows < world or tile_rows % world:
        raise ValueError("tile rows (%d) must be a multiple of the number of ranks (%d)" % (tile_rows, world))
    bd = [(i * ctu_rows) // tile_rows for i in range(tile_rows + 1)]
    per = tile_rows // world
    return [(bd[r * per], bd[(r + 1) * per] - bd[r * per]) for r in range(world)]


class BandEncoder:
    """one rank's share of the split encoder; `dist` is torch.distributed (initialised) or None for a single process"""

    def __init__(self, width, height, tile_rows, rank, world, options=(), device=0, dist=None, pipelined=False):
        self.torch, self.dist = None, dist
        if world > 1:
            import torch                        # (import torch before this library is first loaded in a process: two HIP runtimes)
            self.torch = torch
        self.lib = N.load_library()
        self.api = self.lib.kvz_api_get(8).contents
        self.w, self.h, self.rank, self.world = width, height, rank, world
        self.ctu_rows = (height + 63) // 64
        self.row0, self.nrows = band_partition(self.ctu_rows, tile_rows, world)[rank]
        self.cfg = self.api.config_alloc()
        self.api.config_init(self.cfg)
        opts = {"preset": "ultrafast", "input-res": "%dx%d" % (width, height), "input-fps": "30/1", "vps-period": "1", "tiles": "1x%d" % tile_rows, "gpu": str(device),
                "band-row0": str(self.row0), "band-rows": str(self.nrows)}
        opts.update({k: str(v) for k, v in options})
        for k, v in opts.items():
            if self.api.config_parse(self.cfg, k.encode(), v.encode()) != 1 and k != "preset":
                raise ValueError("option %s=%s rejected" % (k, v))
        if "bitrate" not in opts:
            self.cfg.contents.target_bitrate = 0
        self.enc = self.api.encoder_open(self.cfg)
        if not self.enc:
            raise RuntimeError("encoder_open failed (no usable HIP device? there is no CPU fallback)")
        L = self.lib
        L.kvzx_encoder_band_halo_bytes.restype = C.c_size_t
        L.kvzx_encoder_band_halo_bytes.argtypes = [C.c_void_p]
        L.kvzx_encoder_band_phase1.argtypes = [C.c_void_p, C.c_void_p]
        L.kvzx_encoder_band_report_au.restype = None
        L.kvzx_encoder_band_report_au.argtypes = [C.c_void_p, C.c_long, C.c_uint32]
        L.kvzx_encoder_band_export_halo.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.kvzx_encoder_band_import_halo.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.kvzx_encoder_band_phase2.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.kvzx_encoder_band_phase2a.argtypes = [C.c_void_p]
        L.kvzx_encoder_band_phase2b.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        nh = int(L.kvzx_encoder_band_halo_bytes(self.enc))
        self.halo_out = self.halo_in = None
        if world > 1:
            torch = self.torch
            self.dev = torch.device("cuda", device)
            self.halo_out = [torch.empty(nh, dtype=torch.uint8, device=self.dev) for _ in range(2)]    # up, down
            self.halo_in = [torch.empty(nh, dtype=torch.uint8, device=self.dev) for _ in range(2)]     # from up, from down
        self.buf = np.empty(width * height * 3 + (1 << 20), dtype=np.uint8)
        self.sizes = np.zeros(self.ctu_rows, dtype=np.uint32)
        self.halo_bytes_exchanged = 0
        self.pipelined, self.pending, self.intra_count = bool(pipelined), None, 0
        self.assembled, self.last_au, self.reported = 0, (-1, 0), -1      # rate control: access units assembled (rank 0), the latest (index, bytes), the latest index told to the encoder
        self.dev = getattr(self, "dev", None)

    # ---- the exchange step, in two halves: the transfers are started, phase 2a (inner horizontal edges, tokenizer, host arithmetic
    # coder -- nothing of it needs the neighbours) runs while they travel, then they are completed and imported
    def _exchange_start(self):
        d, t = self.dist, self.torch
        up, down = self.rank - 1, self.rank + 1
        have_up, have_down = up >= 0, down < self.world
        if d is None or self.world == 1:
            return None
        staged = d.get_backend() != "nccl"
        send = [x.cpu() if staged else x for x in self.halo_out]
        recv = [t.empty_like(x) for x in send]
        ops = []
        if have_up:
            ops += [d.P2POp(d.isend, send[0], up), d.P2POp(d.irecv, recv[0], up)]
        if have_down:
            ops += [d.P2POp(d.isend, send[1], down), d.P2POp(d.irecv, recv[1], down)]
        reqs = d.batch_isend_irecv(ops) if ops else []
        return (reqs, send, recv, staged, have_up, have_down)

    def _exchange_finish(self, st):
        if st is None:
            return False, False
        reqs, _send, recv, staged, have_up, have_down = st
        for r in reqs:
            r.wait()
        for i, have in enumerate((have_up, have_down)):
            if have:
                self.halo_in[i].copy_(recv[i])
                self.halo_bytes_exchanged += 2 * recv[i].numel()
        if not staged:
            self.torch.cuda.synchronize(self.dev)
        return have_up, have_down

    # ---- substreams to rank 0: fixed-size headers to everybody (every rank then knows the largest payload), payloads padded to that
    # size to rank 0.  No pickling; with `pipelined` the payload gather of picture t completes during picture t + 1.
    HDR = 6            # substream count, POC, QP, NAL type, index and size of the latest access unit rank 0 has assembled (rate control)

    def _report(self, index, nbytes):
        """every rank tells its encoder the size of access unit `index` (the same numbers everywhere: the controllers stay in step)"""
        if index > self.reported:
            self.lib.kvzx_encoder_band_report_au(self.enc, index, nbytes)
            self.reported = index

    def _gather_start(self, sizes, data, info):
        d, t = self.dist, self.torch
        mine = (sizes, data, info.poc, info.qp, info.nal_unit_type)
        if d is None or self.world == 1:
            return ("local", mine)
        dev = self.dev if d.get_backend() == "nccl" else "cpu"
        hdr = t.zeros(self.HDR + self.ctu_rows, dtype=t.int64, device=dev)
        hdr[:self.HDR + len(sizes)] = t.tensor([len(sizes), info.poc, info.qp, info.nal_unit_type, self.last_au[0], self.last_au[1]] + sizes, dtype=t.int64)
        hdrs = [t.empty_like(hdr) for _ in range(self.world)]
        d.all_gather(hdrs, hdr)
        hdrs = [h.cpu().tolist() for h in hdrs]
        if hdrs[0][4] >= 0:
            self._report(int(hdrs[0][4]), int(hdrs[0][5]))
        totals = [sum(h[self.HDR:self.HDR + h[0]]) for h in hdrs]
        cap = (max(totals) + 4095) & ~4095
        pay = t.zeros(cap, dtype=t.uint8)
        pay[:len(data)] = t.frombuffer(bytearray(data), dtype=t.uint8) if data else pay[:0]
        pay = pay.to(dev)
        outs = [t.empty_like(pay) for _ in range(self.world)] if self.rank == 0 else None
        work = d.gather(pay, outs, dst=0, async_op=True)
        return ("dist", work, hdrs, totals, outs, pay)

    def _gather_finish(self, st):
        if st[0] == "local":
            parts = [st[1]]
        else:
            _, work, hdrs, totals, outs, _pay = st
            work.wait()
            if self.rank != 0:
                return None
            parts = []
            for h, n, o in zip(hdrs, totals, outs):
                parts.append(([int(x) for x in h[self.HDR:self.HDR + h[0]]], bytes(o[:n].cpu().numpy().tobytes()), h[1], h[2], h[3]))
        if self.rank != 0:
            return None
        poc, nal = parts[0][2], parts[0][4]
        idr = nal == 19
        vp = self.cfg.contents.vps_period                 # parameter sets with every vps-period-th IDR, as the single encoder does
        write_ps = idr and (self.intra_count == 0 or (vp > 0 and self.intra_count % vp == 0))
        if idr:
            self.intra_count += 1
        au = assemble(self.lib, self.cfg, parts, write_ps)
        self.last_au = (self.assembled, len(au))
        self.assembled += 1
        if self.world == 1:
            self._report(*self.last_au)
        return au

    def encode(self, d_i420_ptr):
        """one picture; returns the access unit on rank 0 (None elsewhere).  With `pipelined` the access unit returned is the PREVIOUS
        picture's (None for the first call); flush() returns the last one."""
        L = self.lib
        import time
        T = self.times = getattr(self, "times", {})
        def lap(name, t0):
            T[name] = T.get(name, 0.0) + time.perf_counter() - t0
            return time.perf_counter()
        t0 = time.perf_counter()
        if not L.kvzx_encoder_band_phase1(self.enc, d_i420_ptr):
            raise RuntimeError("band_phase1 failed")
        t0 = lap("phase1", t0)
        if self.world > 1 and not L.kvzx_encoder_band_export_halo(self.enc, self.halo_out[0].data_ptr() if self.rank > 0 else None,
                                                                  self.halo_out[1].data_ptr() if self.rank + 1 < self.world else None):
            raise RuntimeError("band_export_halo failed")
        t0 = lap("export", t0)
        ex = self._exchange_start()
        t0 = lap("exchange_start", t0)
        if not L.kvzx_encoder_band_phase2a(self.enc):          # runs while the halo blocks travel
            raise RuntimeError("band_phase2a failed")
        t0 = lap("phase2a", t0)
        prev = None
        if self.pending is not None:                            # the previous picture's payloads have had phase 1 and 2a of this one to arrive
            prev = self._gather_finish(self.pending)
            self.pending = None
        t0 = lap("gather_finish", t0)
        have_up, have_down = self._exchange_finish(ex)
        t0 = lap("exchange_finish", t0)
        if self.world > 1 and not L.kvzx_encoder_band_import_halo(self.enc, self.halo_in[0].data_ptr() if have_up else None, self.halo_in[1].data_ptr() if have_down else None):
            raise RuntimeError("band_import_halo failed")
        nsub = C.c_int(0)
        info = N.KvzFrameInfo()
        if not L.kvzx_encoder_band_phase2b(self.enc, self.buf.ctypes.data, len(self.buf), self.sizes.ctypes.data, len(self.sizes), C.byref(nsub), C.byref(info)):
            raise RuntimeError("band_phase2b failed")
        t0 = lap("import+phase2b", t0)
        sizes = [int(x) for x in self.sizes[:nsub.value]]
        data = bytes(self.buf[:sum(sizes)])
        st = self._gather_start(sizes, data, info)
        t0 = lap("gather_start", t0)
        if self.pipelined:
            self.pending = st
            return prev
        return self._gather_finish(st)

    def flush(self):
        """pipelined mode: the access unit of the last picture (rank 0; None elsewhere or when nothing is pending)"""
        if self.pending is None:
            return None
        st, self.pending = self.pending, None
        return self._gather_finish(st)

    def close(self):
        if self.enc:
            self.api.encoder_close(self.enc)
            self.enc = None
        if self.cfg:
            self.api.config_destroy(self.cfg)
            self.cfg = None


class BandDecoder:
    """one rank's share of the split DECODER: every rank gets every NAL unit, parses its own tile rows' substreams and reconstructs its band
    of CTU rows; deblocking across the band boundaries takes two small exchanges per picture with rank - 1 / rank + 1 (the band's last four
    rows down before deblocking, the same rows back up afterwards).  `dist` is torch.distributed (initialised) or None for one process."""

    def __init__(self, ctu_rows, tile_rows, rank, world, device=0, dist=None, download=True):
        self.torch, self.dist, self.download = None, dist, download
        if world > 1 and dist is not None:
            import torch
            self.torch = torch
        self.lib = L = N.load_library()
        self.rank, self.world = rank, world
        self.row0, self.nrows = band_partition(ctu_rows, tile_rows, world)[rank]
        self.h = L.libOpenHevcInit(1, 0)
        L.kvzx_decoder_set_device(self.h, device)
        if world > 1:
            L.kvzx_decoder_set_band(self.h, self.row0, self.nrows)
        L.kvzx_decoder_set_download(self.h, 1 if download else 0)       # (off: the picture stays in HBM, kvzx_decoder_output_device)
        if L.libOpenHevcStartDecoder(self.h) == -1:
            raise RuntimeError("libOpenHevcStartDecoder failed (no usable HIP device? there is no CPU fallback)")
        self.device, self.halo = device, None

    def feed(self, nal, pts=0):
        """one NAL unit; returns True when the band of a picture has been reconstructed and waits for the exchange (world > 1), or the
        decoded picture (dict) when this is the only decoder"""
        buf = (C.c_ubyte * len(nal)).from_buffer_copy(nal)
        rc = self.lib.libOpenHevcDecode(C.c_void_p(self.h), buf, len(nal), C.c_int64(pts))
        if rc < 0:
            raise RuntimeError("libOpenHevcDecode error %d" % rc)
        if self.world == 1:
            return self._output() if rc > 0 else None
        return bool(self.lib.kvzx_decoder_band_ready(self.h))

    def finish_exchange(self):
        """the two exchanges and the deblocking in between, over torch.distributed; returns this rank's picture (its band's rows valid)"""
        t, d, L = self.torch, self.dist, self.lib
        if self.halo is None:
            n = int(L.kvzx_decoder_band_halo_bytes(C.c_void_p(self.h)))
            dev = t.device("cuda", self.device)
            self.halo = [t.empty(n, dtype=t.uint8, device=dev) for _ in range(2)]      # out, in
        up, down = self.rank - 1, self.rank + 1
        staged = d.get_backend() != "nccl"
        def swap(send_to, recv_from):
            ops, s, r = [], None, None
            if send_to is not None:
                s = self.halo[0].cpu() if staged else self.halo[0]
                ops.append(d.P2POp(d.isend, s, send_to))
            if recv_from is not None:
                r = t.empty_like(self.halo[1].cpu() if staged else self.halo[1])
                ops.append(d.P2POp(d.irecv, r, recv_from))
            for q in (d.batch_isend_irecv(ops) if ops else []):
                q.wait()
            if r is not None:
                self.halo[1].copy_(r)
                t.cuda.synchronize()
        have_up, have_down = up >= 0, down < self.world
        if have_down and not L.kvzx_decoder_band_export(C.c_void_p(self.h), 0, C.c_void_p(self.halo[0].data_ptr())):
            raise RuntimeError("band_export(0) failed")
        swap(down if have_down else None, up if have_up else None)
        if have_up and not L.kvzx_decoder_band_import(C.c_void_p(self.h), 0, C.c_void_p(self.halo[1].data_ptr())):
            raise RuntimeError("band_import(0) failed")
        if not L.kvzx_decoder_band_deblock(C.c_void_p(self.h)):
            raise RuntimeError("band_deblock failed")
        if have_up and not L.kvzx_decoder_band_export(C.c_void_p(self.h), 1, C.c_void_p(self.halo[0].data_ptr())):
            raise RuntimeError("band_export(1) failed")
        swap(up if have_up else None, down if have_down else None)
        if have_down and not L.kvzx_decoder_band_import(C.c_void_p(self.h), 1, C.c_void_p(self.halo[1].data_ptr())):
            raise RuntimeError("band_import(1) failed")
        if L.kvzx_decoder_band_finish(C.c_void_p(self.h)) <= 0:
            raise RuntimeError("band_finish failed")
        return self._output()

    def _output(self):
        L = self.lib
        fr = N.OpenHevcFrame()
        L.libOpenHevcGetPictureInfo(C.c_void_p(self.h), C.byref(fr.frameInfo))
        w, h = fr.frameInfo.nWidth, fr.frameInfo.nHeight
        if not self.download:
            return {"width": w, "height": h, "rows": (self.row0 * 64, min(h, (self.row0 + self.nrows) * 64))}
        buf = np.empty(w * h * 3 // 2, dtype=np.uint8)
        fr.pvY = buf.ctypes.data; fr.pvU = buf.ctypes.data + w * h; fr.pvV = buf.ctypes.data + w * h + w * h // 4
        if not L.libOpenHevcGetOutputCpy(C.c_void_p(self.h), 1, C.byref(fr)):
            return None
        return {"width": w, "height": h, "i420": buf, "rows": (self.row0 * 64, min(h, (self.row0 + self.nrows) * 64))}

    def close(self):
        if self.h:
            self.lib.libOpenHevcClose(C.c_void_p(self.h))
            self.h = None


def finish_bands_local(decoders):
    """the split decoder's per-picture protocol for BandDecoders living in ONE process (bands of one picture on one GPU, or a test):
    the halo blocks go from decoder to decoder through a device buffer instead of torch.distributed.  Returns the decoders' pictures."""
    hip = C.CDLL("libamdhip64.so")
    L = decoders[0].lib
    n = int(L.kvzx_decoder_band_halo_bytes(C.c_void_p(decoders[0].h)))
    buf = C.c_void_p()
    if hip.hipMalloc(C.byref(buf), C.c_size_t(n)) != 0:
        raise RuntimeError("hipMalloc failed")
    try:
        for a, b in zip(decoders[:-1], decoders[1:]):                                   # unfiltered boundary rows and records, downwards
            if not L.kvzx_decoder_band_export(C.c_void_p(a.h), 0, buf) or not L.kvzx_decoder_band_import(C.c_void_p(b.h), 0, buf):
                raise RuntimeError("band halo, stage 0")
        for d in decoders:
            if not L.kvzx_decoder_band_deblock(C.c_void_p(d.h)):
                raise RuntimeError("band_deblock failed")
        for a, b in zip(decoders[:-1], decoders[1:]):                                   # the same rows, filtered, back upwards
            if not L.kvzx_decoder_band_export(C.c_void_p(b.h), 1, buf) or not L.kvzx_decoder_band_import(C.c_void_p(a.h), 1, buf):
                raise RuntimeError("band halo, stage 1")
        out = []
        for d in decoders:
            if L.kvzx_decoder_band_finish(C.c_void_p(d.h)) <= 0:
                raise RuntimeError("band_finish failed")
            out.append(d._output())
        return out
    finally:
        hip.hipFree(buf)


def assemble(lib, cfg, parts, write_parameter_sets=None, state=None):
    """rank 0: substreams of all bands in picture order -> one access unit (host only, no GPU)"""
    sizes = [s for p in parts for s in p[0]]
    data = b"".join(p[1] for p in parts)
    poc, qp, nal = parts[0][2], parts[0][3], parts[0][4]
    idr = nal == 19
    if write_parameter_sets is None:                 # parameter sets with every vps-period-th IDR, as the single encoder does;
        state = state if state is not None else {}  # `state`: the caller's own counter (a dict kept with its encoder), not anything keyed by an address
        n = state.get("intra_count", 0)
        vp = cfg.contents.vps_period
        write_parameter_sets = idr and (n == 0 or (vp > 0 and n % vp == 0))
        if idr:
            state["intra_count"] = n + 1
    lib.kvzx_assemble_access_unit.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
    sz = np.array(sizes, dtype=np.uint32)
    src = np.frombuffer(data, dtype=np.uint8) if data else np.zeros(1, np.uint8)
    out = np.empty(len(data) * 2 + 4096, dtype=np.uint8)
    n = C.c_uint32(0)
    ok = lib.kvzx_assemble_access_unit(cfg, int(idr), int(poc), int(bool(write_parameter_sets)), int(qp), src.ctypes.data, sz.ctypes.data, len(sizes),
                                       out.ctypes.data, len(out), C.byref(n))
    if not ok:
        raise RuntimeError("kvzx_assemble_access_unit failed (substream count %d)" % len(sizes))
    return bytes(out[:n.value])
