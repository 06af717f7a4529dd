"""bench.py, part 1 of 5 -- the named workloads (BASELINE.json configs), the synthetic clip in device memory and the algorithmic bytes of every kernel
(SURVEY.md 8(d))."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PERIOD = 64                    # pictures per step = the intra period of the named workloads
CLIP_FRAMES = 128              # the named clips are 128 pictures long; longer runs cycle them (the wrap falls on an IDR)
WORKLOADS = {
    # BASELINE.json configs[1]: 1080p, preset=ultrafast, intra period 64, encode + decode on one GPU
    "1080p": dict(w=1920, h=1080, name="1080p-yuv420-ultrafast-p64-qp32-encode+decode", cfg_index=2),
    # configs[2]: 4K encode (decode is run too; reported in the same fps)
    "4k": dict(w=3840, h=2160, name="2160p-yuv420-ultrafast-p64-qp32-encode+decode", cfg_index=3),
    "720p": dict(w=1280, h=720, name="720p-yuv420-ultrafast-p64-qp32-encode+decode", cfg_index=2),
    # configs[4]: ONE 8K stream, its 8 tile rows split over the ranks (strong scaling; see tilesplit_main)
    "8k-tilesplit": dict(w=7680, h=4320, name="4320p-yuv420-ultrafast-p64-qp32-encode-tile-row-split", cfg_index=5),
}
# custom parameters of the host-boundary legs (uvgComm's INI list "parameters", kvazaarfilter.cpp:351-371): the reconstruction is not downloaded
# (uvgComm frees it unread, :476) and encoder_encode(NULL) only returns pictures that are finished (the loop at :440-448 then keeps video/OWF
# pictures in flight instead of emptying the pipeline after every picture) -- INTEGRATION.md
HOST_CUSTOM = (("recon-output", "0"), ("null-input", "poll"))
HBM_PEAK_GBS = 8000.0          # replaced by the device's own figure in main(); this is the guide's (MI355X_MICROARCH.md) and the fallback
HBM_PEAK_SOURCE = "MI355X_MICROARCH.md (the runtime reported no memory clock / bus width)"          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def algorithmic_bytes(kernel, cw, ch, me_range):
    """Compulsory bytes of ONE launch, P = coded luma samples."""
    P = cw * ch
    if kernel == "k_me":                          # "CTU pixels once + search-window pixels once per CTU, (64+2r)^2"
        return (P // 4096) * (4096 + (64 + 2 * me_range) ** 2)
    if kernel in ("k_inter_recon", "k_dec_inter", "k_inter_recon<dec>"):
        return int(4.5 * P) if kernel == "k_inter_recon" else int(3.0 * P)
    if kernel in ("k_intra_recon<P>", "k_dec_intra<P>", "k_intra_analyse<P>"):      # a P picture's few intra units: priced like the whole picture's pass they are a part of
        return int(3.0 * P) if kernel == "k_intra_recon<P>" else (int(1.5 * P) if kernel == "k_dec_intra<P>" else P)
    if kernel in ("k_intra_recon", "k_dec_intra", "k_intra_recon<dec>"):  # source in + reconstruction out (+ the level words, counted with k_tokenize)
        return int(3.0 * P) if kernel == "k_intra_recon" else int(1.5 * P)
    if kernel == "k_intra_analyse":
        return P
    if kernel in ("k_deblock", "k_dec_deblock"):
        return int(3.0 * P)
    if kernel == "k_tokenize":                    # every level of the picture once (int16) + the per-8x8 CU records; tokens out not counted
        return int(3.0 * P) + (P // 64) * 11
    if kernel == "k_tok_compact":                 # the piece table of every CTU ([16 units][17 pieces] {offset, length}); tokens not counted
        return (P // 4096) * 16 * 17 * 8
    if kernel in ("k_sao", "k_dec_sao", "k_sao<dec>"):          # deblocked picture in, filtered picture out (+ the source picture for the statistics)
        return int(4.5 * P) if kernel == "k_sao" else int(3.0 * P)
    if kernel == "k_pad_input":
        return int(3.0 * P)
    if kernel == "k_inter_signal":
        return (P // 64) * 16
    return P


def stream_seed(cfg_index, rank):
    """every rank codes its own synthetic stream (uvgx-synth-v1: seed = 0x5EED0000 + configuration, shifted per stream)"""
    return 0x5EED0000 + cfg_index + 16 * rank


class DeviceClip:
    """the synthetic clip in device memory (kvzx_harness_*: generated on the GPU, no tensor library)"""

    def __init__(self, lib, dev_index, seed, w, h, frames, kind=0):
        import ctypes as C
        lib.kvzx_harness_alloc.restype = C.c_void_p
        lib.kvzx_harness_alloc.argtypes = [C.c_int, C.c_size_t]
        lib.kvzx_harness_free.argtypes = [C.c_void_p]
        lib.kvzx_harness_synth_frame.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int]
        lib.kvzx_harness_download.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        self.lib, self.dev, self.w, self.h, self.n = lib, dev_index, w, h, w * h * 3 // 2
        self.ptr = []
        for t in range(frames):
            p = lib.kvzx_harness_alloc(dev_index, self.n)
            if not p or not lib.kvzx_harness_synth_frame(p, kind, seed & 0xFFFFFFFF, w, h, t):
                raise RuntimeError("device clip: allocation or synthesis failed")
            self.ptr.append(p)
        if not lib.kvzx_harness_sync(dev_index):
            raise RuntimeError("device clip: synthesis failed")

    def host(self, t):
        import numpy as np
        a = np.empty(self.n, dtype=np.uint8)
        if not self.lib.kvzx_harness_download(a.ctypes.data, self.ptr[t], self.n):
            raise RuntimeError("device clip: download failed")
        return a

    def close(self):
        for p in self.ptr:
            self.lib.kvzx_harness_free(p)
        self.ptr = []
