"""bench.py, part 3 of 5 -- the stream workloads: K intra periods of one stream through KvazaarFilter' -> WireAdapter -> OpenHEVCFilter' (run_stream),
and K such chains at once in this process on one GPU (multi_stream)."""
import json
import resource
import time
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from . import workloads as W
from .workloads import PERIOD, CLIP_FRAMES, DeviceClip, algorithmic_bytes, stream_seed
from .host import cpu_budget, _throttled_us, _thread_cpu


def run_stream(args, wl, steps, warmup, ranks, rank, world, quality, host_io=False, extra_custom=(), extra_settings=None, kind=0, bare=False, repeats=None):
    """K = steps intra periods of one stream through KvazaarFilter' -> WireAdapter -> OpenHEVCFilter' on this rank's GPU, timed
    args.repeats times (median of 3 runs); every repetition starts and ends on an empty, flushed pipeline.
    host_io: the reference's own boundary -- pictures enter as HOST I420 through kvz_api->encoder_encode(kvz_picture*) (the filter's
    memcpy into the kvz_picture included, kvazaarfilter.cpp:410-438) and leave through libOpenHevcGetOutput + the filter's row copy into
    host memory (openhevcfilter.cpp:192-239): PCIe both ways inside the timed region.
    kind: the synthetic clip (0 moving objects, 1 flat, 2 noise -- SURVEY 8(d)'s bounds).  bare: NO custom parameter at all reaches the encoder (uvgComm's
    INI list "parameters" empty: the workload's search range and device are the defaults anyway) -- what an unmodified uvgComm.ini gets.
    Returns a dict of measurements.  sync(value=None) = barrier / max over ranks."""
    from kvazzup_amd.pipeline import Pipeline
    import ctypes as C
    dev_index, sync = ranks.dev_index, ranks.sync
    w, h = wl["w"], wl["h"]
    # pictures parsed concurrently (video/OPENHEVC_threads): the ring has to cover the parse of an intra picture -- ~10 ms on one core at
    # 4K, where twelve pictures pass in 6 ms (measured: 2090 frames/s with 12, 2390 with 24; 1080p, 3.5 ms per intra picture: 5900-6170 / 6500-6700)
    D = max(1, args.decoder_frame_threads or 32)      # (round 3, end: 32 against 24: 8 430-8 580 against 8 240-8 350 frames/s at 1080p, 3 150-3 170 against 2 910-3 100 at 4K, two alternating runs each)
    budget = float(os.environ.get("KVAZZUP_BENCH_CPU_BUDGET", 0)) or cpu_budget(world)
    if budget < 13.0:                                # not enough host CPU for the full thread complement: shrink the pools
        D = max(1, min(D, int(budget * 0.45 + 0.5)))
        os.environ.setdefault("KVAZZUP_AMD_ENTROPY_THREADS", str(max(2, min(16, int(budget * 0.4)))))
        os.environ.setdefault("KVAZZUP_AMD_PARSE_THREADS", str(max(1, min(16, int(budget * 0.4)))))       # (row-parallel parser of the synchronous decoder)
    seed = stream_seed(wl["cfg_index"], rank)
    # synthetic clip generated directly in HBM (inputs resident before the timed region)
    nclip = CLIP_FRAMES
    dclip = DeviceClip(ranks.lib, dev_index, seed, w, h, nclip, kind=kind)
    clip = dclip.ptr
    host_clip = [dclip.host(t) for t in range(nclip)] if host_io else None       # pageable host memory, as a camera filter's frames are

    def device_sync():
        if not ranks.lib.kvzx_harness_sync(dev_index):
            raise RuntimeError("device synchronisation failed")

    def make(keep, download):
        st = {"video/QP": 32, "video/Intra": PERIOD, "video/VPS": 1, "uvgx/gpu": dev_index, "uvgx/decoderDownload": int(download),
              "video/OWF": args.owf, "video/OPENHEVC_threads": D, "video/OH_parallelization": "Frame" if D > 1 else "Slice"}
        st.update(extra_settings or {})
        if os.environ.get("KVAZZUP_BENCH_COPY_THREADS"):
            st["uvgx/copyThreads"] = os.environ["KVAZZUP_BENCH_COPY_THREADS"]       # (measurement aid: helpers of the filters' picture copies, default 4)
        if bare:
            if args.me_range != 16 or dev_index != 0:
                raise RuntimeError("the bare leg needs the defaults (me-range 16, device 0)")
            return Pipeline(w, h, settings=st, custom=(), loopback=True, keep_outputs=keep)
        return Pipeline(w, h, settings=st,
                        custom=(("me-range", args.me_range), ("gpu", dev_index)) + ((("sao", "full"),) if args.sao else ()) + ((("me-early-termination", "off"),) if args.full_search else ()) + ((("intra-satd", "0"),) if args.intra_sad else ()) + ((("gpu-entropy", "1"),) if args.gpu_entropy else ()) + ((("subme", str(args.subme)),) if args.subme else ()) + tuple(tuple(kv.split("=", 1)) for kv in args.custom) + tuple(extra_custom),
                        loopback=True, keep_outputs=keep)

    # source -> KvazaarFilter -> WireAdapter -> OpenHEVCFilter -> sink, one thread per filter (csrc/filters.hip)
    pl = make(False, host_io)
    lib = pl.lib
    enc_h, dec_h = pl.encoder_handle(), pl.decoder_handle()
    cw, ch = C.c_int(), C.c_int()
    lib.kvzx_encoder_coded_size(enc_h, C.byref(cw), C.byref(ch))
    cw, ch = cw.value, ch.value

    def run(npic):
        """push `npic` more pictures, flush the pipeline, wait until every one of them has been decoded.  The feeder keeps the
        encoder filter's input buffer short of its overflow threshold (a uvgComm filter drops inputs at 10 buffered, filter.cpp:151-222)."""
        last = pl.pushed + npic
        while pl.pushed < last:
            ok = pl.push_host_paced(host_clip[pl.pushed % nclip], 6, 120000) if host_io else pl.push_device_paced(clip[pl.pushed % nclip], 6, 120000)
            if not ok:
                raise RuntimeError("pipeline stalled")
        pl.flush()
        if not pl.wait(last, 120000):
            raise RuntimeError("pipeline did not deliver %d pictures: %r" % (last, pl.stats()))

    def times(reset):
        ms = (C.c_double * 16)()
        n = (C.c_uint64 * 16)()
        out = {}
        k = lib.kvzx_encoder_kernel_times(enc_h, ms, n, int(reset))
        for i in range(k):
            out[lib.kvzx_encoder_kernel_name(i).decode()] = (ms[i], n[i])
        k = lib.kvzx_decoder_kernel_times(dec_h, ms, n, int(reset))
        for i in range(k):
            name = lib.kvzx_decoder_kernel_name(i).decode()
            a = out.get(name, (0.0, 0))
            out[name] = (a[0] + ms[i], a[1] + n[i])
        return out

    run(max(1, warmup) * PERIOD if warmup > 0 else 8)         # warm-up: whole periods, so that the first timed picture is an IDR (8 pictures when --warmup 0: the pipeline must at least be built)
    if pl.pushed % PERIOD:
        run(PERIOD - pl.pushed % PERIOD)
    # HIP events around every kernel of every 8th picture of the timed region (IDR pictures fall on multiples of 8)
    prof = 0 if os.environ.get("KVAZZUP_BENCH_NOPROF") else args.profile_every
    lib.kvzx_encoder_set_profiling(enc_h, prof)
    lib.kvzx_decoder_set_profiling(dec_h, prof)
    busy0 = pl.busy_ms()
    st0 = pl.stats()
    times(True)
    _thr0 = _thread_cpu() if os.environ.get("KVAZZUP_BENCH_THREADS") else None
    _sampler = None
    if os.environ.get("CPU_SAMPLER_REGION"):                  # tools/cpu_sampler.c preloaded: sample the timed regions only
        _sampler = C.CDLL(None)
    reps = []
    for rep in range(max(1, repeats or args.repeats)):
        device_sync()                                          # the pipeline is empty: everything pushed so far has been decoded
        sync()
        cpu0 = time.process_time()
        thr0 = _throttled_us()
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        if _sampler is not None:
            _sampler.cpu_sampler_begin()
        t0 = time.perf_counter()
        run(steps * PERIOD)
        device_sync()
        sync()
        el = time.perf_counter() - t0
        if _sampler is not None:
            _sampler.cpu_sampler_end()
        # throttled: summed over the job's threads: > 0 means the CPU quota, not the GPU, set the pace for a while; host cores: CPU
        # seconds of all threads of this rank per second of the timed region
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        reps.append({"elapsed": sync(el), "throttled_ms": (_throttled_us() - thr0) / 1e3, "host_cores": (time.process_time() - cpu0) / el,
                     "minflt": (ru1.ru_minflt - ru0.ru_minflt) / (steps * PERIOD), "sys_share": (ru1.ru_stime - ru0.ru_stime) / max(1e-9, (ru1.ru_stime - ru0.ru_stime) + (ru1.ru_utime - ru0.ru_utime))})
    if _thr0 is not None:                                       # KVAZZUP_BENCH_THREADS=1: CPU time per thread over the timed regions (stderr)
        _thr1 = _thread_cpu()
        tot = sum(r["elapsed"] for r in reps)
        rows = sorted(((_thr1[t][1] - _thr0.get(t, ("", 0.0))[1], t, _thr1[t][0]) for t in _thr1), reverse=True)
        for dt, t, name in rows[:40]:
            if dt > 0:
                print("thread %7d %-16s %.3f s (%.2f cores)" % (t, name, dt, dt / tot), file=sys.stderr)
    med = sorted(reps, key=lambda r: r["elapsed"])[len(reps) // 2]
    elapsed, throttled_ms, host_cores = med["elapsed"], med["throttled_ms"], med["host_cores"]
    kt = times(False)
    npic = steps * PERIOD
    nall = npic * len(reps)
    busy = [round((b - a) / nall, 4) for a, b in zip(busy0, pl.busy_ms())]
    st = pl.stats()
    nbytes = st["encoded_bytes"] - st0["encoded_bytes"]
    if st["decoded_pictures"] != pl.pushed or st["dropped"] or st["encoded_pictures"] - st0["encoded_pictures"] != nall:
        raise RuntimeError("pipeline lost pictures: %r" % (st,))
    pl.close()

    out = {"elapsed": elapsed, "pictures": npic, "cw": cw, "ch": ch, "kt": kt, "busy": busy, "bytes_per_picture": nbytes / nall, "D": D,
           "host_cores": host_cores, "budget": budget, "throttled_ms": throttled_ms, "psnr_y": None, "minflt": med["minflt"], "sys_share": med["sys_share"],
           "runs_fps": [round(world * npic / r["elapsed"], 1) for r in reps]}
    if quality:
        # Quality of what was just timed (untimed pass): one intra period through a second pipeline with the decoded pictures
        # downloaded; luma PSNR of the decoder's output against the source, mean over the period's 64 pictures.
        import numpy as np
        q = make(True, True)
        for t in range(PERIOD):
            if not q.push_device_paced(clip[t], 6, 120000):
                raise RuntimeError("quality pass stalled")
        q.flush()
        if not q.wait(PERIOD, 120000):
            raise RuntimeError("quality pass did not deliver")
        ps = []
        for t in range(PERIOD):
            d = q.pop_decoded()
            src = dclip.host(t)[:w * h].astype(np.int32)
            mse = float(((src - d["i420"][:w * h].astype(np.int32)) ** 2).mean(dtype=np.float64))
            ps.append(99.0 if mse == 0 else 10.0 * float(np.log10(255.0 * 255.0 / mse)))
        q.close()
        out["psnr_y"] = round(sum(ps) / len(ps), 3)
    dclip.close()
    return out


def latency_run(args, wl, ranks, settings, custom, source_fps, npic, drop_tail):
    """Per-picture delays at a PACED source (a camera: one picture every 1 / source_fps s, host I420 in, decoded I420 out into host memory -- the
    reference's boundary): encoding delay (picture handed to KvazaarFilter' -> access unit sent on; what kvazaarfilter.cpp:478-479 books) and total delay
    (-> decoded picture out of OpenHEVCFilter'; displayfilter.cpp:113-115), microseconds.  The last `drop_tail` pictures only leave with the flush that ends
    the run (video/OWF, frame threads) and are not counted.  Returns {'encoding_delay_us': {p50, p99, mean, max}, 'total_delay_us': {...}, ...}."""
    import numpy as np
    from kvazzup_amd.pipeline import Pipeline
    w, h = wl["w"], wl["h"]
    dclip = DeviceClip(ranks.lib, ranks.dev_index, stream_seed(wl["cfg_index"], 0), w, h, PERIOD)
    host_clip = [dclip.host(t) for t in range(PERIOD)]
    st = {"video/QP": 32, "video/Intra": PERIOD, "video/VPS": 1, "uvgx/gpu": ranks.dev_index, "uvgx/decoderDownload": 1}
    st.update(settings)
    pl = Pipeline(w, h, settings=st, custom=custom, loopback=True, keep_outputs=False)
    try:
        for t in range(PERIOD):                              # warm-up: one period at full speed, flushed
            if not pl.push_host_paced(host_clip[t], 6, 120000):
                raise RuntimeError("latency leg: pipeline stalled")
        pl.flush()
        if not pl.wait(PERIOD, 120000):
            raise RuntimeError("latency leg: warm-up did not deliver")
        pl.latency_us(0), pl.latency_us(1)                   # (reset)
        period = 1.0 / source_fps
        t_next = time.perf_counter()
        late = 0
        for t in range(npic):
            now = time.perf_counter()
            if now < t_next:
                time.sleep(t_next - now)
            elif now - t_next > period:
                late += 1
            if not pl.push_host_paced(host_clip[t % PERIOD], 9, 120000):
                raise RuntimeError("latency leg: pipeline stalled")
            t_next += period
        time.sleep(2 * period)
        pl.flush()
        if not pl.wait(PERIOD + npic, 120000):
            raise RuntimeError("latency leg: pipeline did not deliver: %r" % (pl.stats(),))
        enc, tot = pl.latency_us(0), pl.latency_us(1)
        stt = pl.stats()
        if stt["dropped"]:
            raise RuntimeError("latency leg: the filters dropped pictures: %r" % (stt,))
    finally:
        pl.close()
        dclip.close()
    keep = max(1, npic - drop_tail)

    def summary(a):
        a = np.asarray(a[:keep], dtype=np.float64)
        if not len(a):
            return None
        return {"p50": round(float(np.percentile(a, 50)), 1), "p99": round(float(np.percentile(a, 99)), 1), "mean": round(float(a.mean()), 1), "max": round(float(a.max()), 1)}
    return {"encoding_delay_us": summary(enc), "total_delay_us": summary(tot), "pictures": keep, "source_fps": source_fps, "source_late": late}


def multi_stream(args, wl, K, steps, ranks):
    """K independent streams (a K-party call) on ONE GPU at the same time, each through its own KvazaarFilter' -> WireAdapter -> OpenHEVCFilter' chain
    (own encoder, decoder, HIP streams, host threads): what the GPU sustains when it is not waiting for one stream's chain of dependent kernels.
    The host side is divided between the streams (decoder frame threads, coder threads).  Returns the aggregate frames/s and each stream's."""
    import threading
    from kvazzup_amd.pipeline import Pipeline
    w, h = wl["w"], wl["h"]
    budget = float(os.environ.get("KVAZZUP_BENCH_CPU_BUDGET", 0)) or cpu_budget(1)
    # the parse ring has to cover an intra picture's parse (3.5 ms at 1080p): 32 pictures at one stream's full rate, 32 / K at a K-th of it
    D = int(os.environ.get("KVAZZUP_BENCH_MULTI_D", 0)) or max(2, min(32, 32 // K))
    threads = max(2, min(16, int(budget * 0.5 / K)))
    clips = [DeviceClip(ranks.lib, ranks.dev_index, stream_seed(wl["cfg_index"], k), w, h, PERIOD) for k in range(K)]
    pls = [Pipeline(w, h, settings={"video/QP": 32, "video/Intra": PERIOD, "video/VPS": 1, "uvgx/gpu": ranks.dev_index, "uvgx/decoderDownload": 0, "video/kvzThreads": threads,
                                    "video/OWF": args.owf, "video/OPENHEVC_threads": D, "video/OH_parallelization": "Frame"},
                    custom=(("me-range", args.me_range), ("gpu", ranks.dev_index), ("input-hold", "1")), loopback=True, keep_outputs=False) for _ in range(K)]
    gate = threading.Barrier(K + 1)
    elapsed = [0.0] * K

    def body(k):
        pl, clip = pls[k], clips[k].ptr

        def run(npic):
            last = pl.pushed + npic
            while pl.pushed < last:
                if not pl.push_device_paced(clip[pl.pushed % PERIOD], 6, 120000):
                    raise RuntimeError("pipeline stalled")
            pl.flush()
            if not pl.wait(last, 120000):
                raise RuntimeError("pipeline did not deliver")
        run(PERIOD)                                     # warm-up: one period
        gate.wait()
        t0 = time.perf_counter()
        run(steps * PERIOD)
        elapsed[k] = time.perf_counter() - t0
        gate.wait()

    import ctypes as C
    lib = ranks.lib
    lib.kvzx_batch_stats.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]
    lib.kvzx_batch_kernel_name.restype = C.c_char_p
    ths = [threading.Thread(target=body, args=(k,)) for k in range(K)]
    for t in ths:
        t.start()
    gate.wait()
    if not os.environ.get("KVAZZUP_BENCH_NOPROF"):
        for pl in pls:
            lib.kvzx_decoder_set_profiling(pl.decoder_handle(), args.profile_every)
    lib.kvzx_batch_stats(ranks.dev_index, None, None, None, None, None, None, 1)
    cpu0, t0 = time.process_time(), time.perf_counter()
    gate.wait()
    wall = time.perf_counter() - t0
    cores = (time.process_time() - cpu0) / wall
    nb, npics = C.c_uint64(), C.c_uint64()
    sizes, bms, bln, bfr = (C.c_uint64 * 9)(), (C.c_double * 4)(), (C.c_uint64 * 4)(), (C.c_uint64 * 4)()
    nk = lib.kvzx_batch_stats(ranks.dev_index, C.byref(nb), C.byref(npics), sizes, bms, bln, bfr, 0)
    # the decoders' submission layer (csrc/batch.h): pictures per launch, and the batched kernels against the HBM roofline -- algorithmic bytes of
    # the pictures in a launch over the launch's duration, beside the single-picture kernels' fractions in `roofline.frac_by_kernel`
    cw, ch = (w + 63) // 64 * 64, (h + 63) // 64 * 64
    batched = {}
    for i in range(nk):
        if bln[i]:
            name = lib.kvzx_batch_kernel_name(i).decode()
            us, per = bms[i] / bln[i] * 1e3, bfr[i] / bln[i]
            single = {"k_dec_inter_n": "k_dec_inter", "k_dec_intra_n": "k_dec_intra", "k_dec_deblock_n": "k_dec_deblock", "k_dec_sao_n": "k_dec_sao"}[name]
            batched[name] = {"avg_launch_us": round(us, 2), "pictures_per_launch": round(per, 2), "us_per_picture": round(us / per, 2),
                             "hbm_frac": round(algorithmic_bytes(single, cw, ch, args.me_range) * per / (us * 1e-6) / 1e9 / W.HBM_PEAK_GBS, 5)}
    for t in ths:
        t.join()
    for pl in pls:
        st = pl.stats()
        if st["decoded_pictures"] != pl.pushed or st["dropped"]:
            raise RuntimeError("a stream lost pictures: %r" % (st,))
        pl.close()
    for c in clips:
        c.close()
    npic = steps * PERIOD
    return {"streams": K, "value": round(K * npic / max(elapsed), 1), "unit": "frames/s (all streams together)", "per_stream": [round(npic / e, 1) for e in elapsed],
            "steps_per_stream": steps, "decoder_frame_threads_per_stream": D, "coder_threads_per_stream": threads, "host_cpu_cores_busy": round(cores, 2),
            "decoder_batches": {"launches": nb.value, "pictures": npics.value, "pictures_per_batch": round(npics.value / max(1, nb.value), 3),
                                "batches_by_size": {str(n): sizes[n] for n in range(1, 9) if sizes[n]}, "kernels": batched}}
