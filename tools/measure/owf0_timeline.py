#!/usr/bin/env python3
"""tools/measure/owf0_timeline.py [1080p|4k] [fps] -- where a picture's time goes at uvgComm's DEFAULT settings (OWF 0, four OpenHEVC threads of type "Slice",
no custom parameter, host I420 in and out, camera-paced source): the host timeline (KVAZZUP_AMD_TIMELINE) of every picture from the filter's copy into the
kvz_picture to the decoded picture leaving OpenHEVCFilter', as median time between consecutive stages.  Run on the GPU box."""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
TL = "/tmp/owf0_tl.txt"
os.environ["KVAZZUP_AMD_TIMELINE"] = TL

from tools.benchkit.workloads import WORKLOADS, PERIOD, DeviceClip, stream_seed          # noqa: E402
from tools.benchkit.host import StreamRanks                                                # noqa: E402
from kvazzup_amd.pipeline import Pipeline                                                  # noqa: E402

STAGES = ["feed0", "copied", "up0", "sub1", "gpudone", "arith", "rec0", "rec1", "col1", "dec0", "dlaunch0", "dlaunch1", "dcomplete", "dec1", "out0", "out1"]


def main():
    wl = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "1080p"]
    fps = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
    if "--child" not in sys.argv:
        # (the timeline is written when the process that recorded it ends: the run is a child process, this one reads its file)
        import subprocess
        if os.path.exists(TL):
            os.remove(TL)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), sys.argv[1] if len(sys.argv) > 1 else "1080p", str(fps), "--child"], capture_output=True, text=True)
        sys.stderr.write(out.stderr[-2000:])
        lat = [l for l in out.stdout.split("\n") if l.startswith("LAT ")]
        report(wl, fps, [float(v) for v in lat[0].split()[1:]] if lat else [0, 0])
        return
    ranks = StreamRanks(1, 0)
    w, h = wl["w"], wl["h"]
    clip = DeviceClip(ranks.lib, ranks.dev_index, stream_seed(wl["cfg_index"], 0), w, h, PERIOD)
    host = [clip.host(t) for t in range(PERIOD)]
    pl = Pipeline(w, h, settings={"video/QP": 32, "video/Intra": PERIOD, "video/VPS": 1, "video/OWF": 0, "video/OPENHEVC_threads": 4, "video/OH_parallelization": "Slice"},
                  custom=(), loopback=True, keep_outputs=False)
    n = 2 * PERIOD
    t_next = time.perf_counter()
    for t in range(n):
        now = time.perf_counter()
        if now < t_next:
            time.sleep(t_next - now)
        assert pl.push_host_paced(host[t % PERIOD], 9, 120000)
        t_next += 1.0 / fps
    pl.flush()
    assert pl.wait(n, 120000)
    enc, tot = pl.latency_us(0), pl.latency_us(1)
    pl.close(); clip.close(); ranks.close()
    print("LAT %.1f %.1f" % (sorted(enc[PERIOD:])[len(enc[PERIOD:]) // 2], sorted(tot[PERIOD:])[len(tot[PERIOD:]) // 2]))


def report(wl, fps, lat):
    ev = collections.defaultdict(dict)
    for line in open(TL):
        ns, tid, what, pic = line.split()
        if what in STAGES:
            d = ev[int(pic)]
            if what == "dec0":
                d[what] = int(ns)                     # (one per NAL unit: the LAST one of the picture is its slice)
            elif what == "dec1":
                d[what] = int(ns)
            else:
                d.setdefault(what, int(ns))
    pics = [p for p in sorted(ev) if p >= PERIOD + 1 and p % PERIOD != 0 and all(s in ev[p] for s in ("feed0", "col1", "out1"))]      # P pictures of the second period
    print("%s at %g pictures/s, OWF 0, Slice/4, no custom parameter: %d P pictures; encoding delay p50 %.0f us, total delay p50 %.0f us" % (
        wl["name"], fps, len(pics), lat[0], lat[1]))
    have = [s for s in STAGES if all(s in ev[p] for p in pics)]
    print("median microseconds between consecutive stages (P pictures):")
    for a, b in zip(have[:-1], have[1:]):
        d = sorted((ev[p][b] - ev[p][a]) / 1e3 for p in pics)
        print("  %-9s -> %-9s %8.1f   (p90 %8.1f)" % (a, b, d[len(d) // 2], d[len(d) * 9 // 10]))
    # the rows of one picture's parse (row0 / row1 records carry the substream's index; the picture is the one whose dec0 .. dlaunch0 window they fall in)
    rows = []
    for line in open(TL):
        ns, tid, what, pic = line.split()
        if what in ("row0", "row1"):
            rows.append((int(ns), int(tid), what, int(pic)))
    if rows and pics:
        p = pics[len(pics) // 2]
        a, b = ev[p]["dec0"], ev[p]["dlaunch0"]
        st = {r[3]: (r[0], r[1]) for r in rows if r[2] == "row0" and a <= r[0] <= b}
        en = {r[3]: r[0] for r in rows if r[2] == "row1" and a <= r[0] <= b}
        print("rows of picture %d's parse (us from dec0; %d threads took part):" % (p, len({v[1] for v in st.values()})))
        print("  " + "  ".join("%d:%.0f-%.0f" % (k, (st[k][0] - a) / 1e3, (en.get(k, b) - a) / 1e3) for k in sorted(st)))
    d = sorted((ev[p]["out1"] - ev[p]["feed0"]) / 1e3 for p in pics)
    print("  feed0 -> out1 %8.1f (p90 %.1f)" % (d[len(d) // 2], d[len(d) * 9 // 10]))
    print("stages: feed0 filter has the input | copied memcpy into the kvz_picture done | up0 encoder_encode entered | sub1 upload + kernels queued | gpudone tokens on the host |"
          " arith substreams coded | rec0..rec1 wait for the reconstruction's copy | col1 access unit ready | dec0 decoder filter has the slice NAL | dlaunch0 parsed |"
          " dlaunch1 upload + kernels queued | dcomplete picture in host memory | dec1 libOpenHevcDecode returned | out0..out1 row copy out of the frame")


if __name__ == "__main__":
    main()
