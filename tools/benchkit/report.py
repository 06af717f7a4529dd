"""bench.py, part 5 of 5 -- the roofline object: the dominant kernel of the timed region against the HBM peak, every kernel's fraction, the
committed counter passes (profiles/)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from . import workloads as W
from .workloads import PERIOD, algorithmic_bytes


def roofline_of(m, steps, me_range, workload_key):
    """dominant kernel = largest share of the timed region: average launch time x launches in the region (the events sample
    every n-th picture, so the launch counts come from the picture types, not from the samples)"""
    kt, cw, ch = m["kt"], m["cw"], m["ch"]
    if not any(v[1] for v in kt.values()):
        return None, {}, {}            # KVAZZUP_BENCH_NOPROF=1: throughput-only run
    n_idr, npic = steps, steps * PERIOD

    def launches(k):
        if k in ("k_intra_analyse", "k_intra_recon", "k_dec_intra", "k_intra_recon<dec>"):
            return n_idr
        if k in ("k_me", "k_inter_recon", "k_inter_signal", "k_dec_inter", "k_inter_recon<dec>", "k_intra_analyse<P>", "k_intra_recon<P>", "k_dec_intra<P>", "k_subpel"):
            return npic - n_idr
        return npic
    kern = [k for k in kt if kt[k][1] > 0 and k.startswith("k_")]
    dom = max(kern, key=lambda k: kt[k][0] / kt[k][1] * launches(k))
    avg_s = kt[dom][0] / kt[dom][1] / 1e3
    ab = algorithmic_bytes(dom, cw, ch, me_range)
    achieved = ab / avg_s / 1e9
    # HBM traffic per launch: from the committed PMC passes of this same command (rocprofv3 --pmc cannot run inside the bench);
    # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950.  null when no pass exists.
    traffic, traffic_src = None, None
    mfma = None
    pmc_kernels = {}
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        try:
            name = "profiles/%s_pmc_traffic_%s.json" % (rnd, workload_key)
            pmc = json.load(open(os.path.join(ROOT, name)))
            traffic = pmc["kernels"][dom]["traffic_bytes"]
            traffic_src = name
            pmc_kernels = pmc["kernels"]
            # matrix-core utilisation of the kernels that use them (same counter passes: SQ_VALU_MFMA_BUSY_CYCLES / (duration x clock x SIMDs))
            mfma = {k: v["mfma_util"] for k, v in pmc["kernels"].items() if v.get("mfma_util")} or None
            break
        except Exception:
            pass
    roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": W.HBM_PEAK_GBS, "peak_source": W.HBM_PEAK_SOURCE, "unit": "GB/s",
            "frac": round(achieved / W.HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": ab, "avg_launch_us": round(avg_s * 1e6, 2)}
    kernels_us = {k: round(v[0] / v[1] * 1e3, 2) for k, v in kt.items() if v[1]}
    share = {k: round(v[0] / v[1] * launches(k) / (m["elapsed"] * 1e3), 4) for k, v in kt.items() if v[1]}
    # every kernel against the HBM roofline (algorithmic bytes of one launch / its average duration)
    per_kernel = {k: round(algorithmic_bytes(k, cw, ch, me_range) / (kt[k][0] / kt[k][1] / 1e3) / 1e9 / W.HBM_PEAK_GBS, 5) for k in kern}
    roof["frac_by_kernel"] = per_kernel
    # the same fractions from the COUNTERS (HBM bytes of a launch in the committed PMC pass / this run's event time / peak) and what a launch moves
    # over what it must: > 1 = re-reads (polling, windows that fall out of L2), < 1 = the "algorithmic" figure counts bytes that L2 serves (overlapping windows)
    def pmc_name(k):
        return "k_deblock" if k == "k_deblock_tile" else k
    tr = {k: pmc_kernels[pmc_name(k)]["traffic_bytes"] for k in kern if pmc_name(k) in pmc_kernels and pmc_kernels[pmc_name(k)].get("traffic_bytes")}
    roof["frac_traffic_by_kernel"] = {k: round(b / (kt[k][0] / kt[k][1] / 1e3) / 1e9 / W.HBM_PEAK_GBS, 5) for k, b in tr.items()} or None
    roof["traffic_over_algorithmic"] = {k: round(b / algorithmic_bytes(k, cw, ch, me_range), 2) for k, b in tr.items()} or None
    roof["frac_traffic"] = (roof["frac_traffic_by_kernel"] or {}).get(dom)
    roof["mfma_util_by_kernel"] = mfma                 # north_star: "MFMA utilisation against gfx950 peak" -- the transforms and Hadamard sums are small products between LDS phases
    return roof, kernels_us, share
