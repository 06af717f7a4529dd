"""bench.py's roofline object on CPU: the dominant kernel's algorithmic bytes are SURVEY.md 8(d)'s (for the motion search: "CTU pixels once + search-window
pixels once per CTU, (64 + 2r)^2"), `frac` is those bytes over the measured launch time over the peak, and the counter-based fields come from the committed
PMC passes (profiles/) -- checked with hand-made measurements, no GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tools.benchkit import workloads as W                      # noqa: E402
from tools.benchkit.report import roofline_of                  # noqa: E402
from tools.benchkit.workloads import algorithmic_bytes, PERIOD  # noqa: E402


def test_motion_search_is_priced_per_ctu_as_the_survey_says():
    # 1920x1088 coded: 510 CTUs x (4096 + 96^2) = 3.25 B per coded luma sample (VERDICT r5 recomputed exactly this); 2160p: 2 040 CTUs
    assert algorithmic_bytes("k_me", 1920, 1088, 16) == 510 * (4096 + 96 * 96) == 6789120
    assert algorithmic_bytes("k_me", 3840, 2176, 16) == 2040 * (4096 + 96 * 96) == 27156480
    assert algorithmic_bytes("k_me", 1920, 1088, 32) == 510 * (4096 + 128 * 128)
    P = 1920 * 1088
    assert algorithmic_bytes("k_inter_recon", 1920, 1088, 16) == int(4.5 * P) and algorithmic_bytes("k_dec_inter", 1920, 1088, 16) == int(3.0 * P)
    assert algorithmic_bytes("k_intra_recon", 1920, 1088, 16) == int(3.0 * P) and algorithmic_bytes("k_dec_intra", 1920, 1088, 16) == int(1.5 * P)
    assert algorithmic_bytes("k_deblock", 1920, 1088, 16) == int(3.0 * P)


def test_roofline_object_from_hand_made_measurements():
    steps = 4
    npic = steps * PERIOD
    # (total ms, launches sampled): k_me 40 us per launch on 63 of 64 pictures dominates a 7 ms step
    kt = {"k_me": (0.040 * 30, 30), "k_inter_recon": (0.025 * 30, 30), "k_deblock": (0.015 * 32, 32), "k_intra_recon": (0.700 * 2, 2), "host_cabac_parse": (1.0 * 32, 32)}
    m = {"kt": kt, "cw": 1920, "ch": 1088, "elapsed": npic * 0.00011}
    W.HBM_PEAK_GBS = 8000.0
    roof, kernels_us, share = roofline_of(m, steps, 16, "1080p")
    assert roof["kernel"] == "k_me" and roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert roof["algorithmic_bytes_per_launch"] == 6789120 and abs(roof["avg_launch_us"] - 40.0) < 1e-6
    assert abs(roof["achieved"] - 6789120 / 40e-6 / 1e9) < 0.01 and abs(roof["frac"] - roof["achieved"] / 8000.0) < 1e-5
    assert abs(kernels_us["k_me"] - 40.0) < 1e-6 and 0 < share["k_me"] < 1
    assert "host_cabac_parse" not in roof["frac_by_kernel"] and set(roof["frac_by_kernel"]) == {"k_me", "k_inter_recon", "k_deblock", "k_intra_recon"}
    # the counter-based twins come from the newest committed PMC pass of the workload
    assert roof["traffic_source"] and roof["traffic_source"].startswith("profiles/r0") and roof["traffic"] > 0
    assert abs(roof["traffic_over_algorithmic"]["k_me"] - roof["traffic"] / 6789120) < 0.006
    assert abs(roof["frac_traffic"] - roof["traffic"] / 40e-6 / 1e9 / 8000.0) < 1e-4
    assert roof["traffic_over_algorithmic"]["k_intra_recon"] > 2.0          # the chain's polls: what the line is there to show
    # a throughput-only run (no kernel timing) yields no object rather than a wrong one
    assert roofline_of({"kt": {"k_me": (0.0, 0)}, "cw": 1920, "ch": 1088, "elapsed": 1.0}, steps, 16, "1080p") == (None, {}, {})
