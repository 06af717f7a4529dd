"""Per-kernel HBM traffic from the three counter passes of tools/pmc_traffic.sh -> JSON on stdout (copied to profiles/).
FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes
(MI355X_MICROARCH.md, HBM section), so it is doubled; WRITE_SIZE is taken as reported (uncalibrated)."""
import csv
import glob
import json
import os
import re
import sys

root, workload = sys.argv[1], sys.argv[2]


def short(name):
    m = re.search(r"kvzx::(k_\w+)(<[^>]*>)?", name)
    if not m:
        return None
    k = m.group(1)
    return {"k_deblock_tile": "k_deblock"}.get(k, k)     # (bench.py's name for the encoder's deblocking kernel; k_tokenize<true> is the all-components variant of the same kernel, pictures beyond 4K)


def per_kernel(pass_name, counter):
    acc = {}
    for path in glob.glob(os.path.join(root, pass_name, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r.get("Counter_Name") != counter:
                continue
            k = short(r["Kernel_Name"])
            if k:
                a = acc.setdefault(k, [0.0, 0])
                a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc


def durations(pass_name):
    """average dispatch duration (ns) per kernel from the kernel trace of the same pass"""
    acc = {}
    for path in glob.glob(os.path.join(root, pass_name, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if k:
                a = acc.setdefault(k, [0.0, 0])
                a[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); a[1] += 1
    return {k: a[0] / a[1] for k, a in acc.items() if a[1]}


fetch, write, mfma = per_kernel("fetch", "FETCH_SIZE"), per_kernel("write", "WRITE_SIZE"), per_kernel("mfma", "SQ_VALU_MFMA_BUSY_CYCLES")
mfma_dur = durations("mfma")
SIMDS, CLOCK_GHZ = 1024, 2.4       # 256 CUs x 4 SIMDs; shader clock (MI355X_MICROARCH.md)
kernels = {}
for k in sorted(set(fetch) | set(write) | set(mfma)):
    f = fetch.get(k, [0.0, 0]); w = write.get(k, [0.0, 0])
    fk = f[0] / f[1] if f[1] else 0.0
    wk = w[0] / w[1] if w[1] else 0.0
    e = {"fetch_size_kb_reported": round(fk, 1), "write_size_kb_reported": round(wk, 1), "launches": [f[1], w[1]],
         "traffic_bytes": int(round((2 * fk + wk) * 1024))}
    if k in mfma and mfma[k][1]:
        busy = mfma[k][0] / mfma[k][1]
        e["mfma_busy_cycles_per_launch"] = round(busy, 1)
        if busy > 0 and k in mfma_dur:
            # matrix-core utilisation: cycles the MFMA pipes were busy, summed over the SIMDs, over what 1024 SIMDs could have been busy in the kernel's duration
            e["mfma_pass_avg_launch_us"] = round(mfma_dur[k] / 1e3, 2)
            e["mfma_util"] = round(busy / (mfma_dur[k] * CLOCK_GHZ * SIMDS), 5)
    kernels[k] = e
print(json.dumps({"command": "tools/pmc_traffic.sh %s  (rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE|SQ_VALU_MFMA_BUSY_CYCLES> --kernel-trace, three separate passes of bench.py --steps 1 --warmup 1 --owf 0 --decoder-frame-threads 1)" % workload,
                  "workload": workload, "unit": "bytes per launch",
                  "mfma_util": "SQ_VALU_MFMA_BUSY_CYCLES per launch / (launch duration in the same pass x 2.4 GHz x 1024 SIMDs); the mfma pass runs with --subme 4",
                  "correction": "MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is doubled; WRITE_SIZE taken as reported (uncalibrated)",
                  "kernels": kernels}, indent=1))
