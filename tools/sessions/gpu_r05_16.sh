#!/bin/bash
# the mode search's occupancy cap: its own duration (isolated) and the other kernels' worst launches in the pipelined 4K / 1080p runs, cap 0 (none) against 4 and 3
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
OUT=gpurun_out/r05_analyse_cap_outliers.txt; : > $OUT
show() { python3 - "$1" "$2" >> $OUT <<'P'
import csv, sys
print("==", sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].split('(')[0].replace('kvzx::', '').replace('void ', '')[:30]
    if n.startswith('k_'): print("  %-32s calls %5s avg %8.1f max %8.1f us" % (n, r['Calls'], float(r['AverageNs']) / 1e3, float(r['MaxNs']) / 1e3))
P
}
for n in 0 4 3; do
  export KVAZZUP_AMD_ANALYSE_PER_CU=$n
  bash tools/kstats_iso.sh 1080p cap${n}_iso --streams-per-gpu 0 --no-preset-line > /dev/null 2>&1
  grep -h "k_intra_analyse" gpurun_out/cap${n}_iso_kernel_stats.csv | awk -F, -v n=$n '{printf "cap %s isolated 1080p k_intra_analyse avg %.1f us\n", n, $4/1e3}' >> $OUT
  bash tools/kstats.sh cap${n}_4k --workload 4k --steps 6 --warmup 1 --streams-per-gpu 0 > /dev/null 2>&1
  show gpurun_out/cap${n}_4k_kernel_stats.csv "4K pipelined, cap $n"
done
cat $OUT
