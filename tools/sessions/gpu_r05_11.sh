#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for v in 3 4 5; do echo "== KVAZZUP_AMD_SIGNAL_VARIANT=$v"; KVAZZUP_AMD_SIGNAL_VARIANT=$v bash tools/kstats_iso.sh 1080p r05_sig$v --streams-per-gpu 0 --no-preset-line 2>&1 | grep -E "k_inter_signal"; done
for v in 3 5; do echo "== 4K KVAZZUP_AMD_SIGNAL_VARIANT=$v"; KVAZZUP_AMD_SIGNAL_VARIANT=$v bash tools/kstats_iso.sh 4k r05_sig4k$v --streams-per-gpu 0 --no-preset-line 2>&1 | grep -E "k_inter_signal"; done
