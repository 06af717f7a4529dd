#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for nw in 1 4; do export KVAZZUP_AMD_TOK_WAVES=$nw; echo "== waves per workgroup $nw"; python tools/tok_timeline.py 3840 2160 | grep -A6 "k_tokenize: span"; bash tools/kstats_iso.sh 4k t19_iso4k_nw$nw 2>&1 | grep "k_tokenize"; grep "k_tokenize" gpurun_out/t19_iso4k_nw${nw}_kernel_stats.csv | awk -F, '{print "  min",$(NF-2),"max",$(NF-1)}'; cd $R; done
