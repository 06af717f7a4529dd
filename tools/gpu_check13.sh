#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for s in 1 2 3 4; do bash tools/kstats_iso.sh 1080p r02_iso_s$s --subme $s | grep "k_subpel\|k_me "; done
