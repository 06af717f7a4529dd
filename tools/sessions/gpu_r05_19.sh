#!/bin/bash
# the intra chains' launch width (anti-diagonals of workgroups; KVAZZUP_AMD_INTRA_DIAGS: both sides; the encoder defaults to 2 when two chains run side by side): the all-intra rate
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; OUT=gpurun_out/r05_intra_diags.txt; : > $OUT
run() { echo "== $1" >> $OUT; shift; env "$@" python tools/measure/all_intra_sides.py --both-only 2>&1 | grep "^both" | tail -3 >> $OUT; }
run "default (2 / 2 beside each other)" A=1
run "enc 3 dec 3" KVAZZUP_AMD_INTRA_DIAGS=3 KVAZZUP_AMD_DEC_INTRA_DIAGS=3
run "enc 2 dec 3" KVAZZUP_AMD_INTRA_DIAGS=2 KVAZZUP_AMD_DEC_INTRA_DIAGS=3
run "enc 3 dec 2" KVAZZUP_AMD_INTRA_DIAGS=3 KVAZZUP_AMD_DEC_INTRA_DIAGS=2
run "enc 1 dec 1" KVAZZUP_AMD_INTRA_DIAGS=1 KVAZZUP_AMD_DEC_INTRA_DIAGS=1
run "enc 1 dec 2" KVAZZUP_AMD_INTRA_DIAGS=1 KVAZZUP_AMD_DEC_INTRA_DIAGS=2
run "enc 2 dec 1" KVAZZUP_AMD_INTRA_DIAGS=2 KVAZZUP_AMD_DEC_INTRA_DIAGS=1
run "default again" A=1
cat $OUT
