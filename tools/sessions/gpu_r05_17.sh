#!/bin/bash
# all-intra, both sides: sweeps of the knobs that exist (search cap, decoder's second-stream priority level, decoder frame threads, encoder OWF)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; OUT=gpurun_out/r05_all_intra_sweep.txt; : > $OUT
run() { echo "== $1" >> $OUT; shift; env "$@" python tools/measure/all_intra_sides.py --both-only ${THREADS:-32} 2>&1 | grep "^both" | tail -3 >> $OUT; }
run "defaults (cap 4, alt prio l)" A=1
run "cap 3" KVAZZUP_AMD_ANALYSE_PER_CU=3
run "cap 5" KVAZZUP_AMD_ANALYSE_PER_CU=5
run "cap 0" KVAZZUP_AMD_ANALYSE_PER_CU=0
run "alt prio n" KVAZZUP_AMD_DEC_ALT_PRIO=n
run "alt prio h" KVAZZUP_AMD_DEC_ALT_PRIO=h
THREADS=16 run "16 frame threads" A=1
THREADS=48 run "48 frame threads" A=1
run "intra diags 2" KVAZZUP_AMD_INTRA_DIAGS=2
run "intra diags 4" KVAZZUP_AMD_INTRA_DIAGS=4
cat $OUT
