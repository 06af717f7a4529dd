#!/bin/bash
# round 6, sixth GPU session: the suite the way the driver runs it + the driver's bench command, then the round's profile set (kernel statistics pipelined and
# isolated, PMC traffic passes, default-mode chain)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
ROUND=r06 bash tools/gpu_round_end.sh
cd $R; ROUND=r06 bash tools/gpu_profiles.sh 2>&1 | tail -60
