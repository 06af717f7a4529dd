#!/bin/bash
# round 6: the whole GPU suite after the merge (serial, as the driver runs it) + the driver's bench line
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
ROUND=r06b bash tools/gpu_round_end.sh
