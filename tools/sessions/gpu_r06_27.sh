#!/bin/bash
# round 6: soak of the random access forms over 400 seeds of the everything-at-once draw
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 2400 python tools/measure/soak_random_access.py 1 640 > gpurun_out/r06_soak_random_access.txt 2>&1; tail -12 gpurun_out/r06_soak_random_access.txt
