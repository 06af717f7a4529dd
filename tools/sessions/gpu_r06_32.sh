#!/bin/bash
# round 6: soak of the concealment over 400 seeds of the everything-at-once draw
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 2400 python tools/measure/soak_lost_pictures.py 1 400 > gpurun_out/r06_soak_lost_pictures.txt 2>&1; grep -v "never arrived" gpurun_out/r06_soak_lost_pictures.txt | tail -12
