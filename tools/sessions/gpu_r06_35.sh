#!/bin/bash
# round 6, last library: more seeds of the three decoder soaks
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
{ timeout 1500 python tools/measure/soak_random_access.py 641 1400 2>&1 | grep -v "never arrived" | tail -3
  timeout 1500 python tools/measure/soak_lost_pictures.py 801 1800 2>&1 | grep -v "never arrived" | tail -3
  timeout 1500 python tools/measure/soak_everything.py 1001 1800 2>&1 | grep -v "never arrived" | tail -3
} > gpurun_out/r06_soaks_more.txt 2>&1; cat gpurun_out/r06_soaks_more.txt | cut -c1-400
