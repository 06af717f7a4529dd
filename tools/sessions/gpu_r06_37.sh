#!/bin/bash
# round 6: long soaks on the final library
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
{ timeout 2400 python tools/measure/soak_everything.py 9301 13000 2>&1 | grep -v "never arrived" | tail -4
  timeout 2400 python tools/measure/soak_lost_pictures.py 6401 9000 2>&1 | grep -v "never arrived" | tail -4
  timeout 2400 python tools/measure/soak_random_access.py 4001 5400 2>&1 | grep -v "never arrived" | tail -4
} > gpurun_out/r06_soaks_more3.txt 2>&1; cut -c1-420 gpurun_out/r06_soaks_more3.txt
