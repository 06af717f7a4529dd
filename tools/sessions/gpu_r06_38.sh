#!/bin/bash
# round 6: last soaks -- the encoder's default mode over 6000 pictures against the checker, the decoder soaks on fresh seeds
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
{ timeout 2400 python tools/measure/soak_default_mode.py 6000 2>&1 | tail -2
  timeout 1500 python tools/measure/soak_everything.py 13001 15000 2>&1 | grep -v "never arrived" | tail -2
  timeout 1500 python tools/measure/soak_lost_pictures.py 9001 10500 2>&1 | grep -v "never arrived" | tail -2
} > gpurun_out/r06_soaks_last.txt 2>&1; cut -c1-300 gpurun_out/r06_soaks_last.txt
