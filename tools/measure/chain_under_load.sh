#!/bin/bash
# the two-rank bench beside some of the GPU test suite (what `pytest -n 3` does to it): tools/measure/chain_under_load.sh <test file> '<pytest selection>' [delay]  (GPU box only)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
( for i in 1 2 3 4; do timeout 100 python -m pytest $1 -q -n 2 -k "$2" > /dev/null 2>&1; done & )
sleep ${3:-15}
for i in 1 2; do
timeout 150 python bench.py --gpus 2 --steps 4 --warmup 1 --repeats 1 --no-cpu-baseline --no-host-boundary 2>/dev/null | tail -1 | python3 -c '
import json,sys
l=json.loads(sys.stdin.readline())
print("value", l["value"], "ms_per_step", l["ms_per_step"])
'
done
