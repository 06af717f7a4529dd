#!/bin/bash
# tests/test_gpu_foreign.py::test_b_stream_through_the_filter_graph, four processes side by side, a few rounds: the output stage under a loaded host
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for round in 1 2 3 4 5 6; do
  for p in 1 2 3 4; do (timeout 200 python -m pytest tests/test_gpu_foreign.py -q -m gpu -k "filter_graph" -p no:cacheprovider 2>&1 | grep -E "passed|failed|AssertionError" | tr '\n' ' '; echo) & done
  wait
done
