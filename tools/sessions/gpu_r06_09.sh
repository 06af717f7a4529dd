#!/bin/bash
# round 6, ninth GPU session: the 64-seed tool sweep with the search on the input picture drawn for half of the seeds (me-source x every other tool), the list-form test
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout 1500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_tok_list.py -q -n 3 2>&1 | tail -6
