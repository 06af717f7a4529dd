#!/bin/bash
# round 6, fifth GPU session: k_tokenize's list form (P pictures: a fixed number of waves works off k_inter_signal's list of (unit, role) pairs) -- parity, isolated kernel
# times at 1080p / 4K against the form of rounds 1-5 (KVAZZUP_AMD_TOK_LIST=0), default mode, the wave census
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_me_source.py tests/test_gpu_configs.py tests/test_gpu_filters.py tests/test_gpu_batch.py tests/test_gpu_golden.py -x -q -n 3 --deselect tests/test_gpu_configs.py::test_config3_bench_command_with_two_ranks 2>&1 | tail -4
cd $R; for v in list grid; do
  [ $v = grid ] && export KVAZZUP_AMD_TOK_LIST=0
  echo "== $v 1080p"; bash tools/kstats_iso.sh 1080p r06_iso1080p_tok_$v --streams-per-gpu 0 --no-preset-line 2>&1 | grep -E "k_tok|k_inter_sig"
  echo "== $v 4k"; bash tools/kstats_iso.sh 4k r06_iso4k_tok_$v --streams-per-gpu 0 2>&1 | grep -E "k_tok|k_inter_sig"
  echo "== $v 1080p default mode"; bash tools/kstats_iso.sh 1080p r06_iso1080p_dm_tok_$v --streams-per-gpu 0 --no-preset-line --custom preset=veryfast --custom bitrate=1000000 --custom rc-algorithm=lambda 2>&1 | grep -E "k_tok|k_inter_sig"
  cd $R; echo "== $v census 1080p"; python tools/tok_timeline.py 1920 1080 5 2>&1 | grep -A4 "P pictures"
done
