#!/bin/bash
# round 6, fourth GPU session: the tokenizer's closed-form count pass -- parity (the encoder suite), isolated kernel times at 1080p / 4K with 5 and 4 waves
# per SIMD (the count brought four spilled VGPRs at 5), the wave census
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_me_source.py -x -q -n 3 2>&1 | tail -4
cd $R; for v in w5 w4; do
  [ $v = w4 ] && export KVAZZUP_AMD_LIBRARY=$R/kvazzup_amd/libkvazzup_amd_w4.so
  echo "== $v 1080p"; bash tools/kstats_iso.sh 1080p r06_iso1080p_$v --streams-per-gpu 0 --no-preset-line 2>&1 | grep -E "k_tok|k_inter_sig"
  echo "== $v 4k"; bash tools/kstats_iso.sh 4k r06_iso4k_$v --streams-per-gpu 0 2>&1 | grep -E "k_tok|k_inter_sig"
  cd $R; echo "== $v census 1080p"; python tools/tok_timeline.py 1920 1080 5 2>&1 | tail -9
done
