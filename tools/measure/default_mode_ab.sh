timeout 1200 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_configs.py tests/test_gpu_filters.py tests/test_gpu_hash.py -q -n 3 2>&1 | tail -3
bash tools/measure/chain_gaps.sh default 2>&1 | grep -E "k_me  |queue 2:"
dm() { python bench.py --no-cpu-baseline --no-secondary --no-host-boundary --streams-per-gpu 0 --repeats 3 --steps 8 --warmup 1 --custom preset=veryfast --custom bitrate=1000000 --custom rc-algorithm=lambda 2>/dev/null | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('$1', l['value'], l['config']['runs_fps'])"; }
dm deferred; dm deferred
KVAZZUP_AMD_DEFER_TOK=0 dm at-once
