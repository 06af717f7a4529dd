cd $GRAFT_REPO_ROOT
KVAZZUP_AMD_TIMELINE=/tmp/tl.txt KVAZZUP_BENCH_NOPROF=1 python bench.py --no-cpu-baseline --no-secondary --no-host-boundary --streams-per-gpu 0 --repeats 1 --steps 3 --warmup 1 --custom preset=veryfast --custom bitrate=1000000 --custom rc-algorithm=lambda | grep -o '"value": [0-9.]*' | head -1
python tools/host_timeline.py /tmp/tl.txt | tail -32
