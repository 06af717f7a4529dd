#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
run() { env "$@" timeout 300 python bench.py --steps 4 --warmup 1 --repeats 2 --no-cpu-baseline --no-secondary --no-host-boundary --streams-per-gpu 0 --no-preset-line --custom period=1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['config']['runs_fps'], d['config']['host_cpu_cores_busy'], {k:d['kernels_us'][k] for k in ('k_intra_recon','k_dec_intra','k_intra_analyse') if k in d['kernels_us']})"; }
for x in h n l; do for e in h n l; do echo "== X=$x E=$e"; run KVAZZUP_AMD_IDR_PRIO=$x KVAZZUP_AMD_DEC_ALT_PRIO=$e; done; done
echo "== owf 8, default levels"; env timeout 300 python bench.py --steps 4 --warmup 1 --repeats 2 --owf 8 --no-cpu-baseline --no-secondary --no-host-boundary --streams-per-gpu 0 --no-preset-line --custom period=1 2>/dev/null | tail -1 | cut -c1-60
# what runs beside what: kernel trace of the all-intra run
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ai
KVAZZUP_BENCH_NOPROF=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ai -o p -- python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-secondary --no-host-boundary --streams-per-gpu 0 --no-preset-line --custom period=1 > /tmp/ai.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/ai/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0].replace('kvzx::','').replace('void ','')[:22],r.get('Queue_Id','?')) for r in rows]
ev.sort()
n=len(ev); mid=ev[n*3//4][0]
print('kernels from the last quarter of the run, us from the first shown; queue id')
for s,e,k,q in ev:
    if mid<=s<mid+4_000_000 and (k.startswith('k_intra_recon') or k.startswith('k_dec_intra') or k.startswith('k_intra_analyse')):
        print('%9.1f .. %9.1f  %-24s q%s'%((s-mid)/1e3,(e-mid)/1e3,k,q))
PY
