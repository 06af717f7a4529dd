#!/bin/bash
# host ISA level A/B: the library's host code at baseline x86-64, at x86-64-v3 (BMI2 / LZCNT / AVX2), and v3 tuned for Zen 4 -- the bench line (the parser alone: no difference, 6.09 ns per bin)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/r05_host_isa_ab.txt
: > $OUT
grep -m1 "model name" /proc/cpuinfo >> $OUT
one() {
  echo -n "bench $1: " >> $OUT
  KVAZZUP_AMD_LIBRARY=$PWD/scratch/abi/lib_$1.so KVAZZUP_BENCH_NOPROF=1 python bench.py --steps 40 --warmup 4 --no-host-boundary --no-preset-line --no-cpu-baseline --no-secondary --streams-per-gpu "" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('host',{}))" >> $OUT
}
one base > /dev/null
for rep in 1 2 3 4; do
  for v in base v3 v3t; do one $v; done
  for v in v3t v3 base; do one $v; done
done
cat $OUT
