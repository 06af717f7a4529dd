#!/bin/bash
# HBM traffic per kernel launch from the L2's memory-side counters (GPU box).  The pipeline runs synchronously (owf 0, one decoder
# thread): with counters on, kernels are serialised anyway, and the FETCH_SIZE pass has hung with the threaded pipeline.  Three separate rocprofv3 passes (FETCH_SIZE and
# WRITE_SIZE do not fit one pass; counters are never combined with API tracing), each under its own timeout:
#   tools/pmc_traffic.sh <workload> [extra bench args]      ->  gpurun_out/pmc_<workload>/{fetch,write,mfma}/...csv + pmc_traffic_<workload>.json
R=${GRAFT_REPO_ROOT:-$PWD}; wl=${1:-1080p}; shift
out=$R/gpurun_out/pmc_$wl
cd /tmp; export TMPDIR=/tmp
passes=${PMC_PASSES:-"fetch:FETCH_SIZE write:WRITE_SIZE mfma:SQ_VALU_MFMA_BUSY_CYCLES"}     # (the FETCH_SIZE pass has been seen to hang: re-run it alone)
mkdir -p $out
for pass in $passes; do
  name=${pass%%:*}; ctr=${pass##*:}
  extra=""; [ "$name" = "mfma" ] && extra="--subme 4"        # (the MFMA pass also runs the fractional-sample search: k_subpel's Hadamard products)
  KVAZZUP_BENCH_NOPROF=1 timeout -k 5 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$name -o p -- \
    python3 $R/bench.py --workload $wl --no-cpu-baseline --no-secondary --no-host-boundary --repeats 1 --steps 1 --warmup 1 --owf 0 --decoder-frame-threads 1 $extra "$@" > $out.$name.log 2>&1 || echo "pass $name failed (rc $?)"
done
python3 $R/tools/pmc_summarise.py $out $wl > $R/gpurun_out/pmc_traffic_$wl.json && tail -c 600 $R/gpurun_out/pmc_traffic_$wl.json
rm -rf $out      # (per-dispatch counter tables are large: gpurun brings back at most 64 MiB)
