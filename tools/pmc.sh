#!/bin/bash
# usage: tools/pmc.sh <tag> <bench args...>   -- kernel trace + 3 PMC passes, summaries under gpurun_out/<tag>/
set -o pipefail
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
B="python bench.py --steps ${PMC_STEPS:-30} --warmup ${PMC_WARMUP:-5} --no-cpu-baseline --no-copy-probe --no-vendor --no-live-counters $*"  # the same command for all four passes: only engine kernels in the trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- $B > $out/kt.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/pmc1 -- $B > $out/pmc1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc2 -- $B > $out/pmc2.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc3 -- $B > $out/pmc3.log 2>&1 || exit 1
python tools/pmc_summary.py $out
