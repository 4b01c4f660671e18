#!/bin/bash
# round 4, GPU call 23: the bytes of the dense-tile route on its probe (fill 0.3 and 0.9), per kernel, from the PMC passes
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
for c in 64:0.3:8 64:0.9:8; do
  tag=$(echo $c | tr ':.' '__'); out=gpurun_out/r04/pmc_mfma_$tag; mkdir -p $out
  export PROBE_CASES=$c PROBE_THR=10
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python tools/probe_mfma.py 128 > $out/kt.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/pmc1 -- python tools/probe_mfma.py 128 > $out/pmc1.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc2 -- python tools/probe_mfma.py 128 > $out/pmc2.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc3 -- python tools/probe_mfma.py 128 > $out/pmc3.log 2>&1 || exit 1
  python tools/pmc_summary.py $out
  grep -v amdgpu.ids $out/kt.log | tail -2
done
