#!/bin/bash
# round 4, GPU call 47: the C++ CLI's table with counters on the low-degree inputs, final sources
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
o=gpurun_out/r04/cli_bundles_final.txt
: > $o
for g in tests/golden/pubmed.csv synth:wiki-vote synth:soc-sign-epinions synth:flickr; do for k in 16 32 64 128; do
  echo "=== $g k=$k" >> $o
  timeout -k 10 300 ./flex_amd/lib/flex $g $k --iters 20 --counters 2>&1 | grep -v amdgpu.ids | grep -E "hipSPARSE|^Ord|^OVO|counters:|L1<->L2|skipped|error|NNZ" >> $o || echo "FAILED" >> $o
done; done
grep -E "===|OVO  cluster" $o | cut -c1-140
