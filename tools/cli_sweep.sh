#!/bin/bash
# SuiteSparse-class + GNN stand-ins through the C++ CLI (hipSPARSE side by side), k = 32 and 128.
# CLI_EXTRA=--counters adds the measured HBM-side bytes, L2 hit rate and B reuse u under every row (in-run counters).
out=${1:-gpurun_out/cli_sweep.txt}
: > $out
for g in tests/golden/pubmed.csv synth:wiki-vote synth:soc-sign-epinions synth:ppi synth:flickr synth:yelp synth:reddit; do
  for k in 32 128; do
    echo "=== $g k=$k" >> $out
    timeout -k 10 300 ./flex_amd/lib/flex $g $k --iters 20 $CLI_EXTRA 2>&1 | grep -v amdgpu.ids | grep -E "hipSPARSE|^Ord|^OVO|^RCM|^RBT|^DFS|^GOR|^DEG|counters:|L1<->L2|skipped|error|NNZ" >> $out || echo "FAILED" >> $out
  done
done
