#!/bin/bash
# round 4, GPU call 36: the small inputs again with the candidate length the rule settled on (16; the first comparison used 32)
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
o=gpurun_out/r04/bundles_small.txt
: > $o
line() {
  local label=$1; shift
  timeout -k 10 300 python bench.py --steps 1000 --no-vendor --no-cpu-baseline --no-copy-probe --no-live-counters "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); pl = j['config']['plan']; r = j['roofline']
print('$label', 'G', pl['lanes_per_nz'], 'bundles', pl['bundles'], 'chunks', pl['chunks'], 'kernel us', round(r['kernel_ms'] * 1e3, 2), 'step us', round(j['ms_per_step'] * 1e3, 2))" | tee -a $o
}
for k in 16 32 64 128; do
  line "pubmed.csv k=$k off" --graph tests/golden/pubmed.csv --k $k --tuning bundle=2
  for len in 8 16; do line "pubmed.csv k=$k on len=$len" --graph tests/golden/pubmed.csv --k $k --tuning bundle=1,bundle_len=$len; done
done
for w in wiki-vote ppi; do for k in 16 32 64 128; do
  line "$w k=$k off" --workload $w --k $k --tuning bundle=2
  line "$w k=$k on len=16" --workload $w --k $k --tuning bundle=1
done; done
for cr in 32 48 96; do line "pubmed.csv k=32 on len=16 chunk=$cr" --graph tests/golden/pubmed.csv --k 32 --tuning bundle=1,chunk_records=$cr; done
