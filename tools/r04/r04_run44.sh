#!/bin/bash
# round 4, GPU call 44: small graphs on the 16-lane tile -- 64 records per chunk against the rule's 96
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
o=gpurun_out/r04/budget_small_g16.txt
: > $o
line() {
  local label=$1; shift
  timeout -k 10 300 python bench.py --steps 1000 --no-vendor --no-cpu-baseline --no-copy-probe --no-live-counters "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); pl = j['config']['plan']; r = j['roofline']
print('$label', 'G', pl['lanes_per_nz'], 'bundles', pl['bundles'], 'chunks', pl['chunks'], 'kernel us', round(r['kernel_ms'] * 1e3, 2))" | tee -a $o
}
for rep in 1 2; do for k in 64 128; do
  line "pubmed.csv k=$k rule" --graph tests/golden/pubmed.csv --k $k
  line "pubmed.csv k=$k chunk=64" --graph tests/golden/pubmed.csv --k $k --tuning chunk_records=64
  line "pubmed.csv k=$k chunk=80" --graph tests/golden/pubmed.csv --k $k --tuning chunk_records=80
  line "wiki-vote k=$k rule" --workload wiki-vote --k $k
  line "wiki-vote k=$k chunk=64" --workload wiki-vote --k $k --tuning chunk_records=64
  line "wiki-vote k=$k chunk=80" --workload wiki-vote --k $k --tuning chunk_records=80
done; done
