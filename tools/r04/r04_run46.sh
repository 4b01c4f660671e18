#!/bin/bash
# round 4, GPU call 46: small graphs (B stays in the L2s) -- contiguous XCD slices against round-robin, and the slice balance
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
o=gpurun_out/r04/small_xcd.txt
: > $o
line() {
  local label=$1; shift
  timeout -k 10 300 python bench.py --steps 1000 --no-vendor --no-cpu-baseline --no-copy-probe --no-live-counters "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); pl = j['config']['plan']; r = j['roofline']
print('$label', 'G', pl['lanes_per_nz'], 'chunks', pl['chunks'], 'kernel us', round(r['kernel_ms'] * 1e3, 2))" | tee -a $o
}
for rep in 1 2; do for k in 32 128; do
  line "pubmed.csv k=$k rule" --graph tests/golden/pubmed.csv --k $k
  line "pubmed.csv k=$k round-robin" --graph tests/golden/pubmed.csv --k $k --tuning xcd_slices=2
  line "pubmed.csv k=$k by count" --graph tests/golden/pubmed.csv --k $k --tuning xcd_balance=2
  line "pubmed.csv k=$k natural order" --graph tests/golden/pubmed.csv --k $k --order natural
  line "wiki-vote k=$k rule" --workload wiki-vote --k $k
  line "wiki-vote k=$k round-robin" --workload wiki-vote --k $k --tuning xcd_slices=2
done; done
