#!/bin/bash
# round 4, GPU call 43: the chunk budget of the 16-lane tile now that short rows sit in bundles (the rule, clamp(8 x degree, 96, 256), dates from before)
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
o=gpurun_out/r04/budget_g16_bundles.txt
: > $o
line() {
  local label=$1; shift
  timeout -k 10 300 python bench.py --steps 300 --no-vendor --no-cpu-baseline --no-copy-probe --no-live-counters "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); pl = j['config']['plan']; r = j['roofline']
print('$label', 'G', pl['lanes_per_nz'], 'bundles', pl['bundles'], 'chunks', pl['chunks'], 'kernel us', round(r['kernel_ms'] * 1e3, 2))" | tee -a $o
}
for w in flickr soc-sign-epinions; do for k in 64 128; do
  line "$w k=$k rule" --workload $w --k $k
  for cr in 64 128 192 256; do line "$w k=$k chunk=$cr" --workload $w --k $k --tuning chunk_records=$cr; done
done; done
line "yelp k=128 rule" --workload yelp --k 128 --steps 100
for cr in 128 256 384; do line "yelp k=128 chunk=$cr" --workload yelp --k 128 --steps 100 --tuning chunk_records=$cr; done
line "pubmed.csv k=128 rule" --graph tests/golden/pubmed.csv --k 128 --steps 1000
for cr in 48 64 128; do line "pubmed.csv k=128 chunk=$cr" --graph tests/golden/pubmed.csv --k 128 --steps 1000 --tuning chunk_records=$cr; done
