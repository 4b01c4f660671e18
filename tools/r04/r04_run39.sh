#!/bin/bash
# round 4, GPU call 39: candidate length of a bundle on the 8- and 4-lane tiles (8 / 12 / 16 records), small and mid-size inputs
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
o=gpurun_out/r04/bundle_len_fine.txt
: > $o
line() {
  local label=$1; shift
  timeout -k 10 300 python bench.py --steps 500 --no-vendor --no-cpu-baseline --no-copy-probe --no-live-counters "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); pl = j['config']['plan']; r = j['roofline']
print('$label', 'G', pl['lanes_per_nz'], 'bundles', pl['bundles'], 'chunks', pl['chunks'], 'kernel us', round(r['kernel_ms'] * 1e3, 2))" | tee -a $o
}
for k in 16 32; do for len in 8 12 16; do
  line "pubmed.csv k=$k len=$len" --graph tests/golden/pubmed.csv --k $k --tuning bundle_len=$len
  for w in wiki-vote soc-sign-epinions flickr yelp; do line "$w k=$k len=$len" --workload $w --k $k --tuning bundle_len=$len; done
done; done
