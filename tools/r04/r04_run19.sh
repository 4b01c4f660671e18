#!/bin/bash
# round 4, GPU call 19: with row bundles on, eight gathers in flight per wave on the narrow tiles (tuning.unroll = 8)?
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
o=gpurun_out/r04/bundles_unroll.txt
: > $o
line() {
  local label=$1; shift
  timeout -k 10 400 python bench.py --steps 200 --no-vendor --no-cpu-baseline --no-copy-probe "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); pl = j['config']['plan']; r = j['roofline']
print('$label', 'G', pl['lanes_per_nz'], 'bundles', pl.get('bundles'), 'chunks', pl['chunks'], 'us', round(r['kernel_ms'] * 1e3, 2), 'traffic', r.get('traffic'))" | tee -a $o
}
for w in soc-sign-epinions flickr yelp; do for k in 16 32 64; do
  line "$w k=$k U=4" --workload $w --k $k
  line "$w k=$k U=8" --workload $w --k $k --tuning unroll=8
done; done
