#!/bin/bash
# round 4, GPU call 21: k = 128 on the low-degree shapes -- the narrower tiles with bundles against the wide tile, with the bytes
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
o=gpurun_out/r04/bundles_k128.txt
: > $o
line() {
  local label=$1; shift
  timeout -k 10 400 python bench.py --steps 100 --no-vendor --no-cpu-baseline --no-copy-probe "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); pl = j['config']['plan']; r = j['roofline']
print('$label', 'G', pl['lanes_per_nz'], 'bundles', pl.get('bundles'), 'chunks', pl['chunks'], 'us', round(r['kernel_ms'] * 1e3, 2), 'traffic MB', round((r.get('traffic') or 0) / 1e6, 1), 'l2hit', r.get('l2_hit_rate'), 'l1l2 MB', round((r.get('l1_l2_bytes') or 0) / 1e6, 1), r.get('wave_insns_per_64_fma'))" | tee -a $o
}
for w in soc-sign-epinions flickr yelp; do
  line "$w k=128 rule" --workload $w --k 128
  line "$w k=128 G=16" --workload $w --k 128 --tuning lanes_per_nz=16
  line "$w k=128 G=8" --workload $w --k 128 --tuning lanes_per_nz=8
  line "$w k=128 G=8 nt" --workload $w --k 128 --tuning lanes_per_nz=8,rec_nt=1
done
