#!/bin/bash
# round 4, GPU call 17: row bundles -- candidate length, chunk budget, the narrow tile, and the large presets
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
o=gpurun_out/r04/bundles_sweep.txt
: > $o
line() {  # label, bench args...
  local label=$1; shift
  timeout -k 10 400 python bench.py --steps 200 --no-vendor --no-cpu-baseline --no-copy-probe "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); pl = j['config']['plan']; r = j['roofline']
print('$label', 'G', pl['lanes_per_nz'], 'bundles', pl.get('bundles'), 'rows', pl.get('bundle_rows'), 'tasks', pl['tasks'], 'chunks', pl['chunks'], 'rec', pl.get('records'),
      'us', round(r['kernel_ms'] * 1e3, 2), 'step', round(j['ms_per_step'] * 1e3, 2), 'traffic', r.get('traffic'), r.get('wave_insns_per_64_fma'))" | tee -a $o
}
for w in soc-sign-epinions flickr yelp; do
  for len in 8 12 16 24; do line "$w k=32 len=$len" --workload $w --k 32 --tuning bundle=1,bundle_len=$len; done
  for len in 4 8 16 32; do line "$w k=64 len=$len" --workload $w --k 64 --tuning bundle=1,bundle_len=$len; done
done
for w in soc-sign-epinions flickr; do
  for cr in 64 128 256 384; do line "$w k=32 len=16 chunk=$cr" --workload $w --k 32 --tuning bundle=1,bundle_len=16,chunk_records=$cr; done
  for len in 16 32; do
    line "$w k=16 G=4 len=$len" --workload $w --k 16 --tuning bundle=1,bundle_len=$len,lanes_per_nz=4
    line "$w k=16 G=8 len=$len" --workload $w --k 16 --tuning bundle=1,bundle_len=$len,lanes_per_nz=8
  done
  line "$w k=16 G=4 plain" --workload $w --k 16 --tuning bundle=2,lanes_per_nz=4
  line "$w k=16 G=8 plain" --workload $w --k 16 --tuning bundle=2,lanes_per_nz=8
done
for k in 32 128; do for b in 2 1; do
  line "amazon k=$k bundle=$b" --workload amazon --k $k --steps 20 --tuning bundle=$b,bundle_len=16
done; done
line "ppi k=32 bundle=2" --workload ppi --k 32 --tuning bundle=2
line "ppi k=32 bundle=1" --workload ppi --k 32 --tuning bundle=1,bundle_len=16
