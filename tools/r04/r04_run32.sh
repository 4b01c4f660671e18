#!/bin/bash
# round 4, GPU call 32: the tile-width thresholds once more on the final sources (reddit shape: 8 vs 16 lanes; amazon: 8 vs 16)
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
o=gpurun_out/r04/tile_width_check.txt
: > $o
line() {
  local label=$1; shift
  timeout -k 10 400 python bench.py --no-vendor --no-cpu-baseline --no-copy-probe "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); pl = j['config']['plan']; r = j['roofline']
print('$label', 'G', pl['lanes_per_nz'], 'chunks', pl['chunks'], 'kernel us', round(r['kernel_ms'] * 1e3, 2), 'traffic GB', round((r.get('traffic') or 0) / 1e9, 3), 'l2hit', r.get('l2_hit_rate'))" | tee -a $o
}
for g in 16 8; do line "reddit k=128 G=$g" --workload reddit --k 128 --steps 50 --tuning lanes_per_nz=$g; done
for g in 16 8; do line "reddit k=128 G=$g chunk=512" --workload reddit --k 128 --steps 50 --tuning lanes_per_nz=$g,chunk_records=512; done
for g in 8 16; do line "amazon k=128 G=$g" --workload amazon --k 128 --steps 10 --tuning lanes_per_nz=$g; done
