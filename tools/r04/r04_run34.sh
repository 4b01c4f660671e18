#!/bin/bash
# round 4, GPU call 34: reddit shape, the 8-lane tile with the grouped tile order (a group's four column tiles back to back) -- group size, U, against the 16-lane rule
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
o=gpurun_out/r04/reddit_g8_groups.txt
: > $o
line() {
  local label=$1; shift
  timeout -k 10 400 python bench.py --workload reddit --k 128 --steps 50 --no-vendor --no-cpu-baseline --no-copy-probe "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); pl = j['config']['plan']; r = j['roofline']
print('$label', 'G', pl['lanes_per_nz'], 'chunks', pl['chunks'], 'kernel us', round(r['kernel_ms'] * 1e3, 2), 'step us', round(j['ms_per_step'] * 1e3, 2), 'traffic GB', round((r.get('traffic') or 0) / 1e9, 3), 'l2hit', r.get('l2_hit_rate'))" | tee -a $o
}
line "G=16 rule"
for tg in 256 512 1024 2048 4096; do line "G=8 tile_group=$tg" --tuning lanes_per_nz=8,tile_group=$tg; done
for tg in 512 1024 2048; do line "G=8 U=8 tile_group=$tg" --tuning lanes_per_nz=8,unroll=8,tile_group=$tg; done
for tg in 512 2048; do line "G=16 tile_group=$tg" --tuning lanes_per_nz=16,tile_group=$tg; done
line "G=16 rule again"
