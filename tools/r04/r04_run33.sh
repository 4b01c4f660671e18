#!/bin/bash
# round 4, GPU call 33: the reddit shape on the 8-lane tile moves 14 % fewer bytes than on the 16-lane tile and is 9 % slower -- what holds it?
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
o=gpurun_out/r04/reddit_g8.txt
: > $o
line() {
  local label=$1; shift
  timeout -k 10 400 python bench.py --workload reddit --k 128 --steps 50 --no-vendor --no-cpu-baseline --no-copy-probe "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); pl = j['config']['plan']; r = j['roofline']
print('$label', 'G', pl['lanes_per_nz'], 'chunks', pl['chunks'], 'kernel us', round(r['kernel_ms'] * 1e3, 2), 'traffic GB', round((r.get('traffic') or 0) / 1e9, 3), 'l2hit', r.get('l2_hit_rate'), r.get('wave_insns_per_64_fma'))" | tee -a $o
}
line "G=16 rule"
line "G=8" --tuning lanes_per_nz=8
line "G=8 U=8" --tuning lanes_per_nz=8,unroll=8
line "G=8 rec temporal" --tuning lanes_per_nz=8,rec_nt=2
line "G=8 chunk=256" --tuning lanes_per_nz=8,chunk_records=256
line "G=8 chunk=1024" --tuning lanes_per_nz=8,chunk_records=1024
line "G=8 tile_group=64" --tuning lanes_per_nz=8,tile_group=64
line "G=8 tile_group=512" --tuning lanes_per_nz=8,tile_group=512
line "G=8 lds_extra=16384" --tuning lanes_per_nz=8,lds_extra=16384
line "G=8 in-launch sum" --tuning lanes_per_nz=8,split_rows=1
line "k=32 (one pass)" --k 32
