#!/bin/bash
# round 4, GPU call 29: small graphs -- do long rows cut into pieces (summed in the launch) shorten the slowest wave?
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
o=gpurun_out/r04/small_long_row.txt
: > $o
line() {
  local label=$1; shift
  timeout -k 10 300 python bench.py --steps 1000 --no-vendor --no-cpu-baseline --no-copy-probe --no-live-counters "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); pl = j['config']['plan']; r = j['roofline']
print('$label', 'G', pl['lanes_per_nz'], 'chunks', pl['chunks'], 'split', pl['split_rows'], 'kernel us', round(r['kernel_ms'] * 1e3, 2), 'step us', round(j['ms_per_step'] * 1e3, 2))" | tee -a $o
}
for k in 32 128; do
  line "pubmed.csv k=$k rule" --graph tests/golden/pubmed.csv --k $k
  for lr in 48 64 96 128 192; do line "pubmed.csv k=$k long_row=$lr" --graph tests/golden/pubmed.csv --k $k --tuning long_row=$lr,piece_records=$lr; done
  line "wiki-vote k=$k rule" --workload wiki-vote --k $k
  for lr in 48 64 96 128 192; do line "wiki-vote k=$k long_row=$lr" --workload wiki-vote --k $k --tuning long_row=$lr,piece_records=$lr; done
done
