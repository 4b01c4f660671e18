#!/bin/bash
# round 4, GPU call 31: XCD slices weighted by the time left after each XCD's start (tuning.xcd_skew) against equal cost shares
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
o=gpurun_out/r04/xcd_skew.txt
: > $o
line() {
  local label=$1; shift
  timeout -k 10 300 python bench.py --steps 1000 --no-vendor --no-cpu-baseline --no-copy-probe --no-live-counters "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); pl = j['config']['plan']; r = j['roofline']
print('$label', 'chunks', pl['chunks'], 'kernel us', round(r['kernel_ms'] * 1e3, 2), 'step us', round(j['ms_per_step'] * 1e3, 2))" | tee -a $o
}
for rep in 1 2 3; do for s in 2 1; do
  line "pubmed.csv k=32 skew=$s" --graph tests/golden/pubmed.csv --k 32 --tuning xcd_skew=$s
  line "pubmed.csv k=128 skew=$s" --graph tests/golden/pubmed.csv --k 128 --tuning xcd_skew=$s
  line "wiki-vote k=32 skew=$s" --workload wiki-vote --k 32 --tuning xcd_skew=$s
  line "ppi k=32 skew=$s" --workload ppi --k 32 --tuning xcd_skew=$s
  line "flickr k=32 skew=$s" --workload flickr --k 32 --tuning xcd_skew=$s
done; done
sort $o | uniq | cut -c1-120
