#!/bin/bash
# round 4, GPU call 35: the autotuner also measures the bundle setting -- its test, then what it picks on the small and mid-size inputs
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_spmm.py -x -q -k "autotune" 2>&1 | tail -3
[ "${PIPESTATUS[0]}" = 0 ] || exit 1
o=gpurun_out/r04/autotune_bundles.txt
: > $o
line() {
  local label=$1; shift
  timeout -k 10 300 python bench.py --steps 300 --no-vendor --no-cpu-baseline --no-copy-probe --no-live-counters "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); pl = j['config']['plan']; r = j['roofline']
print('$label', 'G', pl['lanes_per_nz'], 'bundles', pl['bundles'], 'chunks', pl['chunks'], 'kernel us', round(r['kernel_ms'] * 1e3, 2), 'plan s', pl['plan_s'])" | tee -a $o
}
for w in wiki-vote ppi flickr soc-sign-epinions; do for k in 32 64; do
  line "$w k=$k rule" --workload $w --k $k
  line "$w k=$k autotune" --workload $w --k $k --autotune
done; done
line "pubmed.csv k=32 rule" --graph tests/golden/pubmed.csv --k 32
line "pubmed.csv k=32 autotune" --graph tests/golden/pubmed.csv --k 32 --autotune
