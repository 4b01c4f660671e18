#!/bin/bash
# round 4, GPU call 16: row bundles (tuning.bundle) -- parity, then bundle on / off on the low-degree stand-ins, pubmed.csv and the presets
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_bundles.py tests/test_gpu_spmm.py -x -q -k "bundle or fuzz or k_sweep or widths or inf_nan" 2>&1 | tail -15 | tee gpurun_out/r04/bundles_tests.txt
[ "${PIPESTATUS[0]}" = 0 ] || exit 1
o=gpurun_out/r04/bundles_bench.txt
: > $o
line() {  # label, bench args...
  local label=$1; shift
  timeout -k 10 300 python bench.py --steps 200 --no-vendor --no-cpu-baseline --no-copy-probe "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); pl = j['config']['plan']; r = j['roofline']
print('$label', 'G', pl['lanes_per_nz'], 'bundles', pl.get('bundles'), 'rows', pl.get('bundle_rows'), 'tasks', pl['tasks'], 'chunks', pl['chunks'], 'rec', pl.get('records'),
      'us', round(r['kernel_ms'] * 1e3, 2), 'step', round(j['ms_per_step'] * 1e3, 2), r.get('wave_insns_per_64_fma'))" | tee -a $o
}
for k in 32 128; do
  for b in 2 1; do line "pubmed.csv k=$k bundle=$b" --graph tests/golden/pubmed.csv --k $k --steps 1000 --tuning bundle=$b; done
done
for w in wiki-vote soc-sign-epinions flickr; do for k in 16 32 64 128; do
  for b in 2 1; do line "$w k=$k bundle=$b" --workload $w --k $k --tuning bundle=$b; done
done; done
for w in wiki-vote soc-sign-epinions flickr; do
  line "$w k=128 bundle=1 G=8" --workload $w --k 128 --tuning bundle=1,lanes_per_nz=8
  line "$w k=128 bundle=1 G=16" --workload $w --k 128 --tuning bundle=1,lanes_per_nz=16
  line "$w k=32 bundle=1 len=64" --workload $w --k 32 --tuning bundle=1,bundle_len=64
  line "$w k=32 bundle=1 len=16" --workload $w --k 32 --tuning bundle=1,bundle_len=16
done
for w in yelp reddit; do for k in 32 128; do
  for b in 2 1; do line "$w k=$k bundle=$b" --workload $w --k $k --steps 50 --tuning bundle=$b; done
done; done
