#!/bin/bash
# round 4, GPU call 22: the 16-lane tile with bundles as the rule at low degree -- the GPU suite, repeats against the wide tile, then the
# PMC passes of the BASELINE lines on the final sources and the default bench line
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputest22.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r04/gputest22.log
[ $rc = 0 ] || exit 1
o=gpurun_out/r04/bundles_k128_rule.txt
: > $o
line() {
  local label=$1; shift
  timeout -k 10 400 python bench.py --steps 100 --no-vendor --no-cpu-baseline --no-copy-probe "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); pl = j['config']['plan']; r = j['roofline']
print('$label', 'G', pl['lanes_per_nz'], 'bundles', pl.get('bundles'), 'chunks', pl['chunks'], 'kernel us', round(r['kernel_ms'] * 1e3, 2), 'step us', round(j['ms_per_step'] * 1e3, 2), 'traffic MB', round((r.get('traffic') or 0) / 1e6, 1), 'l2hit', r.get('l2_hit_rate'))" | tee -a $o
}
for rep in 1 2; do for w in flickr soc-sign-epinions yelp; do
  line "$w k=128 rule" --workload $w --k 128
  line "$w k=128 wide tile, no bundles" --workload $w --k 128 --tuning bundle=2
done; done
PMC_STEPS=5 PMC_WARMUP=2 tools/pmc.sh r04/pmc_amazon_f --workload amazon > gpurun_out/r04/pmc_amazon_f.log 2>&1 && echo "pmc amazon ok"
PMC_STEPS=20 PMC_WARMUP=3 tools/pmc.sh r04/pmc_reddit_f --workload reddit > gpurun_out/r04/pmc_reddit_f.log 2>&1 && echo "pmc reddit ok"
PMC_STEPS=30 PMC_WARMUP=3 tools/pmc.sh r04/pmc_flickr_f --workload flickr > gpurun_out/r04/pmc_flickr_f.log 2>&1 && echo "pmc flickr ok"
PMC_STEPS=30 PMC_WARMUP=3 tools/pmc.sh r04/pmc_epinions32_f --workload soc-sign-epinions --k 32 > gpurun_out/r04/pmc_epinions32_f.log 2>&1 && echo "pmc epinions ok"
timeout -k 10 300 python bench.py > gpurun_out/r04/bench_default_f.json 2> gpurun_out/r04/bench_default_f.err; echo "bench rc=$?"
python -c "
import json; j = json.loads(open('gpurun_out/r04/bench_default_f.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic'], j['cpu_baseline']['value'])"
