#!/bin/bash
# round 4, sixth GPU call: the whole GPU suite on the round's sources, smoke, the default bench line, the flickr line again
set -o pipefail
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputest6.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r04/gputest6.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err; echo "bench rc=$?"; python -c "
import json; j=json.loads(open('gpurun_out/r04/bench_default.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic'], j['cpu_baseline'])"
for i in 1 2; do python bench.py --workload flickr --no-vendor --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('flickr', j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['frac'])"; done
python bench.py --workload flickr --no-vendor --no-cpu-baseline --steps 200 --tuning split_rows=2 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('flickr two-launch', j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['frac'])"
