#!/bin/bash
# round 4, GPU call 20: planning time of the headline by host thread count (the box has 256 cores; the rule stops at 32)
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
o=gpurun_out/r04/plan_threads.txt
: > $o
for t in 16 32 64 128; do
  FLEX_PLAN_TIMING=1 timeout -k 10 400 python bench.py --steps 5 --no-vendor --no-cpu-baseline --no-copy-probe --no-live-counters --host-threads $t 2> gpurun_out/r04/plan_threads_$t.err | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); pl = j['config']['plan']
print('threads $t', 'plan_s', pl['plan_s'], 'order_s', pl['order_s'], 'gen_s', pl['gen_s'], 'kernel ms', j['roofline']['kernel_ms'])" | tee -a $o
  grep -E "^plan:|^cluster: (vertex|round 1 |sweep 1)" gpurun_out/r04/plan_threads_$t.err | head -20 >> $o
done
