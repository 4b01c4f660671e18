#!/bin/bash
# round 4, GPU call 28: where a 5-us launch goes -- per-wave timeline of the pubmed shape at k=32 and k=128 (diagnostic trace build)
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
make -C flex_amd/csrc trace > gpurun_out/r04/make_trace.log 2>&1 || { tail -5 gpurun_out/r04/make_trace.log; exit 1; }
o=gpurun_out/r04/trace_small.txt
: > $o
for k in 32 128; do timeout -k 10 200 python tools/trace.py pubmed $k 2 2>&1 | grep -v amdgpu.ids >> $o; done
timeout -k 10 200 python tools/trace.py wiki-vote 32 2 2>&1 | grep -v amdgpu.ids >> $o
cut -c1-260 $o
