#!/bin/bash
# round 4, GPU call 30: is the start skew between the XCDs (first wave of XCD 5 about 1.5 us after XCD 0's) a property of the machine?  A second box, more shapes
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
make -C flex_amd/csrc trace > gpurun_out/r04/make_trace.log 2>&1 || { tail -5 gpurun_out/r04/make_trace.log; exit 1; }
o=gpurun_out/r04/trace_skew.txt
: > $o
for spec in "pubmed 32" "wiki-vote 32" "ppi 32" "flickr 32" "flickr 128" "soc-sign-epinions 32"; do
  for rep in 1 2; do timeout -k 10 200 python tools/trace.py $spec 2 2>&1 | grep -E "span=|xcc[0-9]:" >> $o; done
done
cut -c1-200 $o
