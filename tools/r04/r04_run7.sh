#!/bin/bash
# round 4, seventh GPU call: does the hot-block split pay on the low-degree shapes (yelp, flickr)?
set -o pipefail
mkdir -p gpurun_out/r04
o=gpurun_out/r04/probe_blocks_lowdeg.txt
: > $o
BLOCK_SWEEP="8:0:2:0,8:0:3:0,4:0:2:0,8:0:2:0:3" timeout -k 10 400 python tools/probe_blocks.py yelp 128 >> $o 2>&1
BLOCK_SWEEP="2:0:2:0,4:0:2:0,8:0:2:0,4:0:2:0:3" timeout -k 10 300 python tools/probe_blocks.py flickr 128 >> $o 2>&1
grep -v amdgpu.ids $o
