#!/bin/bash
# round 4, thirteenth GPU call: what the hot kernel's skeleton is made of (timing-only ablations: 3 = no staging, no work; +16 no C read-modify-write; +32 record loads hit one line)
set -o pipefail
mkdir -p gpurun_out/r04
o=gpurun_out/r04/probe_blocks_skeleton.txt
: > $o
export BLOCK_SWEEP="8:0:3:0,8:0:3:0:3,8:0:3:0:19,8:0:3:0:35,8:0:3:0:51"
GEN=p_in=0.75,p_near=0.25 timeout -k 10 600 python tools/probe_blocks.py amazon 128 >> $o 2>&1
GEN=p_in=0.75,p_near=0.25 timeout -k 10 300 python tools/probe_blocks.py reddit 128 >> $o 2>&1
grep -v amdgpu.ids $o | cut -c1-120
