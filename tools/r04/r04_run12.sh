#!/bin/bash
# round 4, twelfth GPU call: groups dealt to the waves in turn (ablate bit 8) against the total-cost deal
set -o pipefail
mkdir -p gpurun_out/r04
o=gpurun_out/r04/probe_blocks_deal.txt
: > $o
export BLOCK_SWEEP="8:0:3:0,8:0:3:0:8,8:0:2:0,8:0:2:0:8,8:0:3:0:3,8:0:3:0:11"
GEN=p_in=0.75,p_near=0.25 timeout -k 10 600 python tools/probe_blocks.py amazon 128 >> $o 2>&1
timeout -k 10 300 python tools/probe_blocks.py reddit 128 >> $o 2>&1
GEN=p_in=0.75,p_near=0.25 timeout -k 10 300 python tools/probe_blocks.py reddit 128 >> $o 2>&1
grep -v amdgpu.ids $o
