#!/bin/bash
# round 4, fourteenth GPU call: run counts packed per panel (four v_readlane per panel instead of two look-ups per run), one branch past the first group of four steps
set -o pipefail
mkdir -p gpurun_out/r04
o=gpurun_out/r04/probe_blocks_counts.txt
: > $o
timeout -k 10 600 python -m pytest tests/test_gpu_blocks.py -x -q 2>&1 | tail -3 | tee -a $o
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
export BLOCK_SWEEP="8:0:3:0,8:0:2:0,8:0:3:0:3,8:0:3:0:51"
GEN=p_in=0.75,p_near=0.25 timeout -k 10 600 python tools/probe_blocks.py amazon 128 >> $o 2>&1
timeout -k 10 300 python tools/probe_blocks.py reddit 128 >> $o 2>&1
GEN=p_in=0.75,p_near=0.25 timeout -k 10 300 python tools/probe_blocks.py reddit 128 >> $o 2>&1
grep -v amdgpu.ids $o | cut -c1-150
