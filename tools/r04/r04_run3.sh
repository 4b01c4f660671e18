#!/bin/bash
# round 4, third GPU call: the hot-block route with the flat part's own tile width; presets and the generator's other points
set -o pipefail
mkdir -p gpurun_out/r04
o=gpurun_out/r04/probe_blocks_run3.txt
: > $o
timeout -k 10 300 python -m pytest tests/test_gpu_blocks.py -x -q 2>&1 | tail -2 | tee -a $o
export BLOCK_SWEEP="8:304:2:0,8:304:3:0,8:304:2:0:3"
timeout -k 10 300 python tools/probe_blocks.py reddit 128 >> $o 2>&1
GEN=p_in=0.75,p_near=0.25 timeout -k 10 300 python tools/probe_blocks.py reddit 128 >> $o 2>&1
timeout -k 10 600 python tools/probe_blocks.py amazon 128 >> $o 2>&1
GEN=p_in=0.75,p_near=0.25 timeout -k 10 600 python tools/probe_blocks.py amazon 128 >> $o 2>&1
GEN=p_in=1.0,p_near=0.0 timeout -k 10 600 python tools/probe_blocks.py amazon 128 >> $o 2>&1
grep -v amdgpu.ids $o
