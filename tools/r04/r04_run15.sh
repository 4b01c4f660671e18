#!/bin/bash
# round 4, fifteenth GPU call: final numbers of the hot-block route (packed run counts), the amazon-best bench line by rule, the GPU suite
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
o=gpurun_out/r04/probe_blocks_counts.txt
: > $o
export BLOCK_SWEEP="8:0:3:0,8:0:2:0,8:0:3:0:3,8:0:3:0:51"
GEN=p_in=0.75,p_near=0.25 timeout -k 10 600 python tools/probe_blocks.py amazon 128 >> $o 2>&1
timeout -k 10 600 python tools/probe_blocks.py amazon 128 >> $o 2>&1
timeout -k 10 300 python tools/probe_blocks.py reddit 128 >> $o 2>&1
GEN=p_in=0.75,p_near=0.25 timeout -k 10 300 python tools/probe_blocks.py reddit 128 >> $o 2>&1
grep -v amdgpu.ids $o | cut -c1-150
PMC_STEPS=5 PMC_WARMUP=2 tools/pmc.sh r04/pmc_amazon_best2 --workload amazon --variant best > gpurun_out/r04/pmc_amazon_best2.log 2>&1 && echo "pmc amazon best ok"
python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputest15.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04/gputest15.log
