#!/bin/bash
# round 4, second GPU call: the hot-block route -- parity first, then time against the flat kernel
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_blocks.py -x -q > gpurun_out/r04/gputest2.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/r04/gputest2.log
tail -15 gpurun_out/r04/gputest2.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/probe_blocks.py reddit 128 > gpurun_out/r04/probe_blocks_reddit.txt 2>&1 && cat gpurun_out/r04/probe_blocks_reddit.txt &&
BLOCK_SWEEP="8:304:2:0,8:304:3:0,8:304:2:0:2,8:304:2:0:3,8:304:2:120,8:304:2:500" timeout -k 10 900 python tools/probe_blocks.py amazon 128 > gpurun_out/r04/probe_blocks_amazon.txt 2>&1 && cat gpurun_out/r04/probe_blocks_amazon.txt
