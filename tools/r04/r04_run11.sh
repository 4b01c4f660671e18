#!/bin/bash
# round 4, eleventh GPU call: the generic hot kernel (unaligned operands with a split plan), then the GPU suite
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_blocks.py -x -q 2>&1 | tail -3
python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputest11.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04/gputest11.log
