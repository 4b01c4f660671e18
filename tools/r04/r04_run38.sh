#!/bin/bash
# round 4, GPU call 38: the GPU suite on the relaxed bundle rule, a 4-minute soak
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r04/gputest38.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r04/gputest38.log | head -20
[ $rc = 0 ] || exit 1
timeout -k 10 330 python tools/soak_gpu.py 240 > gpurun_out/r04/soak38.txt 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/r04/soak38.txt
