#!/bin/bash
# round 4, GPU call 27: final sources -- the GPU suite, smoke, a 5-minute soak
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputest27.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r04/gputest27.log
[ $rc = 0 ] || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout -k 10 400 python tools/soak_gpu.py 300 > gpurun_out/r04/soak27.txt 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/r04/soak27.txt
