#!/bin/bash
# round 4, GPU call 41: a last 8-minute soak on the final sources
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
timeout -k 10 560 python tools/soak_gpu.py 480 > gpurun_out/r04/soak41.txt 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/r04/soak41.txt
