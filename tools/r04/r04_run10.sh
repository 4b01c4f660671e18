#!/bin/bash
# round 4, tenth GPU call: records of a task with the far (likely-miss) columns FIRST -- does clustering the misses inside a wave's
# in-order gather stream raise the rate at which the launch's misses move?
set -o pipefail
mkdir -p gpurun_out/r04
o=gpurun_out/r04/probe_far_first.txt
: > $o
timeout -k 10 300 python -m pytest tests/test_gpu_spmm.py -x -q -k "fuzz or k_sweep or inf" 2>&1 | tail -2 | tee -a $o
timeout -k 10 600 python tools/probe_variant.py reddit 128 far_first=4096 far_first=16384 far_first=65536 >> $o 2>&1
timeout -k 10 900 python tools/probe_variant.py amazon 128 far_first=8192 far_first=40000 far_first=200000 >> $o 2>&1
timeout -k 10 300 python tools/probe_variant.py flickr 128 far_first=1024 far_first=8192 >> $o 2>&1
timeout -k 10 300 python tools/probe_variant.py yelp 128 far_first=2048 far_first=16384 >> $o 2>&1
grep -v amdgpu.ids $o
