#!/bin/bash
# round 4, eighth GPU call: the hot kernel's loader two panels ahead (three buffers of 200 rows, counted vmcnt) against one ahead (two of 304)
set -o pipefail
mkdir -p gpurun_out/r04
o=gpurun_out/r04/probe_blocks_ring.txt
: > $o
timeout -k 10 600 python -m pytest tests/test_gpu_blocks.py -x -q 2>&1 | tail -3 | tee -a $o
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
export BLOCK_SWEEP="8:0:3:0,8:0:2:0,8:0:3:0:3"
for lib in "" libflex_bk_nbuf2.so; do
  echo "== PROBE_LIB=${lib:-product (3 buffers x 200 rows)}" >> $o
  GEN=p_in=0.75,p_near=0.25 PROBE_LIB=$lib timeout -k 10 600 python tools/probe_blocks.py amazon 128 >> $o 2>&1
  PROBE_LIB=$lib timeout -k 10 300 python tools/probe_blocks.py reddit 128 >> $o 2>&1
  GEN=p_in=0.75,p_near=0.25 PROBE_LIB=$lib timeout -k 10 300 python tools/probe_blocks.py reddit 128 >> $o 2>&1
done
PROBE_LIB= timeout -k 10 600 python tools/probe_blocks.py amazon 128 >> $o 2>&1
grep -v amdgpu.ids $o
bash tools/r04/r04_run7.sh
