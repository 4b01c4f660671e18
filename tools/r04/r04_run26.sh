#!/bin/bash
# round 4, GPU call 26: A/B on ONE box -- the dense-tile kernel of rounds 2-3 against round 4's (whole-tile look-ahead), un-profiled probe.
# tools/r04/tile_kernels_round3.hip.txt was `git show 2f5e1ba:flex_amd/csrc/tile_kernels.hip` (shipped for this call only; not kept in the tree).
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
o=gpurun_out/r04/mfma_ab.txt
: > $o
export PROBE_CASES=64:0.9:8,64:0.6:8,64:0.45:8,64:0.3:8,128:0.3:8 PROBE_THR=10
echo "== round 4 kernel" >> $o
timeout -k 10 600 python tools/probe_mfma.py 128 2>&1 | grep -v amdgpu.ids >> $o
cp flex_amd/csrc/tile_kernels.hip /tmp/tile_kernels_new.hip
cp tools/r04/tile_kernels_round3.hip.txt flex_amd/csrc/tile_kernels.hip && make -C flex_amd/csrc > gpurun_out/r04/make_ab.log 2>&1 || exit 1
echo "== round 3 kernel" >> $o
timeout -k 10 600 python tools/probe_mfma.py 128 2>&1 | grep -v amdgpu.ids >> $o
cut -c1-200 $o
