#!/bin/bash
# round 4, GPU call 24: the dense-tile kernel with a whole tile of look-ahead -- parity, then the probe at 3 and at 2 waves per SIMD
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_spmm.py -x -q -k "mfma or tile" 2>&1 | tail -3
[ "${PIPESTATUS[0]}" = 0 ] || exit 1
o=gpurun_out/r04/mfma_lookahead.txt
: > $o
probe() {
  echo "== $1" >> $o
  timeout -k 10 600 python tools/probe_mfma.py 128 2>&1 | grep -v amdgpu.ids >> $o
  PROBE_CASES=64:0.3:8 PROBE_THR=10 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/mfma_kt_$2 -- python tools/probe_mfma.py 128 > gpurun_out/r04/mfma_kt_$2.log 2>&1
  grep tile_kernel gpurun_out/r04/mfma_kt_$2/*/*_kernel_stats.csv >> $o
}
probe "three waves per SIMD (36 bytes of scratch per lane)" w3
touch flex_amd/csrc/tile_kernels.hip && make -C flex_amd/csrc EXTRA=-DFLEX_TILE_WAVES=2 > gpurun_out/r04/make_w2.log 2>&1 || exit 1
probe "two waves per SIMD (no scratch)" w2
cat $o | cut -c1-400
