#!/bin/bash
# round 4, fifth GPU call: the dense-tile kernel (groups of two row tiles sharing B, 128 columns per wave): parity, then the probe
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
o=gpurun_out/r04/mfma_run5.txt
: > $o
timeout -k 10 600 python -m pytest tests/test_gpu_spmm.py -x -q -k "mfma or big_B or 4_GiB" 2>&1 | tail -3 | tee -a $o
for cols in 128 64; do
  echo "== FLEX_MFMA_COLS=$cols" >> $o
  FLEX_MFMA_COLS=$cols timeout -k 10 600 python tools/probe_mfma.py 128 >> $o 2>&1
  FLEX_MFMA_COLS=$cols FLEX_MFMA_ONLY=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/mfma_kt_$cols -- python tools/probe_mfma.py 128 > gpurun_out/r04/mfma_kt_$cols.log 2>&1
  cat gpurun_out/r04/mfma_kt_$cols/*/*_kernel_stats.csv >> $o 2>/dev/null
done
grep -v amdgpu.ids $o
timeout -k 10 600 python -m pytest tests/test_gpu_spmm.py -x -q -k "narrow_tile or k_sweep or fuzz or widths" 2>&1 | tail -3
for k in 16 32; do for w in wiki-vote soc-sign-epinions flickr; do
  python bench.py --workload $w --k $k --steps 200 --no-vendor --no-cpu-baseline --no-copy-probe 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('$w k=$k', 'G', j['config']['plan']['lanes_per_nz'], 'us', round(j['roofline']['kernel_ms']*1e3,2), j['roofline'].get('wave_insns_per_64_fma'))"
  python bench.py --workload $w --k $k --steps 200 --no-vendor --no-cpu-baseline --no-copy-probe --tuning lanes_per_nz=8 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('$w k=$k forced G=8', 'G', j['config']['plan']['lanes_per_nz'], 'us', round(j['roofline']['kernel_ms']*1e3,2), j['roofline'].get('wave_insns_per_64_fma'))"
done; done 2>&1 | tee gpurun_out/r04/narrow_k.txt
