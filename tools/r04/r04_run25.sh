#!/bin/bash
# round 4, GPU call 25: dense-tile kernel, column groups side by side in a workgroup (A from HBM once) -- parity, probe, bytes
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_spmm.py -x -q -k "mfma or tile" 2>&1 | tail -3
[ "${PIPESTATUS[0]}" = 0 ] || exit 1
o=gpurun_out/r04/mfma_side_by_side.txt
: > $o
timeout -k 10 600 python tools/probe_mfma.py 128 2>&1 | grep -v amdgpu.ids >> $o
timeout -k 10 300 python tools/probe_mfma.py 64 2>&1 | grep -v amdgpu.ids | head -3 >> $o
out=gpurun_out/r04/pmc_mfma_new; mkdir -p $out
export PROBE_CASES=64:0.3:8 PROBE_THR=10
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python tools/probe_mfma.py 128 > $out/kt.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/pmc1 -- python tools/probe_mfma.py 128 > $out/pmc1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc2 -- python tools/probe_mfma.py 128 > $out/pmc2.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc3 -- python tools/probe_mfma.py 128 > $out/pmc3.log 2>&1 || exit 1
python tools/pmc_summary.py $out > /dev/null
python - <<'PY' | tee -a gpurun_out/r04/mfma_side_by_side.txt
import json
s = json.load(open("gpurun_out/r04/pmc_mfma_new/summary.json"))
for n, d in s.items():
    print(" ", n[:50], "calls", d.get("calls"), "avg_us", round(d.get("avg_us", 0), 1), "fetch x2 MB", round(d.get("fetch_MB_x2", 0), 1), "write MB", round(d.get("write_MB", 0), 1), "l2 hit", round(d.get("l2_hit_rate", 0), 3))
PY
cut -c1-330 $o
