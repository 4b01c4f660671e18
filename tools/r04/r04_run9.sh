#!/bin/bash
# round 4, ninth GPU call: exact non-finite values (value-sharing padding), then the whole GPU suite and a short soak on the final sources
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_spmm.py tests/test_gpu_blocks.py -x -q -k "inf or finite" 2>&1 | tail -3
python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputest9.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04/gputest9.log
timeout -k 10 300 python tools/soak_gpu.py 200 5150 > gpurun_out/r04/soak2.txt 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/r04/soak2.txt
python bench.py --no-vendor --no-cpu-baseline --steps 10 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('amazon', j['ms_per_step'], j['roofline']['frac'])"
python bench.py --workload reddit --no-vendor --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('reddit', j['ms_per_step'], j['roofline']['frac'])"
python bench.py --workload flickr --no-vendor --no-cpu-baseline --steps 200 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('flickr', j['ms_per_step'], j['roofline']['frac'])"
