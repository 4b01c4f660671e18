set -x
mkdir -p gpurun_out/r03c
timeout -k 10 600 python -m pytest tests/test_gpu_blocks.py -x -q > gpurun_out/r03c/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03c/tests.log
tail -30 gpurun_out/r03c/tests.log
timeout -k 10 600 python tools/probe_blocks.py reddit 128 reddit 32 > gpurun_out/r03c/probe_reddit.txt 2>&1
cat gpurun_out/r03c/probe_reddit.txt
