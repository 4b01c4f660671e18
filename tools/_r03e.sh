set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r03e
timeout -k 10 600 python -m pytest tests/test_gpu_blocks.py -x -q 2>&1 | tail -3
BLOCK_SWEEP="4:480:2:0,4:480:2:0:2,4:480:2:0:4,4:480:2:0:7,4:480:3:0,8:480:2:0" timeout -k 10 900 python tools/probe_blocks.py amazon 128 2>&1 | grep -v "^[WE]2026\|amdgpu.ids" > gpurun_out/r03e/probe_amazon2.txt
cat gpurun_out/r03e/probe_amazon2.txt
BLOCK_SWEEP="4:480:2:0,2:480:2:0,1:480:2:0,4:480:3:0" timeout -k 10 900 python tools/probe_blocks.py reddit 128 reddit 32 2>&1 | grep -v "^[WE]2026\|amdgpu.ids" > gpurun_out/r03e/probe_reddit2.txt
cat gpurun_out/r03e/probe_reddit2.txt
