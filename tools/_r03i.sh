export TMPDIR=/tmp
mkdir -p gpurun_out/r03i
timeout -k 10 600 python -m pytest tests/test_gpu_blocks.py -x -q 2>&1 | tail -3
BLOCK_SWEEP="4:480:2:0,4:480:3:0,2:480:3:0" timeout -k 10 600 python tools/probe_blocks.py reddit 128 2>&1 | grep "reddit" > gpurun_out/r03i/probe.txt
BLOCK_SWEEP="4:480:3:0" timeout -k 10 600 python tools/probe_blocks.py amazon 128 2>&1 | grep "amazon" >> gpurun_out/r03i/probe.txt
GEN="p_in=0.75,p_near=0.25" BLOCK_SWEEP="4:480:3:0" timeout -k 10 600 python tools/probe_blocks.py amazon 128 2>&1 | grep "amazon" >> gpurun_out/r03i/probe.txt
GEN="p_in=0.75,p_near=0.25" BLOCK_SWEEP="4:480:3:0" timeout -k 10 600 python tools/probe_blocks.py reddit 128 2>&1 | grep "reddit" >> gpurun_out/r03i/probe.txt
cat gpurun_out/r03i/probe.txt
make -s -C flex_amd/csrc trace >/dev/null 2>&1
timeout -k 10 300 python tools/trace_blocks.py reddit 128 2>&1 | grep -v amdgpu.ids
