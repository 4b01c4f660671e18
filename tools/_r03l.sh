export TMPDIR=/tmp
mkdir -p gpurun_out/r03l
timeout -k 10 300 python tools/trace_blocks.py reddit 128 > gpurun_out/r03l/trace.txt 2>&1
GEN="p_in=0.75,p_near=0.25" timeout -k 10 300 python tools/trace_blocks.py amazon 128 >> gpurun_out/r03l/trace.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03l/trace.txt
