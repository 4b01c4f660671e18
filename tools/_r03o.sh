mkdir -p gpurun_out/r03o
timeout -k 10 600 python tools/probe_small2.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03o/small2.txt
cat gpurun_out/r03o/small2.txt
