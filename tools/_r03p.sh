mkdir -p gpurun_out/r03p
timeout -k 10 700 python tools/soak_gpu.py 600 30301 > gpurun_out/r03p/soak.txt 2>&1; echo "rc=$?" >> gpurun_out/r03p/soak.txt
tail -8 gpurun_out/r03p/soak.txt
