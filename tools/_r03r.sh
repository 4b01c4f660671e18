mkdir -p gpurun_out/r03r
timeout -k 10 900 python tools/probe_tile_group.py amazon 128 reddit 128 2>&1 | grep -v amdgpu.ids > gpurun_out/r03r/tile_group.txt
cat gpurun_out/r03r/tile_group.txt
