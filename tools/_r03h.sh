export TMPDIR=/tmp
mkdir -p gpurun_out/r03h
for g in "p_in=0.75,p_near=0.25" "p_in=0.85,p_near=0.15" "p_in=0.60,p_near=0.40"; do
GEN=$g BLOCK_SWEEP="4:480:2:0,4:480:3:0,2:480:2:0" timeout -k 10 600 python tools/probe_blocks.py reddit 128 2>&1 | grep "reddit" >> gpurun_out/r03h/probe.txt
done
GEN="p_in=0.75,p_near=0.25" BLOCK_SWEEP="4:480:2:0,4:480:3:0" timeout -k 10 600 python tools/probe_blocks.py amazon 128 2>&1 | grep "amazon" >> gpurun_out/r03h/probe.txt
cat gpurun_out/r03h/probe.txt
