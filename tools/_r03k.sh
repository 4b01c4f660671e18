export TMPDIR=/tmp
mkdir -p gpurun_out/r03k
: > gpurun_out/r03k/probe.txt
timeout -k 10 600 python -m pytest tests/test_gpu_blocks.py -x -q 2>&1 | tail -2
for lib in "" libflex_bk_uhot6.so libflex_bk_uhot8.so; do
  echo "== lib=${lib:-default(U_HOT=4)}" >> gpurun_out/r03k/probe.txt
  PROBE_LIB=$lib BLOCK_SWEEP="4:480:3:0" timeout -k 10 600 python tools/probe_blocks.py reddit 128 2>&1 | grep "reddit" >> gpurun_out/r03k/probe.txt
  PROBE_LIB=$lib GEN="p_in=0.75,p_near=0.25" BLOCK_SWEEP="4:480:3:0" timeout -k 10 600 python tools/probe_blocks.py amazon 128 2>&1 | grep "amazon" >> gpurun_out/r03k/probe.txt
done
cat gpurun_out/r03k/probe.txt
