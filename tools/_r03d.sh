set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r03d
BLOCK_SWEEP="4:480:2:0,4:480:1000000:0,2:480:2:0,1:480:2:0,4:480:2:2000" timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03d/kt_reddit -- python tools/probe_blocks.py reddit 128 > gpurun_out/r03d/probe_reddit.txt 2>&1
cat gpurun_out/r03d/probe_reddit.txt
find gpurun_out/r03d/kt_reddit -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200
BLOCK_SWEEP="4:480:2:0,4:480:3:0,8:480:2:0" timeout -k 10 900 python tools/probe_blocks.py amazon 128 > gpurun_out/r03d/probe_amazon.txt 2>&1
cat gpurun_out/r03d/probe_amazon.txt
