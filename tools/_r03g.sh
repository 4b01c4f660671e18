set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r03g
BLOCK_SWEEP="4:480:2:0,4:480:2:0:1,4:480:2:0:4,4:480:2:0:5,4:480:2:0:7" timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r03g/pmc -- python tools/probe_blocks.py amazon 128 > gpurun_out/r03g/probe.txt 2>&1
grep "amazon" gpurun_out/r03g/probe.txt
f=$(find gpurun_out/r03g/pmc -name "*counter_collection.csv" | head -1)
python tools/pmc_by_variant.py $f spmm_block_kernel 39
