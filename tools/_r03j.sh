export TMPDIR=/tmp
mkdir -p gpurun_out/r03j
BLOCK_SWEEP="4:480:3:0,4:480:3:0:1,4:480:3:0:2,4:480:3:0:4,4:480:3:0:7" timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r03j/pmc -- python tools/probe_blocks.py reddit 128 > gpurun_out/r03j/probe.txt 2>&1
grep "reddit" gpurun_out/r03j/probe.txt
f=$(find gpurun_out/r03j/pmc -name "*counter_collection.csv" | head -1)
python tools/pmc_by_variant.py $f spmm_block_kernel 99
python tools/pmc_by_variant.py $f spmm_flat_kernel 99
