export TMPDIR=/tmp
mkdir -p gpurun_out/r03m
BLOCK_SWEEP="4:480:3:0,4:480:3:0:4" timeout -k 10 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d gpurun_out/r03m/pmc -- python tools/probe_blocks.py reddit 128 > gpurun_out/r03m/probe.txt 2>&1
grep "reddit" gpurun_out/r03m/probe.txt
f=$(find gpurun_out/r03m/pmc -name "*counter_collection.csv" | head -1)
python tools/pmc_by_variant.py $f spmm_block_kernel 99
