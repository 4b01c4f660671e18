set -x
export TMPDIR=/tmp
PMC_STEPS=5 PMC_WARMUP=2 bash tools/pmc.sh r03f_amazon_blocks --no-vendor --tuning blocks=1 > gpurun_out/r03f.log 2>&1
tail -50 gpurun_out/r03f.log
