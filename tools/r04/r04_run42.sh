#!/bin/bash
# round 4, GPU call 42: the rocprofv3 passes (kernel trace + three PMC passes each) of the BASELINE lines on the FINAL sources
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
PMC_STEPS=5 PMC_WARMUP=2 tools/pmc.sh r04/pmc_amazon_z --workload amazon > gpurun_out/r04/pmc_amazon_z.log 2>&1 && echo "pmc amazon ok"
PMC_STEPS=20 PMC_WARMUP=3 tools/pmc.sh r04/pmc_reddit_z --workload reddit > gpurun_out/r04/pmc_reddit_z.log 2>&1 && echo "pmc reddit ok"
PMC_STEPS=30 PMC_WARMUP=3 tools/pmc.sh r04/pmc_flickr_z --workload flickr > gpurun_out/r04/pmc_flickr_z.log 2>&1 && echo "pmc flickr ok"
PMC_STEPS=30 PMC_WARMUP=3 tools/pmc.sh r04/pmc_epinions32_z --workload soc-sign-epinions --k 32 > gpurun_out/r04/pmc_epinions32_z.log 2>&1 && echo "pmc epinions ok"
PMC_STEPS=200 PMC_WARMUP=10 tools/pmc.sh r04/pmc_pubmed32_z --graph tests/golden/pubmed.csv --k 32 > gpurun_out/r04/pmc_pubmed32_z.log 2>&1 && echo "pmc pubmed ok"
