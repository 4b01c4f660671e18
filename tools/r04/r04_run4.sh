#!/bin/bash
# round 4, fourth GPU call: evidence for profiles/ -- rocprofv3 kernel trace + PMC passes per BASELINE workload, bench lines, soak
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
PMC_STEPS=5 PMC_WARMUP=2 tools/pmc.sh r04/pmc_amazon --workload amazon > gpurun_out/r04/pmc_amazon.log 2>&1 && echo "pmc amazon ok"
PMC_STEPS=20 PMC_WARMUP=3 tools/pmc.sh r04/pmc_reddit --workload reddit > gpurun_out/r04/pmc_reddit.log 2>&1 && echo "pmc reddit ok"
PMC_STEPS=20 PMC_WARMUP=3 tools/pmc.sh r04/pmc_reddit_rcm --workload reddit --order rcm > gpurun_out/r04/pmc_reddit_rcm.log 2>&1 && echo "pmc reddit rcm ok"
PMC_STEPS=30 PMC_WARMUP=3 tools/pmc.sh r04/pmc_flickr --workload flickr > gpurun_out/r04/pmc_flickr.log 2>&1 && echo "pmc flickr ok"
PMC_STEPS=20 PMC_WARMUP=3 tools/pmc.sh r04/pmc_yelp --workload yelp > gpurun_out/r04/pmc_yelp.log 2>&1 && echo "pmc yelp ok"
PMC_STEPS=5 PMC_WARMUP=2 tools/pmc.sh r04/pmc_amazon_best --workload amazon --variant best > gpurun_out/r04/pmc_amazon_best.log 2>&1 && echo "pmc amazon best ok"
timeout -k 10 420 python tools/soak_gpu.py 300 4041 > gpurun_out/r04/soak1.txt 2>&1; echo "soak rc=$?"; tail -3 gpurun_out/r04/soak1.txt
tools/bench_all.sh gpurun_out/r04/bench_lines.jsonl; echo "bench_all rc=$?"; wc -l gpurun_out/r04/bench_lines.jsonl
