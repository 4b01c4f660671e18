#!/bin/bash
# round 4, first GPU call: the GPU suite, then the bench lines of the BASELINE configurations
set -o pipefail
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputest1.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r04/gputest1.log
tail -5 gpurun_out/r04/gputest1.log
for w in "amazon" "reddit" "reddit --order rcm" "reddit --order rcm --schedule as-given" "flickr" "yelp"; do
  python bench.py --workload $w --no-vendor --steps 20 2>gpurun_out/r04/bench_err.log | tail -1 >> gpurun_out/r04/bench_lines1.jsonl
  echo "done $w"
done
