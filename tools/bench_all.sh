#!/bin/bash
# One bench.py line (GFLOPS, roofline, hipSPARSE side by side) per BASELINE configuration that fits one GPU,
# appended to gpurun_out/bench_lines.jsonl; copy to profiles/ to commit.
out=${1:-gpurun_out/bench_lines.jsonl}
: > $out
for k in 32 128; do  # the one real data file the reference ships, checked against the oracle
  timeout -k 10 300 python bench.py --graph tests/golden/pubmed.csv --k $k --steps 1000 --warmup 20 --check --no-copy-probe >> $out 2>> ${out%.jsonl}.err || echo "{\"failed\": \"pubmed.csv $k\"}" >> $out
done
for cfg in "pubmed 32" "pubmed 128" "flickr 32" "flickr 128" "yelp 32" "yelp 128" "reddit 32" "reddit 128" "amazon 32" "amazon 128"; do
  set -- $cfg
  steps=200; [ $1 = reddit ] && steps=50; [ $1 = yelp ] && steps=50; [ $1 = amazon ] && steps=10
  timeout -k 10 600 python bench.py --workload $1 --k $2 --steps $steps --warmup 5 --no-cpu-baseline --no-copy-probe >> $out 2>> ${out%.jsonl}.err || echo "{\"failed\": \"$cfg\"}" >> $out
done
# BASELINE configs[2] names RCM-reordered rows for the Reddit shape: the same protocol with RCM as the schedule (the community
# schedule above is what the engine picks by itself)
timeout -k 10 600 python bench.py --workload reddit --k 128 --order rcm --steps 50 --warmup 5 --no-cpu-baseline --no-copy-probe >> $out 2>> ${out%.jsonl}.err || echo "{\"failed\": \"reddit 128 rcm\"}" >> $out
# the range, not only the point: the Reddit and Amazon shapes at the generator's extremes -- no uniformly random edges
# (`best`) and 40 % of them (`worst`) -- next to the preset lines above, and a graph with no communities at all (R-MAT scale 20)
for cfg in "reddit best 50" "reddit worst 50" "amazon best 10" "amazon worst 10"; do
  set -- $cfg
  timeout -k 10 600 python bench.py --workload $1 --variant $2 --k 128 --steps $3 --warmup 5 --no-cpu-baseline --no-copy-probe >> $out 2>> ${out%.jsonl}.err || echo "{\"failed\": \"$cfg\"}" >> $out
done
timeout -k 10 600 python bench.py --workload rmat20 --k 128 --steps 30 --warmup 5 --no-cpu-baseline --no-copy-probe >> $out 2>> ${out%.jsonl}.err || echo "{\"failed\": \"rmat20\"}" >> $out
