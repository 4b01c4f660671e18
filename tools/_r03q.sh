export TMPDIR=/tmp
mkdir -p gpurun_out/r03q
echo "== default bench"; timeout -k 10 600 python bench.py > gpurun_out/r03q/bench_default.json 2> gpurun_out/r03q/bench_default.err; cut -c1-400 gpurun_out/r03q/bench_default.json
for w in amazon reddit flickr yelp; do
  steps=30; [ $w = amazon ] && steps=5
  PMC_STEPS=$steps PMC_WARMUP=3 bash tools/pmc.sh r03q_${w}_k128 --workload $w > gpurun_out/r03q/pmc_$w.log 2>&1 || echo "pmc $w FAILED"
  python -c "import json; d=json.load(open('gpurun_out/r03q_${w}_k128/summary.json')); print('$w', d.get('step'))"
done
PMC_STEPS=5 PMC_WARMUP=3 bash tools/pmc.sh r03q_amazon_best_k128 --workload amazon --variant best > gpurun_out/r03q/pmc_amazon_best.log 2>&1 || echo "pmc amazon best FAILED"
python -c "import json; d=json.load(open('gpurun_out/r03q_amazon_best_k128/summary.json')); print('amazon best', d.get('step'))"
PMC_STEPS=5 PMC_WARMUP=3 bash tools/pmc.sh r03q_amazon_best_flat_k128 --workload amazon --variant best --tuning blocks=2 > gpurun_out/r03q/pmc_amazon_best_flat.log 2>&1 || echo "pmc amazon best flat FAILED"
python -c "import json; d=json.load(open('gpurun_out/r03q_amazon_best_flat_k128/summary.json')); print('amazon best flat', d.get('step'))"
