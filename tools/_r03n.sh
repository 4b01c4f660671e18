mkdir -p gpurun_out/r03n
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout -k 10 300 python bench.py --workload flickr --steps 50 --perm-cache gpurun_out/r03n/flickr.perm > gpurun_out/r03n/bench_flickr.json 2> gpurun_out/r03n/bench_flickr.err; tail -c 1500 gpurun_out/r03n/bench_flickr.json
timeout -k 10 300 python bench.py --workload flickr --steps 50 --perm-cache gpurun_out/r03n/flickr.perm --no-vendor --no-cpu-baseline | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['config']['plan'])"
timeout -k 10 300 python bench.py --graph tests/golden/pubmed.csv --k 32 --check --no-vendor --no-cpu-baseline --steps 200 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['check'], j['config']['plan'])"
timeout -k 10 300 python bench.py --graph tests/golden/pubmed.csv --k 128 --check --no-vendor --no-cpu-baseline --steps 200 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['check'])"
