#!/bin/bash
# round 4, GPU call 40: final sources (candidate length 12 on the narrow tiles) -- GPU suite, smoke, bench lines of the small and low-degree inputs, default line
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r04/gputest40.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r04/gputest40.log | head -20
[ $rc = 0 ] || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
b=gpurun_out/r04/bench_final_lines.jsonl
: > $b
for k in 32 128; do timeout -k 10 300 python bench.py --graph tests/golden/pubmed.csv --k $k --steps 1000 --check --no-cpu-baseline 2>/dev/null | tail -1 >> $b; done
for w in wiki-vote soc-sign-epinions flickr; do for k in 32 128; do timeout -k 10 300 python bench.py --workload $w --k $k --steps 300 --no-cpu-baseline 2>/dev/null | tail -1 >> $b; done; done
python - <<'PY'
import json
for l in open("gpurun_out/r04/bench_final_lines.jsonl"):
    j = json.loads(l); r = j["roofline"]; pl = j["config"]["plan"]
    print(j["config"]["workload"][:44], "| k", j["config"]["k"], "G", pl["lanes_per_nz"], "bundles", pl["bundles"], "kernel us", round(r["kernel_ms"] * 1e3, 2), "step us", round(j["ms_per_step"] * 1e3, 2), "frac", r["frac"], "GFLOPS", j["value"], "vendor us", round(j["hipsparse"]["ms_per_step"] * 1e3, 2) if isinstance(j.get("hipsparse"), dict) else None)
PY
timeout -k 10 300 python bench.py > gpurun_out/r04/bench_default_f2.json 2> gpurun_out/r04/bench_default_f2.err; echo "bench rc=$?"
python -c "
import json; j = json.loads(open('gpurun_out/r04/bench_default_f2.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic'], j['cpu_baseline']['value'])"
