#!/bin/bash
# round 4, GPU call 18: row bundles by rule -- the whole GPU suite, the soak with the bundle knobs, the CLI table with counters on the
# low-degree inputs, the bench lines for profiles/
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputest18.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r04/gputest18.log
[ $rc = 0 ] || exit 1
timeout -k 10 330 python tools/soak_gpu.py 240 > gpurun_out/r04/soak18.txt 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/r04/soak18.txt
o=gpurun_out/r04/cli_bundles.txt
: > $o
for g in synth:soc-sign-epinions synth:wiki-vote synth:flickr tests/golden/pubmed.csv; do for k in 16 32 64; do
  echo "=== $g k=$k" >> $o
  timeout -k 10 300 ./flex_amd/lib/flex $g $k --iters 20 --counters 2>&1 | grep -v amdgpu.ids | grep -E "hipSPARSE|^Ord|^OVO|counters:|L1<->L2|skipped|error|NNZ" >> $o || echo "FAILED" >> $o
done; done
b=gpurun_out/r04/bench_bundles.jsonl
: > $b
for w in soc-sign-epinions flickr yelp; do for k in 16 32 64; do
  timeout -k 10 400 python bench.py --workload $w --k $k --steps 200 --no-cpu-baseline 2>/dev/null | tail -1 >> $b
done; done
for k in 32 128; do timeout -k 10 300 python bench.py --graph tests/golden/pubmed.csv --k $k --steps 1000 --check --no-cpu-baseline 2>/dev/null | tail -1 >> $b; done
python - <<'PY'
import json
for l in open("gpurun_out/r04/bench_bundles.jsonl"):
    j = json.loads(l); r = j["roofline"]; pl = j["config"]["plan"]
    print(j["config"]["workload"][:40], "k", j["config"]["k"], "G", pl["lanes_per_nz"], "bundles", pl["bundles"], "us", round(r["kernel_ms"] * 1e3, 2), "frac", r["frac"], "traffic/alg", round((r.get("traffic") or 0) / max(r["algorithmic_bytes"], 1), 2) if "algorithmic_bytes" in r else None, "vendor", j.get("hipsparse", {}).get("ms") if isinstance(j.get("hipsparse"), dict) else None)
PY
