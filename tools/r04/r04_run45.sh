#!/bin/bash
# round 4, GPU call 45: the chunk floor of 64 on bundled 16-lane plans -- GPU suite, smoke, pubmed.csv lines by rule
set -o pipefail
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r04/gputest45.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r04/gputest45.log | head -20
[ $rc = 0 ] || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-120
b=gpurun_out/r04/bench_pubmed_final.jsonl
: > $b
for k in 16 32 64 128; do timeout -k 10 300 python bench.py --graph tests/golden/pubmed.csv --k $k --steps 1000 --check --no-cpu-baseline 2>/dev/null | tail -1 >> $b; done
for k in 64 128; do timeout -k 10 300 python bench.py --workload wiki-vote --k $k --steps 1000 --no-cpu-baseline 2>/dev/null | tail -1 >> $b; done
python - <<'PY'
import json
for l in open("gpurun_out/r04/bench_pubmed_final.jsonl"):
    j = json.loads(l); r = j["roofline"]; pl = j["config"]["plan"]
    print(j["config"]["workload"][:40], "| k", j["config"]["k"], "G", pl["lanes_per_nz"], "bundles", pl["bundles"], "chunks", pl["chunks"], "kernel us", round(r["kernel_ms"] * 1e3, 2), "step us", round(j["ms_per_step"] * 1e3, 2), "frac", r["frac"], "check", (j.get("check") or {}).get("mismatches"), "vendor us", round(j["hipsparse"]["ms_per_step"] * 1e3, 2) if isinstance(j.get("hipsparse"), dict) else None)
PY
