mkdir -p gpurun_out/r03s
bash tools/bench_all.sh gpurun_out/r03s/bench_lines.jsonl
wc -l gpurun_out/r03s/bench_lines.jsonl
python - <<'PY'
import json
for ln in open("gpurun_out/r03s/bench_lines.jsonl"):
    try:
        j = json.loads(ln)
    except Exception:
        print("unparsed:", ln[:100]); continue
    if "failed" in j: print(j); continue
    print(f"{j['config']['workload'][:70]:70s} {j['ms_per_step']*1e3:10.1f} us {j['value']:9.1f} GFLOPS frac {j['roofline']['frac']:.4f} hipsparse {j.get('hipsparse',{}).get('ms_per_step')}")
PY
