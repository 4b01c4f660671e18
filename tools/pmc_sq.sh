#!/bin/bash
# SQ-side counters of the bench kernel: instruction mix, wave cycles, wait shares.
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-counters $*"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $out/sq1 -- $B > $out/sq1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $out/sq2 -- $B > $out/sq2.log 2>&1 || exit 1
python - <<PY
import csv, glob, collections
for d in ("sq1","sq2"):
    for f in glob.glob("$out/%s/*/*_counter_collection.csv" % d):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "spmm_flat" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in sorted(agg.items()):
            print(f"{c:24s} {sum(v)/len(v):14.0f}")
PY
