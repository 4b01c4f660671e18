set -x
mkdir -p gpurun_out/r03a
timeout -k 10 500 python tools/probe_bound.py reddit 128 amazon 128 reddit 32 > gpurun_out/r03a/folded.txt 2>&1 &&
timeout -k 10 400 python tools/probe_fixup.py > gpurun_out/r03a/fixup.txt 2>&1 &&
timeout -k 10 900 python tools/probe_hot_split.py reddit 128 amazon 128 > gpurun_out/r03a/hot.txt 2>&1
