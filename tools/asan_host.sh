#!/bin/bash
# Host-side sanitizer pass (CPU only; GPU AddressSanitizer is not available on the pool): builds the C ABI's
# host code -- ingest/parsers, orderings, sharding, generator, planner front end -- with g++ -fsanitize=address,undefined
# (kernel launchers stubbed out: tests/hostsim/shim.cpp) and runs the CPU test-suite against that build; the planner itself runs
# in a second, host-simulated build (tests/test_planner_host.py).
set -e
cd "$(dirname "$0")/.."
out=/tmp/flex_asan; mkdir -p $out
SAN="-std=c++20 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -fPIC -shared"
SRC="flex_amd/csrc/plan.cpp flex_amd/csrc/plan_build.cpp flex_amd/csrc/block_plan.cpp flex_amd/csrc/plan_check.cpp flex_amd/csrc/dense_tiles.cpp flex_amd/csrc/ingest.cpp flex_amd/csrc/reorder.cpp flex_amd/csrc/cluster.cpp flex_amd/csrc/rabbit.cpp flex_amd/csrc/gorder.cpp \
    flex_amd/csrc/shard.cpp flex_amd/csrc/synth.cpp tests/hostsim/shim.cpp"
# (1) the library as the CPU suite sees it: no kernels, real HIP runtime (no device here: plans fail loudly)
g++ $SAN -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iflex_amd/csrc -o $out/libflex_spmm.so $SRC -lpthread -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
# (2) the host-simulated build (tests/hostsim): "device memory" = malloc, so the PLANNER runs under the sanitizers too
g++ $SAN -DFLEX_HOSTSIM -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iflex_amd/csrc -Wl,-Bsymbolic-functions -o $out/libflex_hostsim.so $SRC \
    -lpthread -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
echo "built $out/libflex_spmm.so"
FLEX_TEST_LIB=$out/libflex_spmm.so FLEX_HOSTSIM_LIB=$out/libflex_hostsim.so LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so) \
  ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests -x -q -m "not gpu" -k "not multigpu and not header and not exports and not no_cpu and not conv_binary" "$@"
