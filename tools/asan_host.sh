#!/bin/bash
# Host-side sanitizer pass (CPU only; GPU AddressSanitizer is not available on the pool): builds the C ABI's
# host code -- ingest/parsers, orderings, sharding, generator, planner front end -- with g++ -fsanitize=address,undefined
# (kernel launchers stubbed out) and runs the CPU test-suite against that build.
set -e
cd "$(dirname "$0")/.."
out=/tmp/flex_asan; mkdir -p $out
cat > $out/shim.cpp <<'CPP'
#include "internal.h"
namespace flex {  // no device code in this build: launches report "unsupported"
int launch_spmm(const PlanView &, int, bool, bool, const float *, float *, hipStream_t, int) { return FLEX_ERR_UNSUPPORTED; }
int launch_spmm_stamped(const PlanView &, int, bool, const float *, float *, hipStream_t) { return FLEX_ERR_UNSUPPORTED; }
int launch_tiles(const TileView &, bool, const float *, float *, int, int, int, hipStream_t) { return FLEX_ERR_UNSUPPORTED; }
int launch_fixup(const float *, const SplitRow *, uint32_t, int, int, float *, hipStream_t) { return FLEX_ERR_UNSUPPORTED; }
int launch_gather_rows(float *, const float *, const int32_t *, int64_t, int, hipStream_t) { return FLEX_ERR_UNSUPPORTED; }
int kernel_attributes(int, bool, bool, hipFuncAttributes *, int *) { return FLEX_ERR_UNSUPPORTED; }
}
extern "C" int flex_hbm_probe(int, int64_t, int, int, double *, double *) { return FLEX_ERR_UNSUPPORTED; }
CPP
g++ -std=c++20 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -fPIC -shared \
    -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iflex_amd/csrc -o $out/libflex_spmm.so \
    flex_amd/csrc/plan.cpp flex_amd/csrc/ingest.cpp flex_amd/csrc/reorder.cpp flex_amd/csrc/cluster.cpp flex_amd/csrc/rabbit.cpp flex_amd/csrc/gorder.cpp \
    flex_amd/csrc/shard.cpp flex_amd/csrc/synth.cpp $out/shim.cpp -lpthread -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
echo "built $out/libflex_spmm.so"
FLEX_TEST_LIB=$out/libflex_spmm.so LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so) \
  ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests -x -q -m "not gpu" -k "not multigpu and not header and not exports and not no_cpu and not conv_binary" "$@"
