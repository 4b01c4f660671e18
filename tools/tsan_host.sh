#!/bin/bash
# ThreadSanitizer pass over the threaded host code that runs without a GPU: the synthetic generator and the parallel
# community ordering (worker pool, batched proposals), twice on the flickr shape, ranks compared.  CPU only.
set -e
cd "$(dirname "$0")/.."
out=/tmp/flex_tsan; mkdir -p $out
cat > $out/main.cpp <<CPP
#include <cstdio>
#include <vector>
#include "$(pwd)/include/flex_spmm.h"
int main() {
    flex_synth_params p{};
    if (flex_synth_preset("flickr", 1, &p)) return 1;
    flex_host_csr a{};
    if (flex_synth_graph(&p, &a)) return 2;
    flex_csr v{a.m, a.n, a.nnz, a.rowPtr, a.col, a.vals};
    std::vector<uint32_t> r1(a.m), r2(a.m);
    flex_set_host_threads(6);
    if (flex_order_cluster(&v, r1.data()) || flex_order_cluster(&v, r2.data())) return 3;
    if (r1 != r2) return 4;
    // the new FirstError path of host_parallel.h under the race detector too: a per-call thread count inside a process-wide cap
    flex_cluster_tuning ct{};
    ct.batch = 1024;
    if (flex_order_cluster_ex(&v, &ct, r2.data())) return 5;
    std::printf("tsan pass: generator + cluster ordering ok, n=%d\n", a.m);
    flex_host_csr_free(&a);
    return 0;
}
CPP
cat > $out/shim.cpp <<'CPP'
#include "internal.h"
namespace flex {
int launch_spmm(const PlanView &, int, bool, bool, const float *, float *, hipStream_t, int) { return FLEX_ERR_UNSUPPORTED; }
int launch_spmm_stamped(const PlanView &, int, bool, const float *, float *, hipStream_t) { return FLEX_ERR_UNSUPPORTED; }
int launch_tiles(const TileView &, bool, const float *, float *, int, int, int, hipStream_t) { return FLEX_ERR_UNSUPPORTED; }
int launch_blocks(const BlockView &, const float *, float *, hipStream_t) { return FLEX_ERR_UNSUPPORTED; }
int launch_fixup(const float *, const SplitRow *, uint32_t, int, int, float *, hipStream_t) { return FLEX_ERR_UNSUPPORTED; }
int launch_gather_rows(float *, const float *, const int32_t *, int64_t, int, hipStream_t) { return FLEX_ERR_UNSUPPORTED; }
int kernel_attributes(int, bool, bool, hipFuncAttributes *, int *) { return FLEX_ERR_UNSUPPORTED; }
}
extern "C" int flex_hbm_probe(int, int64_t, int, int, double *, double *) { return FLEX_ERR_UNSUPPORTED; }
CPP
g++ -std=c++20 -O1 -g -fsanitize=thread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iflex_amd/csrc -o $out/t $out/main.cpp \
    flex_amd/csrc/plan.cpp flex_amd/csrc/plan_build.cpp flex_amd/csrc/block_plan.cpp flex_amd/csrc/plan_check.cpp flex_amd/csrc/dense_tiles.cpp flex_amd/csrc/ingest.cpp flex_amd/csrc/reorder.cpp flex_amd/csrc/cluster.cpp flex_amd/csrc/rabbit.cpp \
    flex_amd/csrc/gorder.cpp flex_amd/csrc/shard.cpp flex_amd/csrc/synth.cpp $out/shim.cpp -lpthread -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
$out/t
