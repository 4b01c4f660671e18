// plan.h -- the plan object and what the planner's translation units share (host side only).
//
//   plan.cpp          life cycle and launch half of the C ABI (create*, flex_spmm, destroy, info), the autotuner
//   plan_build.cpp    PlanBuilder: CSR rows -> schedule -> pieces -> tasks / chunks / records -> device image
//   dense_tiles.cpp   the block-density detector and the dense-tile store of the MFMA route
//   plan_check.cpp    what is read back from a finished plan: self-check, statistics, measured imbalance
#pragma once
#include <algorithm>
#include <cstdlib>
#include <memory>
#include <new>
#include <type_traits>
#include <utility>
#include <vector>

#include "internal.h"

struct flex_plan {
    int32_t m = 0, n = 0, k = 0, device = 0;
    int32_t ldb = 0, ldc = 0;  // row strides of B and C in floats (== k unless flex_plan_create_ld)
    int64_t nnz = 0;
    int lanes_per_nz = 0;
    bool off32 = false;
    bool xcd_remap = true;
    unsigned lds_extra = 0;
    bool rec_nt = false;
    uint32_t tile_group = 0;
    int unroll = 0;
    uint64_t *trace = nullptr;
    unsigned order = 0;
    uint2 *d_rec = nullptr;
    uint32_t *d_t_beg = nullptr, *d_t_dst = nullptr;
    uint2 *d_t_aux = nullptr;
    uint4 *d_chunk = nullptr;
    uint32_t *d_bd_rows = nullptr;  // row bundles (internal.h, PlanView): nullptr when the plan has none
    uint2 *d_chunk_bd = nullptr;
    uint32_t n_bundles = 0, n_bd_rows = 0;  // bundles; entries of d_bd_rows (S per bundle)
    int64_t bundle_rows = 0;                // rows that sit in bundles
    float *d_partial = nullptr;
    flex::SplitRow *d_split = nullptr;
    uint32_t *d_split_cnt = nullptr;
    bool fused_fixup = false;
    bool two_d = false;  // rows cut by column panel (phases), not only by length
    // dense 32x32 tiles routed to the MFMA kernel (tile_kernels.hip)
    float *d_tile_a = nullptr;
    uint32_t *d_tile_boff = nullptr, *d_tile_mask = nullptr, *d_rt_ptr = nullptr, *d_rt_rows = nullptr;
    uint32_t n_tiles = 0, n_row_tiles = 0;
    int64_t tile_nnz = 0;
    int64_t tile_hist[3] = {0, 0, 0}, tile_cells = 0;  // detector report
    bool tile_hist_valid = false;
    uint32_t panel_rows = 0;
    // hot blocks (block_kernels.hip): the nonzeros they hold are not in the record stream above; their rows are (the flat kernel
    // writes every row, the hot kernel adds to the rows of its blocks)
    uint4 *d_bk_hdr = nullptr;
    uint2 *d_bk_wstart = nullptr, *d_bk_rec = nullptr;
    uint32_t *d_bk_cnt = nullptr, *d_bk_hcol = nullptr, *d_bk_brow = nullptr, *d_bk_link = nullptr;
    uint32_t bk_blocks = 0, bk_rounds = 0, bk_panel_rows = 0, bk_ablate = 0;
    int64_t bk_rows = 0, bk_nnz = 0, bk_hot_nnz = 0, bk_hot_cols = 0, bk_panels = 0, bk_records = 0;
    uint32_t n_tasks = 0, n_chunks = 0, n_slots = 0, n_split = 0, n_partials = 0;  // n_slots: chunk table incl. padding
    uint64_t n_records = 0;   // nnz + padding
    int64_t c_rows = 0;       // rows of C the plan writes into (m, or hostA->m for a mapped plan)
    int64_t device_bytes = 0;
    double plan_ms = 0;
    double lds_hot[2] = {0, 0}, lds_u[2] = {0, 0};  // FLEX_PLAN_STATS: hot share and u of 480-row blocks at thr 2 / 4
    bool has_stats = false;
    flex_plan_stats stats{};
    flex_plan_tuning tuning{};  // the knobs this plan was built with, rules resolved (flex_plan_get_tuning)
    // in-flight guard of a plan that owns a split-row workspace (flex_spmm): the stream of its latest launch
    hipStream_t last_stream = nullptr;
    bool launched = false;
};

namespace flex {


// std::vector<T>(n) zero-fills: for the record stream (8 B per nonzero) that is a single-threaded pass over gigabytes that
// the parallel fill overwrites straight away.  With this allocator resize() leaves trivial elements uninitialised, and the
// pages are first touched by the threads that fill them.
template <typename T>
struct default_init_allocator : std::allocator<T> {
    template <typename U>
    struct rebind {
        using other = default_init_allocator<U>;
    };
    template <typename U>
    void construct(U *ptr) noexcept(std::is_nothrow_default_constructible_v<U>) {
        ::new (static_cast<void *>(ptr)) U;
    }
    template <typename U, typename... Args>
    void construct(U *ptr, Args &&...args) {
        ::new (static_cast<void *>(ptr)) U(std::forward<Args>(args)...);
    }
};
using RecordVec = std::vector<uint2, default_init_allocator<uint2>>;

template <typename T, typename A>
int upload(T **dptr, const std::vector<T, A> &h, int64_t *bytes) {
    *dptr = nullptr;
    const size_t nb = (h.empty() ? 1 : h.size()) * sizeof(T);
    FLEX_HIP_TRY(hipMalloc(reinterpret_cast<void **>(dptr), nb));
    if (!h.empty()) FLEX_HIP_TRY(hipMemcpy(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *bytes += static_cast<int64_t>(nb);
    return FLEX_OK;
}

void free_plan_device(flex_plan *p);

// The kernels' view of a finished plan.  `fused`: split rows are summed inside the launch.
inline PlanView plan_view(const flex_plan *p, bool fused, uint64_t *trace) {
    return PlanView{p->d_rec, p->d_t_beg, p->d_t_dst, p->d_t_aux, p->d_chunk, p->d_partial, p->d_split, p->d_split_cnt,
                    fused ? 1u : 0u, p->n_slots, p->k, p->ldb, p->ldc,
                    p->xcd_remap ? 1u : 0u, p->lds_extra, p->rec_nt ? 1u : 0u, p->tile_group, trace, p->d_bd_rows, p->d_chunk_bd};
}
inline BlockView block_view(const flex_plan *p) {
    return BlockView{p->d_bk_hdr, p->d_bk_wstart, p->d_bk_cnt, p->d_bk_hcol, p->d_bk_brow, p->d_bk_link, p->d_bk_rec, static_cast<uint64_t>(std::max<int64_t>(p->bk_records, 1)),
                     p->bk_blocks, p->bk_rounds, p->bk_panel_rows, p->k, p->ldb, p->ldc, 1u, p->bk_ablate, p->trace};
}
inline TileView tile_view(const flex_plan *p) { return TileView{p->d_tile_a, p->d_tile_boff, p->d_tile_mask, p->d_rt_ptr, p->d_rt_rows, p->n_row_tiles}; }
// float4 path: k and both strides multiples of 4, both base addresses 16-byte aligned
inline bool operands_vec4(const flex_plan *p, const float *dB, const float *dC) {
    return (p->k % 4 == 0) && (p->ldb % 4 == 0) && (p->ldc % 4 == 0) &&
           ((reinterpret_cast<uintptr_t>(dB) | reinterpret_cast<uintptr_t>(dC)) % 16 == 0);
}

// Rows [r0,r1) of A.  col_map: B row read by column c (NULL = c).  dst_map: C row written by row r (NULL = r - r0,
// i.e. slice-local).  sched_cache (or NULL): holds the row schedule once it has been computed, so that several
// candidate plans of one matrix (autotune) order it only once.  force_G (or 0): lanes per record instead of the degree rule.
int build_plan(flex_plan *p, const flex_csr *A, int32_t r0, int32_t r1, const int32_t *col_map, const int32_t *dst_map,
               unsigned flags, const flex_plan_tuning &tuning, std::vector<uint32_t> *sched_cache = nullptr, int force_G = 0);

// ---- block-density detector (dense_tiles.cpp)
struct DenseTiles {
    std::vector<float> a;           // [T][4][64][4]
    std::vector<uint32_t> boff;     // [T][32]
    std::vector<uint32_t> mask;     // [T][32] which (row, column) cells of the tile hold an entry
    std::vector<uint32_t> rt_ptr;   // [R+1]
    std::vector<uint32_t> rt_rows;  // [R][32]
    int64_t nnz = 0;                // entries moved into tiles
    int64_t hist_nnz[3] = {0, 0, 0};
    int64_t n_cells = 0;            // (row tile, column tile) pairs with at least one entry
};
// stride > 1: look at every stride-th row tile only (thr must be 0)
int detect_dense_tiles(const flex_csr *A, int32_t r0, int32_t m, const std::vector<uint32_t> &sched, const std::vector<uint32_t> &colpos,
                       const int32_t *col_map, const int32_t *dst_map, bool off32, uint32_t row_bytes32, uint32_t thr, int64_t stride,
                       std::vector<uint8_t> &in_tile, DenseTiles &out);

// ---- hot blocks (block_plan.cpp)
struct BlockKnobs {
    uint32_t rounds = 8, panel_rows = kBkPanelMax, thr = 2, cap = 256, max_panels = 31, min_last_panel = 32, run_max = kBkRunMax;
};
struct BlockImage {  // host copy of what BlockView points at
    std::vector<uint4> hdr;
    std::vector<uint2> wstart;
    std::vector<uint32_t> cnt, hcol, brow, link;
    RecordVec rec;
    uint32_t n_blocks = 0, rounds = 0, panel_rows = 0;
    int64_t rows = 0, nnz = 0, hot_nnz = 0, hot_cols = 0, panels = 0;
    int64_t cand_nnz = 0, lost_panels = 0, lost_last = 0, lost_run = 0;  // candidates (column uses >= thr) and where some were left cold
};
// What share of the nonzeros of rows sched[...] would be HOT (their column used by >= thr nonzeros of the same block of `rows`
// schedule-consecutive rows), looked at in every `stride`-th block: the planner's cheap look before it commits to the block route.
double estimate_hot_share(const flex_csr *A, const std::vector<uint32_t> &sched, uint32_t rows, uint32_t thr, int64_t stride, double *u = nullptr);
// Rows sched[0..) of A (a row's C row: dst_map, or r - r0) into blocks.  hot_mask[e - rowPtr[r0]] = 1 for every nonzero that went
// into the block image; the caller plans the OTHER nonzeros with the flat planner.
int build_blocks(const flex_csr *A, const std::vector<uint32_t> &sched, const std::vector<uint32_t> &colpos, const int32_t *col_map,
                 const int32_t *dst_map, int32_t r0, uint32_t row_bytes32, const BlockKnobs &kn, BlockImage &img, std::vector<uint8_t> &hot_mask);

// ---- plan_check.cpp
void collect_stats(flex_plan *p, const RecordVec &rec, const std::vector<uint4> &chunk, int64_t split_nnz);

}  // namespace flex
