// plan.cpp -- life cycle and launch half of the C ABI (include/flex_spmm.h): plan creation entry points, the autotuner,
// flex_spmm, destroy and the info getters.  The planner proper is plan_build.cpp (see plan.h for the map).
//
// Replaces Mat::Mat / csr2_DiagTiling / alpha_transfer / launch_prep /
// alpha_freeMatGPU (mat.cu:7-41, 268-293, 680-942; mat.cuh:184-193).  The
// reference re-cuts A into diagonal "pillars" with per-SM queues and marks most
// rows for atomicAdd; here the plan is a *schedule*: rows (in the given, RCM,
// community or Gorder order) are packed into per-wave chunks of about equal cost,
// rows longer than one chunk budget are cut into pieces that write k-wide partial
// sums (combined in piece order inside the launch), the chunk table is cut into
// eight cost-balanced XCD slices, and column ids are pre-multiplied into B-row
// byte offsets.  Columns always refer to
// the ORIGINAL B, rows always write the ORIGINAL C row, so no permuteX pass and
// no shadow copy of B exist (flex.cu:276-289, mat.cu:287-290).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstddef>
#include <cstdlib>
#include <new>

#include "host_parallel.h"
#include "plan.h"

namespace flex {

static thread_local hipError_t g_last_hip = hipSuccess;
void note_hip_error(hipError_t e) { g_last_hip = e; }

int validate_csr(const flex_csr *A) {
    if (!A || A->m < 0 || A->n < 0 || A->nnz < 0) return FLEX_ERR_INVALID;
    if (!A->rowPtr) return FLEX_ERR_INVALID;
    if (A->nnz > 0 && (!A->col || !A->vals)) return FLEX_ERR_INVALID;
    if (A->nnz >= (int64_t(1) << 32)) return FLEX_ERR_UNSUPPORTED;
    if (A->rowPtr[0] != 0 || A->rowPtr[A->m] != static_cast<uint32_t>(A->nnz)) return FLEX_ERR_INVALID;
    for (int32_t r = 0; r < A->m; ++r)
        if (A->rowPtr[r] > A->rowPtr[r + 1]) return FLEX_ERR_INVALID;
    const uint32_t n = static_cast<uint32_t>(A->n);
    constexpr int64_t kBlk = 1 << 20;
    std::atomic<int> bad{0};
    parallel_chunks((A->nnz + kBlk - 1) / kBlk, [&](int64_t b) {
        uint32_t worst = 0;
        for (int64_t e = b * kBlk; e < std::min<int64_t>(A->nnz, (b + 1) * kBlk); ++e) worst = std::max(worst, A->col[e]);
        if (worst >= n) bad.store(1);
    });
    return bad.load() ? FLEX_ERR_INVALID : FLEX_OK;
}

bool plan_timing_enabled() {
    static const bool on = std::getenv("FLEX_PLAN_TIMING") != nullptr;
    return on;
}

void free_plan_device(flex_plan *p) {
    (void)hipFree(p->d_rec);
    (void)hipFree(p->d_t_beg);
    (void)hipFree(p->d_t_dst);
    (void)hipFree(p->d_t_aux);
    (void)hipFree(p->d_chunk);
    (void)hipFree(p->d_bd_rows);
    (void)hipFree(p->d_chunk_bd);
    (void)hipFree(p->d_partial);
    (void)hipFree(p->d_split);
    (void)hipFree(p->d_split_cnt);
    (void)hipFree(p->d_tile_a);
    (void)hipFree(p->d_tile_boff);
    (void)hipFree(p->d_tile_mask);
    (void)hipFree(p->d_rt_ptr);
    (void)hipFree(p->d_rt_rows);
    (void)hipFree(p->d_bk_hdr);
    (void)hipFree(p->d_bk_wstart);
    (void)hipFree(p->d_bk_cnt);
    (void)hipFree(p->d_bk_hcol);
    (void)hipFree(p->d_bk_brow);
    (void)hipFree(p->d_bk_link);
    (void)hipFree(p->d_bk_rec);
}

}  // namespace flex

using namespace flex;

extern "C" int flex_spmm(flex_plan *p, const float *dB, float *dC, flex_stream_t stream);

namespace {

// FLEX_PLAN_AUTOTUNE: the degree rule picks the column-tile width G from two thresholds measured on a handful of
// shapes; this measures instead.  The neighbouring widths are planned too (same row schedule, computed once),
// each candidate is timed on zero-filled operands of the real size (what a gather costs depends on its address,
// not on the value), and the fastest plan is kept.  Costs up to three extra plans and 2 * 4*(n*ldb + m*ldc) bytes for
// the duration of the call.
int autotune(flex_plan **pp, const flex_csr *A, int32_t r0, int32_t r1, const int32_t *col_map, const int32_t *dst_map,
             unsigned flags, const flex_plan_tuning &tuning, std::vector<uint32_t> &sched_cache) {
    flex_plan *best = *pp;
    if (best->m == 0 || best->k % 4 != 0 || best->ldb % 4 != 0 || best->ldc % 4 != 0) return FLEX_OK;
    int g_max = 8;
    while (4 * g_max < best->k && g_max < 32) g_max <<= 1;
    float *dB = nullptr, *dC = nullptr;
    const size_t b_bytes = static_cast<size_t>(best->n) * best->ldb * sizeof(float);
    const size_t c_bytes = static_cast<size_t>(best->c_rows) * best->ldc * sizeof(float);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = FLEX_OK;
    auto time_plan = [&](flex_plan *q, double *us) {
        for (int i = 0; i < 2 && !rc; ++i) rc = flex_spmm(q, dB, dC, nullptr);
        if (!rc && hipEventRecord(e0, nullptr) != hipSuccess) rc = FLEX_ERR_HIP;
        for (int i = 0; i < 5 && !rc; ++i) rc = flex_spmm(q, dB, dC, nullptr);
        float ms = 0.f;
        if (!rc && (hipEventRecord(e1, nullptr) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                    hipEventElapsedTime(&ms, e0, e1) != hipSuccess))
            rc = FLEX_ERR_HIP;
        *us = ms * 1e3 / 5;
    };
    if (hipMalloc(reinterpret_cast<void **>(&dB), std::max<size_t>(b_bytes, 16)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&dC), std::max<size_t>(c_bytes, 16)) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(dB);
        (void)hipFree(dC);
        return FLEX_OK;  // no room to measure: keep the rule's choice
    }
    if (hipMemset(dB, 0, std::max<size_t>(b_bytes, 16)) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
        rc = FLEX_ERR_HIP;
    double best_us = 0.0;
    if (!rc) time_plan(best, &best_us);
    for (int g : {best->lanes_per_nz / 2, best->lanes_per_nz * 2}) {
        if (rc || g < 8 || g > g_max || g == (*pp)->lanes_per_nz) continue;
        flex_plan *q = new (std::nothrow) flex_plan();
        if (!q) break;
        q->m = best->m; q->n = best->n; q->k = best->k; q->device = best->device;
        q->ldb = best->ldb; q->ldc = best->ldc; q->nnz = best->nnz;
        double us = 0.0;
        int rq = build_plan(q, A, r0, r1, col_map, dst_map, flags, tuning, &sched_cache, g);
        if (rq == FLEX_OK && hipDeviceSynchronize() != hipSuccess) rq = FLEX_ERR_HIP;
        if (rq == FLEX_OK) time_plan(q, &us);
        if (rq == FLEX_OK && !rc && us < best_us) {
            std::swap(best, q);
            best_us = us;
        }
        free_plan_device(q);
        delete q;
    }
    // Row bundles the same way: their rule is one threshold (a chunk per wave slot of the card) between two measured regimes, so on
    // the tiles that have them the other setting is planned and timed too, at the width that won above -- unless the caller chose.
    if (!rc && tuning.bundle == 0 && 64 / best->lanes_per_nz >= static_cast<int>(kBundleMinSlots)) {
        flex_plan_tuning other = tuning;
        other.bundle = best->n_bundles ? 2 : 1;
        flex_plan *q = new (std::nothrow) flex_plan();
        if (q) {
            q->m = best->m; q->n = best->n; q->k = best->k; q->device = best->device;
            q->ldb = best->ldb; q->ldc = best->ldc; q->nnz = best->nnz;
            double us = 0.0;
            int rq = build_plan(q, A, r0, r1, col_map, dst_map, flags, other, &sched_cache, best->lanes_per_nz);
            if (rq == FLEX_OK && hipDeviceSynchronize() != hipSuccess) rq = FLEX_ERR_HIP;
            if (rq == FLEX_OK && q->lanes_per_nz == best->lanes_per_nz && (q->n_bundles != 0) != (best->n_bundles != 0)) time_plan(q, &us);
            else rq = FLEX_ERR_UNSUPPORTED;  // nothing to compare (e.g. no row short enough to bundle)
            if (rq == FLEX_OK && !rc && us < best_us) {
                std::swap(best, q);
                best_us = us;
            }
            free_plan_device(q);
            delete q;
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(dB);
    (void)hipFree(dC);
    *pp = best;
    return rc;
}

}  // namespace

extern "C" {

// every knob is "0 = rule" or a small positive number; anything negative or absurd is a caller bug, not a request
static bool tuning_ok(const flex_plan_tuning &t) {
    const int32_t *f = reinterpret_cast<const int32_t *>(&t);
    for (size_t i = 0; i < sizeof(t) / sizeof(int32_t); ++i)
        if (f[i] < 0 || f[i] > (1 << 28)) return false;
    for (int32_t r : t.reserved)
        if (r != 0) return false;
    return true;
}

static int create_common(flex_plan **out, const flex_csr *hostA, int64_t row_begin, int64_t row_end,
                         const int32_t *col_map, const int32_t *dst_map, int k, int device, unsigned flags,
                         int ldb = 0, int ldc = 0, const flex_plan_tuning *tuning_in = nullptr) {
    if (!out) return FLEX_ERR_INVALID;
    *out = nullptr;
    const flex_plan_tuning tuning = tuning_in ? *tuning_in : flex_plan_tuning{};
    if (!tuning_ok(tuning)) return FLEX_ERR_INVALID;
    const HostThreadsScope threads_for_this_call(tuning.host_threads);
    if (k <= 0 || device < 0) return FLEX_ERR_INVALID;
    if (ldb == 0) ldb = k;
    if (ldc == 0) ldc = k;
    if (ldb < k || ldc < k) return FLEX_ERR_INVALID;
    const unsigned order = flags & FLEX_ORDER_MASK;
    if (order > FLEX_ORDER_GORDER || (flags & ~(FLEX_ORDER_MASK | FLEX_PLAN_STATS | FLEX_PLAN_AUTOTUNE | FLEX_PLAN_XCD_INTERLEAVE))) return FLEX_ERR_INVALID;
    int rc = validate_csr(hostA);
    if (rc) return rc;
    if (hostA->m >= INT32_MAX) return FLEX_ERR_UNSUPPORTED;
    if (row_begin < 0 || row_end < row_begin || row_end > hostA->m) return FLEX_ERR_INVALID;
    if (col_map)
        for (int32_t c = 0; c < hostA->n; ++c)
            if (col_map[c] < 0 || col_map[c] >= hostA->n) return FLEX_ERR_INVALID;
    if (dst_map)
        for (int64_t r = row_begin; r < row_end; ++r)
            if (dst_map[r] < 0 || dst_map[r] >= hostA->m) return FLEX_ERR_INVALID;
    const auto t0 = std::chrono::steady_clock::now();
    int prev = -1;
    FLEX_HIP_TRY(hipGetDevice(&prev));
    FLEX_HIP_TRY(hipSetDevice(device));
    flex_plan *p = new (std::nothrow) flex_plan();
    if (!p) return FLEX_ERR_NOMEM;
    p->m = static_cast<int32_t>(row_end - row_begin);
    p->n = hostA->n;
    p->k = k;
    p->ldb = ldb;
    p->ldc = ldc;
    p->nnz = static_cast<int64_t>(hostA->rowPtr[row_end]) - hostA->rowPtr[row_begin];
    p->device = device;
    std::vector<uint32_t> sched_cache;
    const bool tune = (flags & FLEX_PLAN_AUTOTUNE) != 0;
    try {
        rc = build_plan(p, hostA, static_cast<int32_t>(row_begin), static_cast<int32_t>(row_end), col_map, dst_map, flags, tuning,
                        tune ? &sched_cache : nullptr);
        if (rc == FLEX_OK && tune) rc = autotune(&p, hostA, static_cast<int32_t>(row_begin), static_cast<int32_t>(row_end), col_map, dst_map, flags, tuning, sched_cache);
    } catch (const std::bad_alloc &) {  // nothing crosses the C ABI as an exception
        rc = FLEX_ERR_NOMEM;
    } catch (...) {
        rc = FLEX_ERR_INVALID;
    }
    if (rc == FLEX_OK && hipDeviceSynchronize() != hipSuccess) rc = FLEX_ERR_HIP;
    (void)hipSetDevice(prev);
    if (rc) {
        free_plan_device(p);
        delete p;
        return rc;
    }
    p->plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *out = p;
    return FLEX_OK;
}

int flex_plan_create(flex_plan **out, const flex_csr *hostA, int k, int device, unsigned flags) {
    if (!hostA) return FLEX_ERR_INVALID;
    // dst_map == NULL means slice-local rows, which for the full range is the identity
    return create_common(out, hostA, 0, hostA->m, nullptr, nullptr, k, device, flags);
}

int flex_plan_create_ld(flex_plan **out, const flex_csr *hostA, int k, int ldb, int ldc, int device, unsigned flags) {
    if (!hostA || ldb < k || ldc < k) return FLEX_ERR_INVALID;
    return create_common(out, hostA, 0, hostA->m, nullptr, nullptr, k, device, flags, ldb, ldc);
}

int flex_plan_create_mapped(flex_plan **out, const flex_csr *hostA, const int32_t *vo_mp, int k, int device,
                            unsigned flags) {
    if (!hostA) return FLEX_ERR_INVALID;
    if (vo_mp && hostA->m != hostA->n) return FLEX_ERR_INVALID;
    return create_common(out, hostA, 0, hostA->m, vo_mp, vo_mp, k, device, flags);
}

int flex_plan_create_rows(flex_plan **out, const flex_csr *hostA, int64_t row_begin, int64_t row_end,
                          const int32_t *col_map, int k, int device, unsigned flags) {
    if ((flags & FLEX_ORDER_MASK) != FLEX_ORDER_NATURAL) return FLEX_ERR_INVALID;
    return create_common(out, hostA, row_begin, row_end, col_map, nullptr, k, device, flags);
}

int flex_plan_create_ex(flex_plan **out, const flex_plan_desc *d) {
    constexpr size_t kSizeAbi2 = offsetof(flex_plan_desc, tuning);  // a caller built against ABI 2: no tuning member
    if (!d || !d->A || (d->struct_size != sizeof(flex_plan_desc) && d->struct_size != kSizeAbi2)) return FLEX_ERR_INVALID;
    const flex_plan_tuning *tuning = d->struct_size == sizeof(flex_plan_desc) ? d->tuning : nullptr;
    const bool all_rows = (d->flags & FLEX_PLAN_ROW_RANGE) == 0;  // with the flag, (0,0) is an EMPTY shard, not "everything"
    // without the flag the range members must be zero: a shard range passed by a caller that forgot the flag (or was built before
    // the flag existed) would otherwise get a plan over ALL rows and flex_spmm would write past the shard's C buffer
    if (all_rows && (d->row_begin != 0 || d->row_end != 0)) return FLEX_ERR_INVALID;
    const int64_t r0 = d->row_begin, r1 = all_rows ? d->A->m : d->row_end;
    if (d->row_map && (!all_rows || d->A->m != d->A->n)) return FLEX_ERR_INVALID;  // a row map renames ALL rows of a graph
    if (!all_rows && (d->flags & FLEX_ORDER_MASK) != FLEX_ORDER_NATURAL) return FLEX_ERR_INVALID;  // reorder first, then shard
    return create_common(out, d->A, all_rows ? 0 : r0, r1, d->col_map, d->row_map, d->k, d->device, d->flags & ~FLEX_PLAN_ROW_RANGE, d->ldb, d->ldc, tuning);
}

int flex_spmm(flex_plan *p, const float *dB, float *dC, flex_stream_t stream) {
    if (!p) return FLEX_ERR_INVALID;
    if (p->m == 0) return FLEX_OK;
    if (!dC || (!dB && p->nnz > 0)) return FLEX_ERR_INVALID;
    int cur = -1;
    FLEX_HIP_TRY(hipGetDevice(&cur));
    if (cur != p->device) FLEX_HIP_TRY(hipSetDevice(p->device));
    const bool vec4 = operands_vec4(p, dB, dC);
    const bool fused = vec4 && p->fused_fixup;  // the generic kernel always leaves the sum to spmm_fixup_kernel
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int rc = FLEX_OK;
    // A plan with split rows owns their partial-sum workspace (and, in the in-launch form, their arrival counters): two launches
    // of it must not overlap.  Launches on ONE stream are ordered by the stream; a launch on ANOTHER stream while the stream of the
    // latest one still has work pending is refused instead of silently corrupting those rows.  The check is a stream query at the
    // moment the stream changes -- nothing is added to the launch path of a plan that stays on its stream (an event per launch
    // cost the Flickr-size launches 2-4 us of device time each).  It is conservative: unrelated work queued behind the plan's launch
    // on the old stream also counts as pending.  A launch being captured into a graph is neither checked nor remembered.
    bool guard = p->n_partials > 0;
    if (guard) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess) (void)hipGetLastError();
        if (cap != hipStreamCaptureStatusNone) guard = false;
    }
    if (guard && p->launched && s != p->last_stream) {
        const hipError_t q = hipStreamQuery(p->last_stream);
        if (q == hipErrorNotReady) {
            if (cur != p->device) (void)hipSetDevice(cur);
            return FLEX_ERR_INVALID;
        }
        if (q != hipSuccess) (void)hipGetLastError();  // e.g. the old stream has been destroyed: nothing of ours can be pending on it
    }
    rc = launch_spmm(plan_view(p, fused, p->trace), p->lanes_per_nz, p->off32, vec4, dB, dC, s, p->unroll);
    if (rc == FLEX_OK && !fused) rc = launch_fixup(p->d_partial, p->d_split, p->n_split, p->k, p->ldc, dC, s);
    // the dense tiles' share, added to the rows the kernels above have written
    if (rc == FLEX_OK && p->n_tiles) rc = launch_tiles(tile_view(p), p->off32, dB, dC, p->k, p->ldb, p->ldc, s);
    // the hot blocks' share (the nonzeros with reuse on chip: B rows staged in LDS), added to the rows the flat kernel has written
    if (rc == FLEX_OK && p->bk_blocks) rc = launch_blocks(block_view(p), dB, dC, s, vec4);  // unaligned operands: the generic hot kernel
    if (rc == FLEX_OK && guard) {
        p->last_stream = s;
        p->launched = true;
    }
    if (cur != p->device) (void)hipSetDevice(cur);
    return rc;
}

int flex_plan_destroy(flex_plan *p) {
    if (!p) return FLEX_OK;
    int cur = -1;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(p->device);
    free_plan_device(p);
    if (cur >= 0) (void)hipSetDevice(cur);
    delete p;
    return FLEX_OK;
}

int flex_plan_get_info(const flex_plan *p, flex_plan_info *o) {
    if (!p || !o) return FLEX_ERR_INVALID;
    o->m = p->m;
    o->n = p->n;
    o->k = p->k;
    o->device = p->device;
    o->nnz = p->nnz;
    o->n_tasks = p->n_tasks;
    o->n_chunks = p->n_chunks;
    o->n_split_rows = p->n_split;
    o->n_partials = p->n_partials;
    o->device_bytes = p->device_bytes;
    o->lanes_per_nz = p->lanes_per_nz;
    o->order = static_cast<int32_t>(p->order);
    o->plan_ms = p->plan_ms;
    o->n_slots = p->n_slots;
    o->two_d = p->two_d ? 1 : 0;
    o->n_tiles = p->n_tiles;
    o->tile_nnz = p->tile_nnz;
    o->n_records = static_cast<int64_t>(p->n_records);
    o->panel_rows = p->two_d ? static_cast<int32_t>(p->panel_rows) : 0;
    o->n_blocks = p->bk_blocks;
    o->block_rows = p->bk_rows;
    o->block_nnz = p->bk_nnz;
    o->block_hot_nnz = p->bk_hot_nnz;
    o->block_hot_cols = p->bk_hot_cols;
    o->block_panels = p->bk_panels;
    o->block_records = p->bk_records;
    o->n_bundles = p->n_bundles;
    o->bundle_rows = p->bundle_rows;
    return FLEX_OK;
}

int flex_plan_get_tuning(const flex_plan *p, flex_plan_tuning *o) {
    if (!p || !o) return FLEX_ERR_INVALID;
    *o = p->tuning;
    return FLEX_OK;
}

int flex_set_host_threads(int n) {
    if (n < 0) n = 0;
    return host_threads_cap().exchange(n);
}

int flex_plan_get_stats(const flex_plan *p, flex_plan_stats *o) {
    if (!p || !o) return FLEX_ERR_INVALID;
    if (!p->has_stats) return FLEX_ERR_UNSUPPORTED;
    *o = p->stats;
    return FLEX_OK;
}

int flex_plan_kernel_info(const flex_plan *p, flex_kernel_info *o) {
    if (!p || !o) return FLEX_ERR_INVALID;
    int cur = -1;
    FLEX_HIP_TRY(hipGetDevice(&cur));
    if (cur != p->device) FLEX_HIP_TRY(hipSetDevice(p->device));
    hipFuncAttributes a{};
    int waves = 0;
    const bool vec4 = p->k % 4 == 0 && p->ldb % 4 == 0 && p->ldc % 4 == 0;
    const int rc = kernel_attributes(p->lanes_per_nz, p->off32, vec4, &a, &waves);
    if (cur != p->device) (void)hipSetDevice(cur);
    if (rc) return rc;
    *o = flex_kernel_info{a.numRegs, 0, static_cast<int32_t>(a.sharedSizeBytes), static_cast<int32_t>(a.localSizeBytes), 64 * kWavesPerBlock, waves};
    return FLEX_OK;
}

int flex_gather_rows(float *dst, const float *src, const int32_t *idx, int64_t n, int k, flex_stream_t stream) {
    if (n < 0 || k <= 0) return FLEX_ERR_INVALID;
    if (n == 0) return FLEX_OK;
    if (!dst || !src || !idx) return FLEX_ERR_INVALID;
    return launch_gather_rows(dst, src, idx, n, k, reinterpret_cast<hipStream_t>(stream));
}

#ifdef FLEX_TRACE
// diagnostic build only (libflex_spmm_trace.so, tools/trace.py); not part of the ABI
int flex_debug_set_trace(flex_plan *p, uint64_t *dev_log) {
    if (!p) return FLEX_ERR_INVALID;
    p->trace = dev_log;
    return FLEX_OK;
}
#endif

const char *flex_strerror(int status) {
    switch (status) {
        case FLEX_OK: return "ok";
        case FLEX_ERR_INVALID: return "invalid argument";
        case FLEX_ERR_NOMEM: return "host allocation failed";
        case FLEX_ERR_HIP: return "HIP runtime call failed (see flex_last_hip_error_string)";
        case FLEX_ERR_UNSUPPORTED: return "shape not supported";
        case FLEX_ERR_IO: return "file could not be read";
        case FLEX_ERR_FORMAT: return "input does not parse as a 3-line CSR CSV";
        case FLEX_ERR_DUPLICATE: return "duplicate (row,col) entry";
        default: return "unknown flex status";
    }
}

int flex_last_hip_error(void) { return static_cast<int>(g_last_hip); }
const char *flex_last_hip_error_string(void) { return hipGetErrorString(g_last_hip); }
int flex_abi_version(void) { return FLEX_ABI_VERSION; }

}  // extern "C"
