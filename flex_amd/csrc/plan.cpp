// plan.cpp -- host planner + the plan/launch half of the C ABI (include/flex_spmm.h).
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
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>

#include "host_parallel.h"
#include "internal.h"

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

static long env_long(const char *name, long dflt) {
    const char *s = std::getenv(name);
    if (!s || !*s) return dflt;
    char *e = nullptr;
    long v = std::strtol(s, &e, 10);
    return (e && *e == 0 && v > 0) ? v : dflt;
}

}  // namespace flex

using namespace flex;

struct flex_plan {
    int32_t m = 0, n = 0, k = 0, device = 0;
    int32_t ldb = 0, ldc = 0;  // row strides of B and C in floats (== k unless flex_plan_create_ld)
    int64_t nnz = 0;
    int lanes_per_nz = 0;
    bool off32 = false;
    bool xcd_remap = true;
    unsigned lds_extra = 0;
    bool rec_nt = false;
    int unroll = 0;
    uint64_t *trace = nullptr;
    unsigned order = 0;
    uint2 *d_rec = nullptr;
    uint32_t *d_t_beg = nullptr, *d_t_dst = nullptr;
    uint2 *d_t_aux = nullptr;
    uint4 *d_chunk = nullptr;
    float *d_partial = nullptr;
    SplitRow *d_split = nullptr;
    uint32_t *d_split_cnt = nullptr;
    bool fused_fixup = false;
    bool two_d = false;  // rows cut by column panel (phases), not only by length
    // dense 32x32 tiles routed to the MFMA kernel (tile_kernels.hip)
    float *d_tile_a = nullptr;
    uint32_t *d_tile_boff = nullptr, *d_rt_ptr = nullptr, *d_rt_rows = nullptr;
    uint32_t n_tiles = 0, n_row_tiles = 0;
    int64_t tile_nnz = 0;
    int64_t tile_hist[3] = {0, 0, 0}, tile_cells = 0;  // detector report
    bool tile_hist_valid = false;
    uint32_t panel_rows = 0;
    uint32_t n_tasks = 0, n_chunks = 0, n_slots = 0, n_split = 0, n_partials = 0;  // n_slots: chunk table incl. padding
    uint64_t n_records = 0;   // nnz + padding
    int64_t c_rows = 0;       // rows of C the plan writes into (m, or hostA->m for a mapped plan)
    int64_t device_bytes = 0;
    double plan_ms = 0;
    bool has_stats = false;
    flex_plan_stats stats{};
};

namespace {

template <typename T>
int upload(T **dptr, const std::vector<T> &h, int64_t *bytes) {
    *dptr = nullptr;
    const size_t nb = (h.empty() ? 1 : h.size()) * sizeof(T);
    FLEX_HIP_TRY(hipMalloc(reinterpret_cast<void **>(dptr), nb));
    if (!h.empty()) FLEX_HIP_TRY(hipMemcpy(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *bytes += static_cast<int64_t>(nb);
    return FLEX_OK;
}

void free_plan_device(flex_plan *p) {
    (void)hipFree(p->d_rec);
    (void)hipFree(p->d_t_beg);
    (void)hipFree(p->d_t_dst);
    (void)hipFree(p->d_t_aux);
    (void)hipFree(p->d_chunk);
    (void)hipFree(p->d_partial);
    (void)hipFree(p->d_split);
    (void)hipFree(p->d_split_cnt);
    (void)hipFree(p->d_tile_a);
    (void)hipFree(p->d_tile_boff);
    (void)hipFree(p->d_rt_ptr);
    (void)hipFree(p->d_rt_rows);
}

// ≙ alpha_stats_collect (mat.cu:944-1065): distinct B rows per chunk / workgroup / XCD slice by
// stamping, and how evenly records are cut.  Padding records repeat the row's last column, so they
// change no distinct count.
void collect_stats(flex_plan *p, const std::vector<uint2> &rec, const std::vector<uint4> &chunk, int64_t split_nnz) {
    flex_plan_stats &st = p->stats;
    st = flex_plan_stats{};
    const uint32_t n_chunks = static_cast<uint32_t>(chunk.size());
    uint32_t nblk = (n_chunks + kWavesPerBlock - 1) / kWavesPerBlock;
    nblk = (nblk + kXcds - 1) / kXcds * kXcds;  // as launch_spmm cuts the grid
    const uint32_t cpx = std::max(1u, nblk / kXcds);
    const uint32_t row_bytes32 = static_cast<uint32_t>(p->ldb) * 4u;
    std::vector<uint32_t> seen_wave(p->n, 0u), seen_wg(p->n, 0u), seen_xcd(p->n, 0u);
    int64_t xcd_rec[kXcds] = {0};
    for (uint32_t c = 0; c < n_chunks; ++c) {
        const uint32_t wg = c / kWavesPerBlock;
        const uint32_t xcd = p->xcd_remap ? std::min<uint32_t>(wg / cpx, kXcds - 1) : wg % kXcds;
        const uint32_t n_rec = chunk[c].w - chunk[c].z;
        if (chunk[c].y == 0) continue;  // padding slot
        for (uint32_t z = chunk[c].z; z < chunk[c].w; ++z) {
            const uint32_t col = p->off32 ? rec[z].x / row_bytes32 : rec[z].x;
            if (seen_wave[col] != c + 1) seen_wave[col] = c + 1, st.cols_wave++;
            if (seen_wg[col] != wg + 1) seen_wg[col] = wg + 1, st.cols_wg++;
            if (seen_xcd[col] != xcd + 1) seen_xcd[col] = xcd + 1, st.cols_xcd++;
        }
        xcd_rec[xcd] += n_rec;
        st.chunk_rec_max = std::max<int64_t>(st.chunk_rec_max, n_rec);
    }
    st.records = static_cast<int64_t>(rec.size());
    st.n_workgroups = nblk;
    const double nnz = static_cast<double>(p->nnz - p->tile_nnz);  // what the vector kernel processes
    st.reuse_wave = st.cols_wave ? nnz / st.cols_wave : 0.0;
    st.reuse_wg = st.cols_wg ? nnz / st.cols_wg : 0.0;
    st.reuse_xcd = st.cols_xcd ? nnz / st.cols_xcd : 0.0;
    st.gather_bytes = 4.0 * (p->m + 1) + 8.0 * nnz + 4.0 * nnz * p->k + 4.0 * p->m * p->k;
    st.l2_bytes = 4.0 * (p->m + 1) + 8.0 * st.records + 4.0 * p->k * st.cols_xcd + 4.0 * p->m * p->k;
    st.chunk_rec_mean = p->n_chunks ? static_cast<double>(st.records) / p->n_chunks : 0.0;
    st.chunk_imb_pct = st.chunk_rec_mean > 0 ? 100.0 * st.chunk_rec_max / st.chunk_rec_mean - 100.0 : 0.0;
    const int64_t xmax = *std::max_element(xcd_rec, xcd_rec + kXcds);
    st.xcd_imb_pct = st.records ? 100.0 * xmax * kXcds / st.records - 100.0 : 0.0;
    st.split_nnz_pct = nnz > 0 ? 100.0 * split_nnz / nnz : 0.0;
    st.pad_pct = nnz > 0 ? 100.0 * (st.records - nnz) / nnz : 0.0;
    // detector report + what was routed to the MFMA kernel
    const double all = static_cast<double>(p->nnz);
    st.tile_nnz_pct_10 = all > 0 ? 100.0 * p->tile_hist[0] / all : 0.0;
    st.tile_nnz_pct_25 = all > 0 ? 100.0 * p->tile_hist[1] / all : 0.0;
    st.tile_nnz_pct_50 = all > 0 ? 100.0 * p->tile_hist[2] / all : 0.0;
    st.tile_mean_fill = p->tile_cells > 0 ? all / (1024.0 * p->tile_cells) : 0.0;
    st.mfma_tiles = p->n_tiles;
    st.mfma_nnz_pct = all > 0 ? 100.0 * p->tile_nnz / all : 0.0;
    p->has_stats = true;
}

// ---- block-density detector (north_star: "MFMA only where RCM/Gorder reordering yields dense block-sparse tiles").
// The matrix is looked at in SCHEDULE coordinates: row tile = 32 consecutive rows of the schedule, column tile = 32
// consecutive column positions.  Every (row tile, column tile) pair with at least one entry is counted; `hist_nnz`
// reports which share of the nonzeros sits in tiles of fill >= 0.10 / 0.25 / 0.50 (the verdict `flex ... --stats`
// prints for every graph), and -- when `thr` > 0 -- tiles holding >= thr entries are taken OUT of the record stream
// (`in_tile[e - e_base] = 1`) and stored as dense fp32 blocks in the A-operand order of v_mfma_f32_32x32x2_f32.
struct DenseTiles {
    std::vector<float> a;           // [T][4][64][4]
    std::vector<uint32_t> boff;     // [T][32]
    std::vector<uint32_t> rt_ptr;   // [R+1]
    std::vector<uint32_t> rt_rows;  // [R][32]
    int64_t nnz = 0;                // entries moved into tiles
    int64_t hist_nnz[3] = {0, 0, 0};
    int64_t n_cells = 0;            // (row tile, column tile) pairs with at least one entry
};

int detect_dense_tiles(const flex_csr *A, int32_t r0, int32_t m, const std::vector<uint32_t> &sched, const std::vector<uint32_t> &colpos,
                       const int32_t *col_map, const int32_t *dst_map, bool off32, uint32_t row_bytes32, uint32_t thr, int64_t stride,
                       std::vector<uint8_t> &in_tile, DenseTiles &out) {  // stride > 1: look at every stride-th row tile only (thr must be 0)
    const uint32_t e_base = A->rowPtr[r0];
    const int64_t n_rt = (static_cast<int64_t>(m) + 31) / 32;
    constexpr int64_t kBlk = 64;  // row tiles per work item
    const int64_t nblk = (n_rt + kBlk - 1) / kBlk;
    struct Found {
        uint32_t rt, ct;
        std::vector<float> a;  // 1024, operand order
    };
    std::vector<std::vector<Found>> found(static_cast<size_t>(nblk));
    std::vector<int64_t> h0(static_cast<size_t>(nblk), 0), h1(h0), h2(h0), cells(h0), moved(h0);
    std::atomic<int> failed{0};
    parallel_chunks(nblk, [&](int64_t b) {
        try {
            std::vector<uint64_t> key;  // (column tile << 32) | (row in tile << 27) | index of the entry in the row
            std::vector<uint32_t> ebeg(33);
            for (int64_t rt = b * kBlk; rt < std::min(n_rt, (b + 1) * kBlk); ++rt) {
                if (rt % stride != 0) continue;
                key.clear();
                const int rows = static_cast<int>(std::min<int64_t>(32, m - rt * 32));
                bool fits = true;
                for (int i = 0; i < rows; ++i) {
                    const uint32_t r = sched[rt * 32 + i];
                    const uint32_t e0 = A->rowPtr[r], e1 = A->rowPtr[r + 1];
                    ebeg[i] = e0;
                    if (e1 - e0 >= (1u << 27)) fits = false;
                    for (uint32_t e = e0; e < e1 && fits; ++e) {
                        const uint32_t c = A->col[e];
                        const uint32_t cp = colpos.empty() ? c : colpos[c];
                        key.push_back((static_cast<uint64_t>(cp >> 5) << 32) | (static_cast<uint64_t>(i) << 27) | (e - e0));
                    }
                }
                if (!fits) continue;  // a row of >= 2^27 entries: left to the vector kernel
                std::sort(key.begin(), key.end());
                for (size_t z = 0; z < key.size();) {
                    size_t z1 = z;
                    while (z1 < key.size() && (key[z1] >> 32) == (key[z] >> 32)) ++z1;
                    const int64_t cnt = static_cast<int64_t>(z1 - z);
                    ++cells[b];
                    if (cnt * 10 >= 1024) h0[b] += cnt;
                    if (cnt * 4 >= 1024) h1[b] += cnt;
                    if (cnt * 2 >= 1024) h2[b] += cnt;
                    if (thr > 0 && cnt >= thr) {
                        Found f{static_cast<uint32_t>(rt), static_cast<uint32_t>(key[z] >> 32), std::vector<float>(1024, 0.f)};
                        uint8_t taken[32][32] = {};
                        for (size_t y = z; y < z1; ++y) {
                            const int i = static_cast<int>((key[y] >> 27) & 31);
                            const uint32_t e = ebeg[i] + static_cast<uint32_t>(key[y] & ((1u << 27) - 1));
                            const uint32_t c = A->col[e];
                            const int j = static_cast<int>((colpos.empty() ? c : colpos[c]) & 31);
                            if (taken[i][j]) continue;  // a duplicate (row, col) entry stays with the vector kernel
                            taken[i][j] = 1;
                            const int kk = j >> 1, lane = i + 32 * (j & 1);
                            f.a[((kk >> 2) * 64 + lane) * 4 + (kk & 3)] = A->vals[e];
                            in_tile[e - e_base] = 1;
                            ++moved[b];
                        }
                        found[static_cast<size_t>(b)].push_back(std::move(f));
                    }
                    z = z1;
                }
            }
        } catch (...) {
            failed.store(1);
        }
    });
    if (failed.load()) return FLEX_ERR_NOMEM;
    for (int64_t b = 0; b < nblk; ++b) {
        out.hist_nnz[0] += h0[b];
        out.hist_nnz[1] += h1[b];
        out.hist_nnz[2] += h2[b];
        out.n_cells += cells[b];
        out.nnz += moved[b];
    }
    if (thr == 0 || out.nnz == 0) return FLEX_OK;
    // tiles in (row tile, column tile) order
    const uint32_t n_cols = static_cast<uint32_t>(A->n);
    try {
        for (auto &blk : found)
            for (Found &f : blk) {
                out.a.insert(out.a.end(), f.a.begin(), f.a.end());
                f.a = std::vector<float>();
                for (uint32_t j = 0; j < 32; ++j) {
                    uint32_t pos = f.ct * 32 + j;
                    if (pos >= n_cols) pos = f.ct * 32;  // hangs over the last column: any valid row, its A entries are zero
                    uint32_t c = colpos.empty() ? pos : sched[pos];
                    if (col_map) c = static_cast<uint32_t>(col_map[c]);
                    out.boff.push_back(off32 ? c * row_bytes32 : c);
                }
            }
    } catch (const std::bad_alloc &) {
        return FLEX_ERR_NOMEM;
    }
    // row-tile directory (second pass over the found list, which is already in (rt, ct) order)
    out.rt_ptr.clear();
    uint32_t t = 0, last_rt = 0xFFFFFFFFu;
    for (auto &blk : found)
        for (Found &f : blk) {
            if (f.rt != last_rt) {
                out.rt_ptr.push_back(t);
                for (uint32_t i = 0; i < 32; ++i) {
                    const int64_t sp = static_cast<int64_t>(f.rt) * 32 + i;
                    uint32_t dst = 0xFFFFFFFFu;
                    if (sp < m) {
                        const uint32_t r = sched[sp];
                        dst = dst_map ? static_cast<uint32_t>(dst_map[r]) : r - static_cast<uint32_t>(r0);
                    }
                    out.rt_rows.push_back(dst);
                }
                last_rt = f.rt;
            }
            ++t;
        }
    out.rt_ptr.push_back(t);
    return FLEX_OK;
}

// Rows [r0,r1) of A.  col_map: B row read by column c (NULL = c).  dst_map: C row
// written by row r (NULL = r - r0, i.e. slice-local).
// sched_cache (or NULL): holds the row schedule once it has been computed, so that several candidate plans of
// one matrix (autotune) order it only once.  force_G (or 0): lanes per record instead of the degree rule.
int build_plan(flex_plan *p, const flex_csr *A, int32_t r0, int32_t r1, const int32_t *col_map,
               const int32_t *dst_map, unsigned flags, std::vector<uint32_t> *sched_cache = nullptr, int force_G = 0) {
    const int32_t m = r1 - r0;
    const int k = p->k;
    const bool timing = std::getenv("FLEX_PLAN_TIMING") != nullptr;  // phase times of the planner on stderr
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "plan: %-22s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    const unsigned order = flags & FLEX_ORDER_MASK;
    if (order > FLEX_ORDER_GORDER) return FLEX_ERR_INVALID;
    // graph orderings need the whole square matrix
    if (order != FLEX_ORDER_NATURAL && (A->m != A->n || r0 != 0 || r1 != A->m)) return FLEX_ERR_INVALID;
    p->order = order;

    // G lanes x float4 cover one column tile of 4*G columns; k wider than that runs as several tiles
    // (blockIdx.y, dispatched one after the other).  Widest tile (fewest instructions per byte) for
    // low-degree graphs; high-degree graphs are bound by L2-miss traffic instead -- the B rows touched by
    // the resident waves (waves x records x 16*G bytes) overflow the 4 MiB L2s -- and a narrower tile
    // shrinks that footprint at the price of re-reading the records once per tile.  Measured on MI355X,
    // k=128: reddit-like generator, G=16 vs 32: -7 % at degree 12, +7 % at 24, +15 % at
    // 36..100; amazon shape +12 % (G=16), +16 % (G=8); flickr (degree 11) -7 %, yelp (19.5) -2.5 %.
    const double avg_deg = m > 0 ? static_cast<double>(A->rowPtr[r1] - A->rowPtr[r0]) / m : 0.0;
    int G = 8;
    while (4 * G < k && G < 64) G <<= 1;
    if (force_G) {
        G = std::min(G, force_G);
    } else if (const long g_env = env_long("FLEX_LANES", 0); g_env == 8 || g_env == 16 || g_env == 32 || g_env == 64) {
        G = std::min<int>(G, static_cast<int>(g_env));  // tuning experiments
    } else {
        G = std::min(G, 32);  // k = 256 as two 128-column tiles beats one 256-column tile on every shape measured
        if (avg_deg >= 24.0) G = std::min(G, 16);
        if (avg_deg >= 128.0) G = std::min(G, 8);
    }
    p->lanes_per_nz = G;
    p->off32 = static_cast<uint64_t>(A->n) * static_cast<uint64_t>(p->ldb) * 4u <= (uint64_t(1) << 32);

    // schedule: sched[i] = row of A processed i-th
    std::vector<uint32_t> sched_local;
    std::vector<uint32_t> &sched = sched_cache ? *sched_cache : sched_local;
    if (sched_cache && sched.size() == static_cast<size_t>(m) && m > 0) {
        // computed by an earlier candidate of the same matrix
    } else if (sched.assign(static_cast<size_t>(m), 0u); order != FLEX_ORDER_NATURAL) {
        std::vector<uint32_t> rank;
        int rc = order == FLEX_ORDER_RCM       ? order_rcm_host(m, A->rowPtr, A->col, rank)
                 : order == FLEX_ORDER_CLUSTER ? order_cluster_host(m, A->rowPtr, A->col, rank)
                                               : order_gorder_host(m, A->rowPtr, A->col, 3, rank);
        if (rc) {
            sched.clear();
            return rc;
        }
        for (int32_t r = 0; r < m; ++r) sched[rank[r]] = static_cast<uint32_t>(r);
    } else {
        std::iota(sched.begin(), sched.end(), static_cast<uint32_t>(r0));
    }

    lap("row schedule");
    // position of a column's vertex in the schedule (what "near" means for a reordered square matrix); for a
    // natural-order plan, a mapped plan or a row shard the column ids of A are positions already
    std::vector<uint32_t> colpos;
    if (order != FLEX_ORDER_NATURAL) {
        colpos.resize(static_cast<size_t>(m));
        for (int32_t i = 0; i < m; ++i) colpos[sched[i]] = static_cast<uint32_t>(i);
    }

    // ---- dense tiles -> MFMA kernel (FLEX_MFMA: 1 = route tiles of fill >= FLEX_MFMA_FILL %, 2 = never; default:
    // route when a sampled look at every 64th row tile finds at least 10 % of the nonzeros in such tiles -- most
    // graphs have none and then pay 1/64 of one pass).  With FLEX_PLAN_STATS the detector looks at every tile, so
    // that its report (share of nonzeros in tiles of fill >= 0.10 / 0.25 / 0.50) is exact.
    // Default threshold 60 %: measured on MI355X at k = 128 (tools/probe_mfma.py, 64-row diagonal blocks + 8 random
    // entries per row, 200 K rows): routing blocks of fill 0.9 takes 297 -> 220 us, fill 0.6 228 -> 219 (break-even),
    // fill 0.3 160 -> 217 (slower: a tile costs the same whatever its fill, and its 32 C rows are read and written
    // once more), DESIGN.md 3.5.
    const flex_csr *const A_in = A;
    flex_csr A_f{};
    std::vector<uint32_t> f_rowptr, f_col;
    std::vector<float> f_vals;
    DenseTiles tiles;
    {
        const long mode_mfma = env_long("FLEX_MFMA", 0);
        const uint32_t fill_pct = static_cast<uint32_t>(std::clamp<long>(env_long("FLEX_MFMA_FILL", 60), 1, 100));
        const uint32_t thr = (1024u * fill_pct + 99u) / 100u;
        const bool report = (flags & FLEX_PLAN_STATS) != 0;
        const int64_t nnz_in = static_cast<int64_t>(A->rowPtr[r1]) - A->rowPtr[r0];
        bool route = mode_mfma == 1;
        const uint32_t row_bytes32_t = static_cast<uint32_t>(p->ldb) * 4u;
        std::vector<uint8_t> in_tile;
        int rc_t = FLEX_OK;
        if (mode_mfma != 1 && mode_mfma != 2 && m >= 2048 && nnz_in >= (1 << 16)) {  // the sampled look
            DenseTiles probe;
            rc_t = detect_dense_tiles(A, r0, m, sched, colpos, col_map, dst_map, p->off32, row_bytes32_t, 0, 64, in_tile, probe);
            if (rc_t) return rc_t;
            const int64_t share = fill_pct <= 10 ? probe.hist_nnz[0] : fill_pct <= 25 ? probe.hist_nnz[1] : probe.hist_nnz[2];  // >= 0.5 also screens for 0.6
            route = share * 64 * 10 >= nnz_in;  // >= 10 % of the nonzeros, extrapolated from the sample (amazon shape: 2.3 % in
                                                //    such tiles; routing them changed nothing, 9.08 vs 9.08 ms, and cost 1.1 s of planning)
        }
        if (route || report) {
            try {
                if (route) in_tile.assign(static_cast<size_t>(nnz_in), 0);
            } catch (const std::bad_alloc &) {
                return FLEX_ERR_NOMEM;
            }
            rc_t = detect_dense_tiles(A, r0, m, sched, colpos, col_map, dst_map, p->off32, row_bytes32_t, route ? thr : 0, 1, in_tile, tiles);
            if (rc_t) return rc_t;
            p->tile_hist[0] = tiles.hist_nnz[0];
            p->tile_hist[1] = tiles.hist_nnz[1];
            p->tile_hist[2] = tiles.hist_nnz[2];
            p->tile_cells = tiles.n_cells;
            p->tile_hist_valid = true;
        }
        if (tiles.nnz > 0) {  // the vector kernel gets A minus the entries that moved into tiles (same rows, same ids)
            try {
                f_rowptr.assign(static_cast<size_t>(A->m) + 1, 0u);
                f_col.resize(static_cast<size_t>(nnz_in - tiles.nnz));
                f_vals.resize(static_cast<size_t>(nnz_in - tiles.nnz));
            } catch (const std::bad_alloc &) {
                return FLEX_ERR_NOMEM;
            }
            const uint32_t eb = A->rowPtr[r0];
            uint32_t o = 0;
            for (int32_t r = 0; r < A->m; ++r) {
                f_rowptr[r] = o;
                if (r < r0 || r >= r1) continue;
                for (uint32_t e = A->rowPtr[r]; e < A->rowPtr[r + 1]; ++e)
                    if (!in_tile[e - eb]) {
                        f_col[o] = A->col[e];
                        f_vals[o] = A->vals[e];
                        ++o;
                    }
            }
            f_rowptr[A->m] = o;
            A_f = flex_csr{A->m, A->n, static_cast<int64_t>(o), f_rowptr.data(), f_col.data(), f_vals.data()};
            A = &A_f;
        }
        lap("dense-tile detector");
    }
    (void)A_in;
    // chunk budget in records; rows longer than one budget are cut into pieces
    // chunk budget: short chunks keep the dispatcher's load balancing fine-grained on low-degree
    // graphs (flickr: best at ~96 records), long ones amortise the per-chunk descriptor chain on
    // high-degree graphs (reddit: best at >= 256).  Measured on MI355X, DESIGN.md 3.3.
    // (k <= 32, G = 8: a step consumes 8 records, so the same number of steps needs more records per chunk --
    //  flickr k=32 best at 192, ppi 192, yelp 256, pubmed 128; 7-23 % over the k=128 rule)
    // (round 2, G = 8 again: the upper clamp was 256; amazon shape k=128 9.08 -> 8.87 ms and k=32 2.34 -> 2.26 ms at 512,
    //  reddit k=32 174 -> 165 us at 512, yelp k=32 131 -> 126 us at 384 (its rule value: 16 x 19.5 = 312); 768-1024 lose
    //  it again; G = 16 (reddit k=128) is flat from 256 to 512 and keeps 256)
    const long lo_budget = G <= 8 ? 128 : 96;
    long auto_budget = std::clamp<long>(static_cast<long>((G <= 8 ? 16.0 : 8.0) * avg_deg), lo_budget, G <= 8 ? 512 : 256);
    // small inputs: keep at least ~2048 chunks (two waves per SIMD) before growing them (wiki-Vote shape, k=32:
    // 5.5 us at 128-160 records per chunk, 6.3 at 200)
    auto_budget = std::min(auto_budget, std::max<long>(lo_budget, static_cast<long>(A->rowPtr[r1] - A->rowPtr[r0]) / 2048));
    const uint32_t wave_nnz = static_cast<uint32_t>(env_long("FLEX_WAVE_NNZ", auto_budget));
    const uint32_t row_cost = static_cast<uint32_t>(env_long("FLEX_ROW_COST", 16));
    p->xcd_remap = env_long("FLEX_XCD_REMAP", 1) != 2;  // 2 = off (tuning experiments only)
    p->lds_extra = static_cast<unsigned>(env_long("FLEX_LDS_EXTRA", 1)) & ~15u;  // default 0 (1 -> 0)
    // The record stream is read once per column tile.  Non-temporal loads keep it from displacing B rows in the L2s and
    // the Infinity Cache, but they also come back slower and sit on the header -> records -> gathers chain of every chunk.
    // Measured on MI355X (same box each):
    //   several tiles (k > 4G), stream >= 32 MB:  amazon shape k=128 9.48 -> 9.00 ms, reddit 687 -> 679 us, yelp 524 -> 515 us
    //   one tile, stream of 0.1-0.2 GB:           reddit k=32 178 -> 196 us, yelp k=32 130 -> 148 us   (worse)
    //   one tile, stream of 2.1 GB (8x the Infinity Cache): amazon k=32 2.43 -> 2.29 ms
    //   small streams:                            flickr k=128 37.9 -> 40.4 us                          (worse)
    // hence: on for multi-tile launches from 32 MB, for single-tile launches only from 1 GiB.  FLEX_REC_NT = 1 / 2 forces.
    {
        const long nt_env = env_long("FLEX_REC_NT", 0);
        const int ktiles_nt = (k + 4 * G - 1) / (4 * G);
        const uint64_t stream_bytes = static_cast<uint64_t>(A->rowPtr[r1] - A->rowPtr[r0]) * 8u;
        p->rec_nt = nt_env == 1 || (nt_env != 2 && stream_bytes >= (ktiles_nt >= 2 ? (32ull << 20) : (1ull << 30)));
    }
    p->unroll = static_cast<int>(env_long("FLEX_U", 0));
    const uint32_t S = 64u / static_cast<uint32_t>(G);      // records per step: rows are padded to it
    // Rows longer than one budget are cut into pieces of one budget, each a chunk of its own that
    // writes a k-wide partial sum: one wave keeps only U gathers in flight, so a long row is much
    // faster as several concurrent pieces (flickr, MI355X: 88 us with no splitting, 43 us with
    // rows > 192 records split, 40 us with rows > 96 split; reddit is flat from 256 to 512;
    // DESIGN.md 3.3).  Pieces stay in schedule order: moving them to the
    // front of the XCD slices helped flickr by 3 % and cost reddit 12 % (half its chunks are pieces).
    const uint32_t long_row = static_cast<uint32_t>(env_long("FLEX_LONG_ROW", wave_nnz));
    const uint32_t piece_len = std::max<uint32_t>(S, static_cast<uint32_t>(env_long("FLEX_PIECE", wave_nnz)) / S * S);

    // ---- column panels (the 2-D schedule; ≙ the column spans of csr2_DiagTiling's rounds 2-3, mat.cu:680-942, and
    // csr2seg_Cmajor, mat.cu:1192-1269, re-thought for eight private 4 MiB L2s).  An XCD walks ONE contiguous slice of
    // the rows; in 1-D that slice is walked row by row and the B rows it needs within +-w communities (megabytes)
    // are evicted between uses.  In 2-D the slice is walked PHASE by PHASE: phase q holds, for every row of the
    // slice, the records whose column lies in panel q of B (P rows = `panel_bytes` of one column tile, about half an
    // L2), so whatever the resident waves gather at one time comes from one or two panels and hits the L2 by
    // construction; the price is that a row with records in several phases is summed from several pieces (a k-wide
    // partial sum written and read once per piece).  Only runs of >= `seg_min` records of a row in one panel become a
    // piece; the rest of the row (its scattered columns, which miss either way) is ONE more piece in a last phase.
    const int64_t slice_nnz = static_cast<int64_t>(A->rowPtr[r1]) - A->rowPtr[r0];
    const uint64_t tile_bytes = 16ull * static_cast<uint64_t>(G);  // one B row of one column tile
    const long mode_2d = env_long("FLEX_2D", 0);                    // 1 = on, 2 = off, otherwise the rule below
    bool two_d = mode_2d == 1 && m > 0;  // forced (tests, tuning): any size
    if (mode_2d != 1 && mode_2d != 2) two_d = false;  // rule: decided by measurement
    const uint64_t panel_bytes = static_cast<uint64_t>(env_long("FLEX_PANEL_KB", 2048)) << 10;
    uint32_t pshift = 0;
    while ((2ull << pshift) * tile_bytes <= panel_bytes) ++pshift;  // P = 2^pshift rows of B per panel
    const uint32_t seg_min = static_cast<uint32_t>(env_long("FLEX_SEG_MIN", 4));
    constexpr uint32_t kFarPhase = 0xFFFFFFFEu;

    struct Piece {
        uint32_t spos;       // position of the piece's row in the schedule
        uint32_t beg, end;   // its records [beg,end) in rcol/rval
        uint32_t phase;      // 0 in 1-D; column panel + 1, or kFarPhase, in 2-D
        uint32_t own_chunk;  // a slice of a run longer than one budget: a chunk of its own
    };
    std::vector<Piece> pieces;
    std::vector<uint32_t> row_first_piece(static_cast<size_t>(m) + 1, 0u);
    std::vector<uint32_t> pcol;  // 2-D: the records of every row re-grouped by piece (index e - e_base)
    std::vector<float> pval;
    const uint32_t e_base = A->rowPtr[r0];
    const uint32_t *rcol = A->col;
    const float *rval = A->vals;
    uint32_t r_off = 0;  // rcol[e - r_off]
    // a run of `len` records as pieces: one, or (longer than a budget) several of about one budget.  The last piece to
    // arrive sums all of them with ONE wave, so a hub of 10^6 nonzeros cut into 10^4 budget-sized pieces spent 0.9 ms
    // in that sum alone (tools/probe_hub.py: 1263 us against 358 us without the hub): at most 256 pieces per run,
    // longer ones instead: 564 us (774 pieces: 601 us).
    auto cut_run = [&](std::vector<Piece> &out, uint32_t spos, uint32_t b, uint32_t e, uint32_t phase) {
        const uint32_t len = e - b;
        if (len <= long_row) {
            out.push_back({spos, b, e, phase, 0u});
            return;
        }
        constexpr uint32_t kMaxPieces = 256;
        const uint32_t nchunk = std::min<uint32_t>((len + piece_len - 1) / piece_len, kMaxPieces);
        const uint32_t per = ((len + nchunk - 1) / nchunk + S - 1) / S * S;  // whole steps
        for (uint32_t c0 = b; c0 < e; c0 += per) out.push_back({spos, c0, std::min(e, c0 + per), phase, 1u});
    };
    try {
        if (!two_d) {
            pieces.reserve(static_cast<size_t>(m) + 1024);
            for (int32_t i = 0; i < m; ++i) {
                const uint32_t r = sched[i];
                row_first_piece[i] = static_cast<uint32_t>(pieces.size());
                cut_run(pieces, static_cast<uint32_t>(i), A->rowPtr[r], A->rowPtr[r + 1], 0u);
            }
            row_first_piece[m] = static_cast<uint32_t>(pieces.size());
        } else {
            pcol.resize(static_cast<size_t>(slice_nnz));
            pval.resize(static_cast<size_t>(slice_nnz));
            constexpr int64_t kBlk = 1024;  // schedule positions per work item
            const int64_t nblk = (m + kBlk - 1) / kBlk;
            std::vector<std::vector<Piece>> blk(static_cast<size_t>(nblk));
            std::vector<uint32_t> row_np(static_cast<size_t>(m), 0u);
            std::atomic<int> failed{0};
            parallel_chunks(nblk, [&](int64_t b) {
                try {
                    std::vector<Piece> &out = blk[static_cast<size_t>(b)];
                    std::vector<uint64_t> key;  // (panel << 32) | index within the row
                    std::vector<uint32_t> run_beg, run_pan;
                    for (int64_t i = b * kBlk; i < std::min<int64_t>(m, (b + 1) * kBlk); ++i) {
                        const uint32_t r = sched[i];
                        const uint32_t e0 = A->rowPtr[r], e1 = A->rowPtr[r + 1], len = e1 - e0;
                        const size_t before = out.size();
                        const uint32_t o0 = e0 - e_base;
                        if (len == 0) {
                            out.push_back({static_cast<uint32_t>(i), o0, o0, kFarPhase, 0u});
                            row_np[i] = 1;
                            continue;
                        }
                        key.resize(len);
                        bool sorted = true;
                        for (uint32_t z = 0; z < len; ++z) {
                            const uint32_t c = A->col[e0 + z];
                            const uint32_t pan = (colpos.empty() ? c : colpos[c]) >> pshift;
                            key[z] = (static_cast<uint64_t>(pan) << 32) | z;
                            sorted = sorted && (z == 0 || key[z - 1] <= key[z]);
                        }
                        if (!sorted) std::sort(key.begin(), key.end());  // by panel, original order within a panel
                        run_beg.clear();
                        run_pan.clear();
                        for (uint32_t z = 0; z < len; ++z)
                            if (z == 0 || (key[z] >> 32) != (key[z - 1] >> 32)) {
                                run_beg.push_back(z);
                                run_pan.push_back(static_cast<uint32_t>(key[z] >> 32));
                            }
                        run_beg.push_back(len);
                        // layout of the row in pcol/pval: the kept runs in panel order, then everything else
                        uint32_t o = o0, n_far = 0, n_kept = 0;
                        for (size_t q = 0; q + 1 < run_beg.size(); ++q) {
                            const uint32_t cnt = run_beg[q + 1] - run_beg[q];
                            if (cnt < seg_min) {
                                n_far += cnt;
                                continue;
                            }
                            for (uint32_t z = run_beg[q]; z < run_beg[q + 1]; ++z) {
                                const uint32_t e = e0 + static_cast<uint32_t>(key[z] & 0xFFFFFFFFu);
                                pcol[o] = A->col[e];
                                pval[o] = A->vals[e];
                                ++o;
                            }
                            ++n_kept;
                            cut_run(out, static_cast<uint32_t>(i), o - cnt, o, run_pan[q] + 1);
                        }
                        const uint32_t far_beg = o;
                        if (n_far)
                            for (size_t q = 0; q + 1 < run_beg.size(); ++q) {
                                if (run_beg[q + 1] - run_beg[q] >= seg_min) continue;
                                for (uint32_t z = run_beg[q]; z < run_beg[q + 1]; ++z) {
                                    const uint32_t e = e0 + static_cast<uint32_t>(key[z] & 0xFFFFFFFFu);
                                    pcol[o] = A->col[e];
                                    pval[o] = A->vals[e];
                                    ++o;
                                }
                            }
                        if (n_far) {
                            // a remainder too short to pay for a piece of its own rides with the row's last kept run
                            if (n_far < seg_min && n_kept > 0 && out.back().own_chunk == 0) out.back().end = o;
                            else cut_run(out, static_cast<uint32_t>(i), far_beg, o, kFarPhase);
                        }
                        row_np[i] = static_cast<uint32_t>(out.size() - before);
                    }
                } catch (...) {
                    failed.store(1);
                }
            });
            if (failed.load()) return FLEX_ERR_NOMEM;
            size_t total = 0;
            for (int32_t i = 0; i < m; ++i) {
                row_first_piece[i] = static_cast<uint32_t>(total);
                total += row_np[i];
            }
            row_first_piece[m] = static_cast<uint32_t>(total);
            if (total >= (size_t(1) << 31)) return FLEX_ERR_UNSUPPORTED;
            pieces.resize(total);
            parallel_chunks(nblk, [&](int64_t b) {
                const std::vector<Piece> &src = blk[static_cast<size_t>(b)];
                if (!src.empty()) std::copy(src.begin(), src.end(), pieces.begin() + row_first_piece[b * kBlk]);
            });
            rcol = pcol.data();
            rval = pval.data();
            r_off = 0;  // piece ranges are already relative to e_base
        }
    } catch (const std::bad_alloc &) {
        return FLEX_ERR_NOMEM;
    }
    const size_t n_pieces = pieces.size();
    lap("pieces");

    // ---- XCD slices of the rows (2-D: cut here, by records + row overhead; 1-D: the chunk table is cut by cost below)
    uint32_t slice_row[kXcds + 1] = {0};
    slice_row[kXcds] = static_cast<uint32_t>(m);
    if (two_d) {
        std::vector<uint64_t> cum(static_cast<size_t>(m) + 1, 0);
        for (int32_t i = 0; i < m; ++i) {
            const uint32_t r = sched[i];
            cum[i + 1] = cum[i] + (A->rowPtr[r + 1] - A->rowPtr[r]) + row_cost;
        }
        for (int x = 1; x < kXcds; ++x)
            slice_row[x] = std::max<uint32_t>(slice_row[x - 1], static_cast<uint32_t>(std::lower_bound(cum.begin(), cum.end(), cum[m] * x / kXcds) - cum.begin()));
    }

    // ---- emission order: 1-D = schedule order; 2-D = per slice, phase by phase (rows in schedule order inside a phase)
    std::vector<uint32_t> emit(n_pieces);
    std::iota(emit.begin(), emit.end(), 0u);
    if (two_d) {
        parallel_chunks(kXcds, [&](int64_t x) {
            auto b = emit.begin() + row_first_piece[slice_row[x]], e = emit.begin() + row_first_piece[slice_row[x + 1]];
            std::stable_sort(b, e, [&](uint32_t u, uint32_t v) { return pieces[u].phase < pieces[v].phase; });
        });
    }
    lap("emission order");

    // ---- rows summed from several pieces: partial slots (consecutive per row, in piece order) + arrival bookkeeping
    std::vector<SplitRow> split;
    std::vector<uint32_t> row_sidx(static_cast<size_t>(m), 0u), row_first_partial(static_cast<size_t>(m), 0u);
    uint32_t n_partials = 0;
    int64_t split_nnz = 0;
    try {
        for (int32_t i = 0; i < m; ++i) {
            const uint32_t np = row_first_piece[i + 1] - row_first_piece[i];
            if (np <= 1) continue;
            const uint32_t r = sched[i];
            const uint32_t dst = dst_map ? static_cast<uint32_t>(dst_map[r]) : r - static_cast<uint32_t>(r0);
            row_sidx[i] = static_cast<uint32_t>(split.size());
            row_first_partial[i] = n_partials;
            split.push_back({dst, n_partials, np});
            n_partials += np;
            split_nnz += A->rowPtr[r + 1] - A->rowPtr[r];
        }
    } catch (const std::bad_alloc &) {
        return FLEX_ERR_NOMEM;
    }

    // ---- tasks (one per piece, in emission order), their records, and the chunks (one wave each)
    const uint32_t n_tasks = static_cast<uint32_t>(n_pieces);
    std::vector<uint32_t> t_beg, t_dst, w_task;
    std::vector<uint2> t_aux, rec;
    uint32_t slice_chunk[kXcds + 1] = {0};
    constexpr uint32_t kMaxTasksPerWave = 63;  // the kernel hands descriptors out by lane (compute_chunk)
    try {
        t_beg.resize(static_cast<size_t>(n_tasks) + 1);
        t_dst.resize(n_tasks);
        t_aux.resize(n_tasks);
        uint64_t pos = 0;
        uint32_t wave_cost = 0, cur_phase = 0, prev_own = 0;
        int cur_slice = 0;
        for (uint32_t t = 0; t < n_tasks; ++t) {
            const Piece &pc = pieces[emit[t]];
            const uint32_t len = pc.end - pc.beg;
            bool fresh = w_task.empty() || wave_cost >= wave_nnz || t - w_task.back() >= kMaxTasksPerWave || pc.own_chunk || prev_own;
            if (two_d) {
                while (pc.spos >= slice_row[cur_slice + 1]) {  // first task of the next XCD slice
                    slice_chunk[++cur_slice] = static_cast<uint32_t>(w_task.size());
                    fresh = true;
                }
                if (pc.phase != cur_phase) fresh = true;  // a chunk never straddles two panels
                cur_phase = pc.phase;
            }
            if (fresh) {
                w_task.push_back(t);
                wave_cost = 0;
            }
            wave_cost += len + row_cost;
            prev_own = pc.own_chunk;
            t_beg[t] = static_cast<uint32_t>(pos);
            pos += (len + S - 1) / S * S;  // padded to a whole number of steps
            if (pos >= (uint64_t(1) << 32)) return FLEX_ERR_UNSUPPORTED;  // 32-bit record offsets
        }
        t_beg[n_tasks] = static_cast<uint32_t>(pos);
        w_task.push_back(n_tasks);
        if (m == 0) w_task.assign(1, 0u);
        if (two_d)
            while (cur_slice < kXcds) slice_chunk[++cur_slice] = static_cast<uint32_t>(w_task.size() - 1);
        rec.resize(static_cast<size_t>(pos));
    } catch (const std::bad_alloc &) {
        return FLEX_ERR_NOMEM;
    }
    const uint32_t row_bytes32 = static_cast<uint32_t>(p->ldb) * 4u;
    {
        constexpr int64_t kTaskBlk = 4096;
        parallel_chunks((static_cast<int64_t>(n_tasks) + kTaskBlk - 1) / kTaskBlk, [&](int64_t b) {
            for (int64_t t = b * kTaskBlk; t < std::min<int64_t>(n_tasks, (b + 1) * kTaskBlk); ++t) {
                const uint32_t pi = emit[t];
                const Piece &pc = pieces[pi];
                const uint32_t i = pc.spos, r = sched[i];
                const uint32_t np = row_first_piece[i + 1] - row_first_piece[i];
                if (np > 1) {
                    t_dst[t] = kPartialFlag | (row_first_partial[i] + (pi - row_first_piece[i]));
                    t_aux[t] = make_uint2(row_sidx[i], np);
                } else {
                    t_dst[t] = dst_map ? static_cast<uint32_t>(dst_map[r]) : r - static_cast<uint32_t>(r0);
                    t_aux[t] = make_uint2(0u, 0u);
                }
                uint2 *o = rec.data() + t_beg[t];
                for (uint32_t e = pc.beg; e < pc.end; ++e) {
                    uint32_t c = rcol[e - r_off];
                    if (col_map) c = static_cast<uint32_t>(col_map[c]);
                    uint32_t bits;
                    std::memcpy(&bits, &rval[e - r_off], 4);
                    *o++ = make_uint2(p->off32 ? c * row_bytes32 : c, bits);
                }
                // pad to a whole number of steps: value 0, B row = the last real one (always a valid address)
                for (uint2 *end = rec.data() + t_beg[t + 1]; o < end; ++o) *o = make_uint2(o[-1].x, 0u);
            }
        });
    }
    pcol = std::vector<uint32_t>();
    pval = std::vector<float>();
    lap("records and tasks");

    p->n_tasks = n_tasks;
    p->n_records = rec.size();
    p->c_rows = dst_map ? A->m : m;
    p->n_chunks = static_cast<uint32_t>(w_task.size() - 1);
    p->n_split = static_cast<uint32_t>(split.size());
    p->n_partials = n_partials;
    p->two_d = two_d;
    p->panel_rows = 1u << pshift;

    int rc;
    if ((rc = upload(&p->d_rec, rec, &p->device_bytes))) return rc;
    if ((rc = upload(&p->d_t_beg, t_beg, &p->device_bytes))) return rc;
    if ((rc = upload(&p->d_t_dst, t_dst, &p->device_bytes))) return rc;
    if ((rc = upload(&p->d_t_aux, t_aux, &p->device_bytes))) return rc;
    p->n_tiles = static_cast<uint32_t>(tiles.boff.size() / 32);
    p->n_row_tiles = tiles.rt_ptr.empty() ? 0u : static_cast<uint32_t>(tiles.rt_ptr.size() - 1);
    p->tile_nnz = tiles.nnz;
    if (p->n_tiles) {
        if ((rc = upload(&p->d_tile_a, tiles.a, &p->device_bytes))) return rc;
        if ((rc = upload(&p->d_tile_boff, tiles.boff, &p->device_bytes))) return rc;
        if ((rc = upload(&p->d_rt_ptr, tiles.rt_ptr, &p->device_bytes))) return rc;
        if ((rc = upload(&p->d_rt_rows, tiles.rt_rows, &p->device_bytes))) return rc;
    }
    lap("upload records/tasks");
    // Chunk table in launch order: the kernel gives XCD x the x-th eighth of it.  The eighths are cut
    // by COST (records + per-row and per-chunk overhead), not by chunk count, and padded with empty
    // chunks to a common length: schedules that put the heavy rows at one end (degree order, RCM,
    // Gorder) otherwise leave one XCD with up to 1.9x the mean work (flickr shape).
    // (2-D: the eighths are the row slices cut above -- a slice's phases must stay on one XCD.)
    const uint32_t n_real = static_cast<uint32_t>(w_task.size() - 1);
    auto header = [&](uint32_t c) {
        return make_uint4(w_task[c], w_task[c + 1] - w_task[c], t_beg[w_task[c]], t_beg[w_task[c + 1]]);
    };
    std::vector<uint4> chunk;
    if (two_d || (p->xcd_remap && n_real >= 8u * kXcds * kWavesPerBlock && env_long("FLEX_XCD_BALANCE", 1) != 2)) {
        uint32_t cut[kXcds + 1];
        cut[0] = 0;
        cut[kXcds] = n_real;
        if (two_d) {
            for (uint32_t x = 1; x < kXcds; ++x) cut[x] = slice_chunk[x];
        } else {
            // cost of a chunk in units of one 512-byte gather (a record at k = 128): measured per-XCD times
            // on the flickr shape fit  t = a * records + ~20 a * chunks  with rows nearly free
            const uint64_t chunk_cost = static_cast<uint64_t>(env_long("FLEX_CHUNK_COST", 16)) * 32u;
            const uint64_t task_cost = static_cast<uint64_t>(env_long("FLEX_TASK_COST", 2)) * 32u;
            std::vector<uint64_t> cum(n_real + 1, 0);
            for (uint32_t c = 0; c < n_real; ++c) {
                const uint4 h = header(c);
                cum[c + 1] = cum[c] + static_cast<uint64_t>(h.w - h.z) * static_cast<uint32_t>(G) + task_cost * h.y + chunk_cost;
            }
            for (uint32_t x = 1; x < kXcds; ++x) {
                const uint64_t want = cum[n_real] * x / kXcds;
                uint32_t c = static_cast<uint32_t>(std::lower_bound(cum.begin(), cum.end(), want) - cum.begin());
                c = (c + kWavesPerBlock / 2) / kWavesPerBlock * kWavesPerBlock;  // whole workgroups
                cut[x] = std::clamp(c, cut[x - 1], n_real);
            }
        }
        uint32_t longest = 0;
        for (uint32_t x = 0; x < kXcds; ++x) longest = std::max(longest, cut[x + 1] - cut[x]);
        longest = (longest + kWavesPerBlock - 1) / kWavesPerBlock * kWavesPerBlock;
        chunk.assign(static_cast<size_t>(longest) * kXcds, make_uint4(0u, 0u, 0u, 0u));  // empty: no tasks, no records
        for (uint32_t x = 0; x < kXcds; ++x)
            for (uint32_t c = cut[x]; c < cut[x + 1]; ++c) chunk[static_cast<size_t>(x) * longest + (c - cut[x])] = header(c);
    } else {
        chunk.resize(n_real);
        for (uint32_t c = 0; c < n_real; ++c) chunk[c] = header(c);
    }
    p->n_chunks = n_real;
    p->n_slots = static_cast<uint32_t>(chunk.size());
    if ((rc = upload(&p->d_chunk, chunk, &p->device_bytes))) return rc;
    if (flags & FLEX_PLAN_STATS) {
        try {
            collect_stats(p, rec, chunk, split_nnz);
        } catch (const std::bad_alloc &) {
            return FLEX_ERR_NOMEM;
        }
    }
    if ((rc = upload(&p->d_split, split, &p->device_bytes))) return rc;
    const size_t ktiles = (static_cast<size_t>(k) + 4 * G - 1) / (4 * G);
    std::vector<uint32_t> zeros(std::max<size_t>(1, split.size() * ktiles), 0u);
    if ((rc = upload(&p->d_split_cnt, zeros, &p->device_bytes))) return rc;
    const size_t pbytes = std::max<size_t>(1, static_cast<size_t>(n_partials) * k) * sizeof(float);
    p->fused_fixup = env_long("FLEX_FUSED_FIXUP", 1) == 1;
    FLEX_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_partial), pbytes));
    p->device_bytes += static_cast<int64_t>(pbytes);
    lap("chunk table, stats");
    return FLEX_OK;
}

}  // namespace

extern "C" int flex_spmm(flex_plan *p, const float *dB, float *dC, flex_stream_t stream);

namespace {

// FLEX_PLAN_AUTOTUNE: the degree rule picks the column-tile width G from two thresholds measured on a handful of
// shapes; this measures instead.  The neighbouring widths are planned too (same row schedule, computed once),
// each candidate is timed on zero-filled operands of the real size (what a gather costs depends on its address,
// not on the value), and the fastest plan is kept.  Costs two extra plans and 2 * 4*(n*ldb + m*ldc) bytes for the
// duration of the call.
int autotune(flex_plan **pp, const flex_csr *A, int32_t r0, int32_t r1, const int32_t *col_map, const int32_t *dst_map,
             unsigned flags, std::vector<uint32_t> &sched_cache) {
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
        int rq = build_plan(q, A, r0, r1, col_map, dst_map, flags, &sched_cache, g);
        if (rq == FLEX_OK && hipDeviceSynchronize() != hipSuccess) rq = FLEX_ERR_HIP;
        if (rq == FLEX_OK) time_plan(q, &us);
        if (rq == FLEX_OK && !rc && us < best_us) {
            std::swap(best, q);
            best_us = us;
        }
        free_plan_device(q);
        delete q;
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

static int create_common(flex_plan **out, const flex_csr *hostA, int64_t row_begin, int64_t row_end,
                         const int32_t *col_map, const int32_t *dst_map, int k, int device, unsigned flags,
                         int ldb = 0, int ldc = 0) {
    if (!out) return FLEX_ERR_INVALID;
    *out = nullptr;
    if (k <= 0 || device < 0) return FLEX_ERR_INVALID;
    if (ldb == 0) ldb = k;
    if (ldc == 0) ldc = k;
    if (ldb < k || ldc < k) return FLEX_ERR_INVALID;
    const unsigned order = flags & FLEX_ORDER_MASK;
    if (order > FLEX_ORDER_GORDER || (flags & ~(FLEX_ORDER_MASK | FLEX_PLAN_STATS | FLEX_PLAN_AUTOTUNE))) return FLEX_ERR_INVALID;
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
        rc = build_plan(p, hostA, static_cast<int32_t>(row_begin), static_cast<int32_t>(row_end), col_map, dst_map, flags,
                        tune ? &sched_cache : nullptr);
        if (rc == FLEX_OK && tune) rc = autotune(&p, hostA, static_cast<int32_t>(row_begin), static_cast<int32_t>(row_end), col_map, dst_map, flags, sched_cache);
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
    if (!d || !d->A || d->struct_size != sizeof(flex_plan_desc)) return FLEX_ERR_INVALID;
    const bool all_rows = (d->flags & FLEX_PLAN_ROW_RANGE) == 0;  // with the flag, (0,0) is an EMPTY shard, not "everything"
    const int64_t r0 = d->row_begin, r1 = all_rows ? d->A->m : d->row_end;
    if (d->row_map && (!all_rows || d->A->m != d->A->n)) return FLEX_ERR_INVALID;  // a row map renames ALL rows of a graph
    if (!all_rows && (d->flags & FLEX_ORDER_MASK) != FLEX_ORDER_NATURAL) return FLEX_ERR_INVALID;  // reorder first, then shard
    return create_common(out, d->A, all_rows ? 0 : r0, r1, d->col_map, d->row_map, d->k, d->device, d->flags & ~FLEX_PLAN_ROW_RANGE, d->ldb, d->ldc);
}

int flex_spmm(flex_plan *p, const float *dB, float *dC, flex_stream_t stream) {
    if (!p) return FLEX_ERR_INVALID;
    if (p->m == 0) return FLEX_OK;
    if (!dC || (!dB && p->nnz > 0)) return FLEX_ERR_INVALID;
    int cur = -1;
    FLEX_HIP_TRY(hipGetDevice(&cur));
    if (cur != p->device) FLEX_HIP_TRY(hipSetDevice(p->device));
    const bool vec4 = (p->k % 4 == 0) && (p->ldb % 4 == 0) && (p->ldc % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(dB) | reinterpret_cast<uintptr_t>(dC)) % 16 == 0);
    const bool fused = vec4 && p->fused_fixup;  // the generic kernel always leaves the sum to spmm_fixup_kernel
    PlanView v{p->d_rec, p->d_t_beg, p->d_t_dst, p->d_t_aux, p->d_chunk, p->d_partial, p->d_split, p->d_split_cnt,
               fused ? 1u : 0u, p->n_slots, p->k, p->ldb, p->ldc,
               p->xcd_remap ? 1u : 0u, p->lds_extra, p->rec_nt ? 1u : 0u, p->trace};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int rc = launch_spmm(v, p->lanes_per_nz, p->off32, vec4, dB, dC, s, p->unroll);
    if (rc == FLEX_OK && !fused) rc = launch_fixup(p->d_partial, p->d_split, p->n_split, p->k, p->ldc, dC, s);
    if (rc == FLEX_OK && p->n_tiles) {  // the dense tiles' share, added to the rows the kernels above have written
        const TileView tv{p->d_tile_a, p->d_tile_boff, p->d_rt_ptr, p->d_rt_rows, p->n_row_tiles};
        rc = launch_tiles(tv, p->off32, dB, dC, p->k, p->ldb, p->ldc, s);
    }
    if (cur != p->device) (void)hipSetDevice(cur);
    return rc;
}

int flex_plan_measure_imbalance(flex_plan *p, const float *dB, float *dC, flex_stream_t stream, flex_imbalance *out) try {
    if (!p || !out || !dC || (!dB && p->nnz > 0)) return FLEX_ERR_INVALID;
    *out = flex_imbalance{};
    if (p->m == 0 || p->n_slots == 0) return FLEX_OK;
    const bool vec4 = (p->k % 4 == 0) && (p->ldb % 4 == 0) && (p->ldc % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(dB) | reinterpret_cast<uintptr_t>(dC)) % 16 == 0);
    if (!vec4) return FLEX_ERR_UNSUPPORTED;  // the stamped twin exists for the vector kernel only
    int cur = -1;
    FLEX_HIP_TRY(hipGetDevice(&cur));
    if (cur != p->device) FLEX_HIP_TRY(hipSetDevice(p->device));
    const size_t ktiles = (static_cast<size_t>(p->k) + 4 * p->lanes_per_nz - 1) / (4 * p->lanes_per_nz);
    const size_t words = static_cast<size_t>(p->n_slots) * ktiles * 3;
    uint64_t *d_log = nullptr;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int rc = FLEX_OK;
    std::vector<uint64_t> log(words);
    if (hipMalloc(reinterpret_cast<void **>(&d_log), words * 8) != hipSuccess) rc = FLEX_ERR_HIP;
    if (!rc && hipMemsetAsync(d_log, 0, words * 8, s) != hipSuccess) rc = FLEX_ERR_HIP;
    if (!rc) {
        PlanView v{p->d_rec, p->d_t_beg, p->d_t_dst, p->d_t_aux, p->d_chunk, p->d_partial, p->d_split, p->d_split_cnt,
                   p->fused_fixup ? 1u : 0u, p->n_slots, p->k, p->ldb, p->ldc, p->xcd_remap ? 1u : 0u, p->lds_extra, p->rec_nt ? 1u : 0u, d_log};
        rc = launch_spmm_stamped(v, p->lanes_per_nz, p->off32, dB, dC, s);
        if (rc == FLEX_OK && !p->fused_fixup) rc = launch_fixup(p->d_partial, p->d_split, p->n_split, p->k, p->ldc, dC, s);
        if (rc == FLEX_OK && p->n_tiles) {
            const TileView tv{p->d_tile_a, p->d_tile_boff, p->d_rt_ptr, p->d_rt_rows, p->n_row_tiles};
            rc = launch_tiles(tv, p->off32, dB, dC, p->k, p->ldb, p->ldc, s);
        }
    }
    if (!rc && (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(log.data(), d_log, words * 8, hipMemcpyDeviceToHost) != hipSuccess)) rc = FLEX_ERR_HIP;
    (void)hipFree(d_log);
    if (cur != p->device) (void)hipSetDevice(cur);
    if (rc) return rc;
    // reduce: CU = (XCC id, SE/SH/CU bits of HW_ID [15:8]); clock = 100 MHz
    struct Acc {
        uint64_t busy = 0, first = ~0ull, last = 0;
    };
    std::vector<Acc> cu(16 * 256), xcd(16);
    uint64_t t_min = ~0ull, t_max = 0, busy_all = 0, w_max = 0;
    int64_t waves = 0;
    for (size_t i = 0; i < words; i += 3) {
        const uint64_t t0 = log[i], t1 = log[i + 1], id = log[i + 2];
        if (t1 == 0 || t1 < t0) continue;  // a padding entry of the chunk table: the wave left before the stamps
        const uint32_t x = static_cast<uint32_t>(id >> 32) & 15u, c = (static_cast<uint32_t>(id) >> 8) & 255u;
        for (Acc *a : {&cu[x * 256 + c], &xcd[x]}) {
            a->busy += t1 - t0;
            a->first = std::min(a->first, t0);
            a->last = std::max(a->last, t1);
        }
        t_min = std::min(t_min, t0);
        t_max = std::max(t_max, t1);
        busy_all += t1 - t0;
        w_max = std::max(w_max, t1 - t0);
        ++waves;
    }
    if (waves == 0) return FLEX_OK;
    auto summarise = [&](const std::vector<Acc> &v, int32_t *seen, double *busy_imb, double *end_spread) {
        uint64_t bmax = 0, bsum = 0, emin = ~0ull, emax = 0;
        int cnt = 0;
        for (const Acc &a : v) {
            if (a.last == 0) continue;
            ++cnt;
            bmax = std::max(bmax, a.busy);
            bsum += a.busy;
            emin = std::min(emin, a.last);
            emax = std::max(emax, a.last);
        }
        *seen = cnt;
        *busy_imb = bsum ? 100.0 * bmax * cnt / bsum - 100.0 : 0.0;
        *end_spread = t_max > t_min ? 100.0 * (emax - emin) / (t_max - t_min) : 0.0;
    };
    out->waves = waves;
    out->span_us = (t_max - t_min) * 0.01;
    summarise(cu, &out->cus_seen, &out->cu_busy_imb_pct, &out->cu_end_spread_pct);
    summarise(xcd, &out->xcds_seen, &out->xcd_busy_imb_pct, &out->xcd_end_spread_pct);
    out->wave_us_mean = busy_all * 0.01 / waves;
    out->wave_us_max = w_max * 0.01;
    return FLEX_OK;
} catch (const std::bad_alloc &) {
    return FLEX_ERR_NOMEM;
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
    return FLEX_OK;
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

// ≙ the reference's tiler round-trip (mat.cu:905-940: every entry of the pillar format exists exactly once,
// the queues are contiguous): read the plan's DEVICE image back and check that it is a partition --
// chunks tile the tasks, tasks tile the records, every record names a valid B row, every C row is written by
// exactly one task or by exactly one split row whose pieces are contiguous partial slots, padding entries of
// the chunk table are empty.  Independent of the planner's host arrays: it validates what the kernels read.
int flex_plan_self_check(const flex_plan *p) try {
    if (!p) return FLEX_ERR_INVALID;
    if (p->m == 0) return FLEX_OK;
    int cur = -1;
    FLEX_HIP_TRY(hipGetDevice(&cur));
    if (cur != p->device) FLEX_HIP_TRY(hipSetDevice(p->device));
    std::vector<uint2> rec(p->n_records), t_aux(p->n_tasks);
    std::vector<uint32_t> t_beg(static_cast<size_t>(p->n_tasks) + 1), t_dst(p->n_tasks);
    std::vector<uint4> chunk(p->n_slots);
    std::vector<SplitRow> split(p->n_split);
    auto down = [&](void *dst, const void *src, size_t bytes) { return bytes == 0 || hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess; };
    const bool ok_copy = down(rec.data(), p->d_rec, rec.size() * sizeof(uint2)) && down(t_beg.data(), p->d_t_beg, t_beg.size() * 4) &&
                         down(t_dst.data(), p->d_t_dst, t_dst.size() * 4) && down(chunk.data(), p->d_chunk, chunk.size() * sizeof(uint4)) &&
                         down(split.data(), p->d_split, split.size() * sizeof(SplitRow)) && down(t_aux.data(), p->d_t_aux, t_aux.size() * sizeof(uint2));
    if (cur != p->device) (void)hipSetDevice(cur);
    if (!ok_copy) return FLEX_ERR_HIP;

    // tasks tile the record stream
    if (t_beg[0] != 0 || t_beg[p->n_tasks] != p->n_records) return FLEX_ERR_FORMAT;
    for (uint32_t t = 0; t < p->n_tasks; ++t)
        if (t_beg[t] > t_beg[t + 1]) return FLEX_ERR_FORMAT;
    // chunks tile the tasks (in table order, skipping the empty padding entries), each within the kernel's limits
    uint32_t next_task = 0, real = 0;
    std::vector<std::pair<uint32_t, uint32_t>> seen;  // real chunks as (first task, #tasks)
    for (const uint4 &c : chunk) {
        if (c.y == 0) {
            if (c.x | c.z | c.w) return FLEX_ERR_FORMAT;
            continue;
        }
        if (c.y > 63 || c.x + c.y > p->n_tasks || c.z != t_beg[c.x] || c.w != t_beg[c.x + c.y]) return FLEX_ERR_FORMAT;
        seen.emplace_back(c.x, c.y);
        ++real;
    }
    if (real != p->n_chunks) return FLEX_ERR_FORMAT;
    std::sort(seen.begin(), seen.end());
    for (const auto &c : seen) {
        if (c.first != next_task) return FLEX_ERR_FORMAT;
        next_task += c.second;
    }
    if (next_task != p->n_tasks) return FLEX_ERR_FORMAT;
    // records name valid B rows
    const uint64_t row_bytes = static_cast<uint64_t>(p->ldb) * 4u;
    for (const uint2 &r : rec) {
        const uint64_t col = p->off32 ? r.x / row_bytes : r.x;
        if (col >= static_cast<uint64_t>(p->n) || (p->off32 && r.x % row_bytes != 0)) return FLEX_ERR_FORMAT;
    }
    // every C row exactly once: by one task, or by one split row whose pieces own consecutive partial slots; every
    // partial slot is written by exactly one task, and that task names its row and the row's piece count (t_aux:
    // what the arrival counter is compared with).  Pieces of a row need NOT be consecutive tasks (2-D schedules).
    std::vector<uint8_t> written(static_cast<size_t>(p->c_rows), 0), slot_taken(p->n_partials, 0);
    for (uint32_t t = 0; t < p->n_tasks; ++t) {
        const uint32_t d = t_dst[t];
        if (d & kPartialFlag) {
            const uint32_t ps = d & ~kPartialFlag;
            if (ps >= p->n_partials || slot_taken[ps]++) return FLEX_ERR_FORMAT;
            const uint2 a = t_aux[t];
            if (a.x >= p->n_split || a.y != split[a.x].count || ps < split[a.x].first || ps >= split[a.x].first + split[a.x].count) return FLEX_ERR_FORMAT;
        } else {
            if (d >= p->c_rows || written[d]++) return FLEX_ERR_FORMAT;
        }
    }
    for (uint8_t w : slot_taken)
        if (w != 1) return FLEX_ERR_FORMAT;
    uint32_t first = 0;
    for (uint32_t i = 0; i < p->n_split; ++i) {
        const SplitRow &sr = split[i];
        if (sr.first != first || sr.count < 2 || sr.row >= p->c_rows || written[sr.row]++) return FLEX_ERR_FORMAT;
        first += sr.count;
    }
    if (first != p->n_partials) return FLEX_ERR_FORMAT;
    // dense tiles: the row-tile directory tiles the tile list, every listed C row is valid and named by one row tile only,
    // every tile column names a valid B row
    if (p->n_tiles) {
        std::vector<uint32_t> rt_ptr(static_cast<size_t>(p->n_row_tiles) + 1), rt_rows(static_cast<size_t>(p->n_row_tiles) * 32),
            boff(static_cast<size_t>(p->n_tiles) * 32);
        if (cur != p->device) FLEX_HIP_TRY(hipSetDevice(p->device));
        const bool ok_t = down(rt_ptr.data(), p->d_rt_ptr, rt_ptr.size() * 4) && down(rt_rows.data(), p->d_rt_rows, rt_rows.size() * 4) &&
                          down(boff.data(), p->d_tile_boff, boff.size() * 4);
        if (cur != p->device) (void)hipSetDevice(cur);
        if (!ok_t) return FLEX_ERR_HIP;
        if (rt_ptr[0] != 0 || rt_ptr[p->n_row_tiles] != p->n_tiles) return FLEX_ERR_FORMAT;
        for (uint32_t i = 0; i < p->n_row_tiles; ++i)
            if (rt_ptr[i] >= rt_ptr[i + 1]) return FLEX_ERR_FORMAT;
        std::vector<uint8_t> in_rt(static_cast<size_t>(p->c_rows), 0);
        for (uint32_t d : rt_rows) {
            if (d == 0xFFFFFFFFu) continue;
            if (d >= p->c_rows || in_rt[d]++) return FLEX_ERR_FORMAT;
        }
        for (uint32_t o : boff) {
            const uint64_t col = p->off32 ? o / row_bytes : o;
            if (col >= static_cast<uint64_t>(p->n) || (p->off32 && o % row_bytes != 0)) return FLEX_ERR_FORMAT;
        }
    }
    // a full plan (not a row shard of a mapped matrix) writes every row of C
    if (p->c_rows == p->m)
        for (uint8_t w : written)
            if (w != 1) return FLEX_ERR_FORMAT;
    return FLEX_OK;
} catch (const std::bad_alloc &) {
    return FLEX_ERR_NOMEM;
} catch (...) {
    return FLEX_ERR_INVALID;
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
