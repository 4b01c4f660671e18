// plan_build.cpp -- the planner: rows [r0,r1) of a host CSR -> the device image the kernels read.
//
// PlanBuilder::run() is the list of stages; each stage is one member function that reads what the stages before it
// left in the builder and leaves its own result there:
//
//   choose_tile_width      G lanes per record (column tile of 4G columns), 32- or 64-bit B addressing
//   order_rows             sched[i] = row processed i-th (natural / RCM / community / Gorder), colpos = its inverse
//   route_dense_tiles      detector report; 32x32 tiles dense enough for the MFMA kernel leave the record stream
//   read_knobs             chunk budget, piece length, 2-D panel size ... (rules measured on MI355X, env overrides)
//   cut_rows_into_pieces   a row is one piece, or several: by length (hubs) and, in 2-D, by column panel
//   slice_rows_for_xcds    2-D: the eight row slices (1-D cuts the chunk table by cost instead, below)
//   order_pieces           emission order: schedule order, or per slice phase by phase
//   number_split_rows      rows summed from several pieces: consecutive partial slots + the arrival bookkeeping
//   pack_tasks_into_chunks one task per piece, tasks packed into per-wave chunks of about one budget
//   fill_records           {B-row offset, value} per nonzero, padded to whole steps (parallel over tasks)
//   upload_tasks           records / tasks / dense tiles to the device
//   build_chunk_table      chunk headers in launch order, cut into eight cost-balanced XCD slices; statistics; the
//                          split-row workspace
//
// The result does not depend on the number of host threads (tests/test_planner_host.py compares the images).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <new>
#include <numeric>

#include "host_parallel.h"
#include "plan.h"

namespace flex {

namespace {

// A run of one row's records that one task processes.
struct Piece {
    uint32_t spos;       // position of the piece's row in the schedule
    uint32_t beg, end;   // its records [beg,end) in rcol/rval
    uint32_t phase;      // 0 in 1-D; column panel + 1, or kFarPhase, in 2-D
    uint32_t own_chunk;  // a slice of a run longer than one budget: a chunk of its own
};
constexpr uint32_t kFarPhase = 0xFFFFFFFEu;
// What a task is made of: one piece, or a BUNDLE -- up to S short whole rows side by side, slot s of every step working on row s,
// so that nothing is reduced across slots and a finished bundle is one 16-byte store per lane (spmm_kernels.hip, flush).
struct TaskSpec {
    uint32_t ref;    // a piece: its position in `emit`; a bundle: its first entry in `bundle_piece`
    uint32_t rows;   // 0 = a piece; else the rows of the bundle (<= S)
    uint32_t steps;  // bundle: the length of its longest row
};
constexpr uint32_t kMaxTasksPerWave = 63;  // the kernel hands descriptors out by lane (compute_chunk)

class PlanBuilder {
   public:
    PlanBuilder(flex_plan *plan, const flex_csr *csr, int32_t row_begin, int32_t row_end, const int32_t *col_map_, const int32_t *dst_map_,
                unsigned flags_, const flex_plan_tuning &tuning, std::vector<uint32_t> *sched_cache_, int force_G_, bool want_blocks_ = false)
        : p(plan), A(csr), r0(row_begin), r1(row_end), m(row_end - row_begin), k(plan->k), col_map(col_map_), dst_map(dst_map_),
          flags(flags_), order(flags_ & FLEX_ORDER_MASK), force_G(force_G_), want_blocks(want_blocks_), tn(tuning), sched(sched_cache_ ? *sched_cache_ : sched_local),
          have_cache(sched_cache_ != nullptr), timing(plan_timing_enabled()), t_last(std::chrono::steady_clock::now()) {}

    // The first two stages alone, for the row-block route (build_plan below): the schedule and its inverse.
    int schedule_only(std::vector<uint32_t> **sched_out, std::vector<uint32_t> **colpos_out) {
        if (order > FLEX_ORDER_GORDER) return FLEX_ERR_INVALID;
        if (order != FLEX_ORDER_NATURAL && (A->m != A->n || r0 != 0 || r1 != A->m)) return FLEX_ERR_INVALID;
        choose_tile_width();
        const int rc = order_rows();
        lap("row schedule");
        *sched_out = &sched;
        *colpos_out = &colpos;
        return rc;
    }

    int run() {
        if (order > FLEX_ORDER_GORDER) return FLEX_ERR_INVALID;
        // graph orderings need the whole square matrix
        if (order != FLEX_ORDER_NATURAL && (A->m != A->n || r0 != 0 || r1 != A->m)) return FLEX_ERR_INVALID;
        p->order = order;
        int rc;
        choose_tile_width();
        if ((rc = order_rows())) return rc;
        lap("row schedule");
        if ((flags & FLEX_PLAN_STATS) && m > 0) {  // the reuse a workgroup could have above the L2 (flex_plan_stats.lds_*)
            p->lds_hot[0] = 100.0 * estimate_hot_share(A, sched, 480, 2, 4, &p->lds_u[0]);
            p->lds_hot[1] = 100.0 * estimate_hot_share(A, sched, 480, 4, 4, &p->lds_u[1]);
        }
        if ((rc = route_dense_tiles())) return rc;
        lap("dense-tile detector");
        if (want_blocks && tiles.nnz == 0 && m > 0) {  // the two routes do not combine (yet): dense tiles first
            if ((rc = route_hot_blocks())) return rc;
            lap("hot blocks");
        }
        read_knobs();
        if ((rc = cut_rows_into_pieces())) return rc;
        lap("pieces");
        slice_rows_for_xcds();
        order_pieces();
        lap("emission order");
        number_split_rows();
        form_tasks();
        if ((rc = pack_tasks_into_chunks())) return rc;
        fill_records();
        lap("records and tasks");
        if ((rc = upload_tasks())) return rc;
        lap("upload records/tasks");
        if ((rc = build_chunk_table())) return rc;
        lap("chunk table, stats");
        return FLEX_OK;
    }

   private:
    // ---- inputs
    flex_plan *const p;
    const flex_csr *A;  // after route_dense_tiles: A minus the entries that moved into tiles
    const int32_t r0, r1, m;
    const int k;
    const int32_t *const col_map;  // B row read by column c (NULL = c)
    const int32_t *const dst_map;  // C row written by row r (NULL = r - r0)
    const unsigned flags, order;
    const int force_G;
    const bool want_blocks;  // split the matrix: nonzeros with reuse inside a block of rows go to the hot-block image (block_plan.cpp)
    const flex_plan_tuning &tn;  // the caller's knobs: 0 = the rule of the stage that reads it; p->tuning receives what was used
    std::vector<uint32_t> sched_local;
    std::vector<uint32_t> &sched;  // sched[i] = row of A processed i-th
    const bool have_cache;
    const bool timing;  // FLEX_PLAN_TIMING (the one environment variable left: it changes nothing but stderr): phase times
    std::chrono::steady_clock::time_point t_last;

    // ---- stage results
    int G = 8;                     // lanes per record
    uint32_t S = 8;                // records per step = 64 / G: tasks are padded to it
    double avg_deg = 0.0;          // of the rows as given (before tiles leave)
    std::vector<uint32_t> colpos;  // position of a column's vertex in the schedule; empty = the column id itself
    DenseTiles tiles;
    flex_csr A_f{};                // the filtered copy of A (same rows, same ids)
    std::vector<uint32_t> f_rowptr, f_col;
    std::vector<float> f_vals;
    // knobs
    uint32_t wave_nnz = 0, row_cost = 16, long_row = 0, piece_len = 0, seg_min = 4, pshift = 0, far_window = 0;
    bool two_d = false;
    bool xcd_dealt = false;  // tuning.xcd_slices = 3: stretches of the schedule dealt to the XCDs in turn (build_chunk_table)
    // pieces
    std::vector<Piece> pieces;
    std::vector<uint32_t> row_first_piece;  // [m+1] pieces of schedule position i
    std::vector<uint32_t> pcol;             // 2-D: the records of every row re-grouped by piece (index e - e_base)
    std::vector<float> pval;
    const uint32_t *rcol = nullptr;  // where piece ranges point: A's arrays (1-D) or pcol/pval (2-D)
    const float *rval = nullptr;
    uint32_t slice_row[kXcds + 1] = {0};  // 2-D: schedule positions of the XCD slices
    std::vector<uint32_t> emit;           // pieces in emission order
    // tasks
    bool bundles_on = false;
    uint32_t bundle_len = 0;              // rows of at most this many records are candidates for a bundle
    std::vector<TaskSpec> tasks;
    std::vector<uint32_t> bundle_piece;   // positions in `emit` of the rows of each bundle, longest first
    std::vector<uint32_t> bd_rows;        // PlanView::bd_rows
    std::vector<uint2> chunk_bd;          // per chunk {first entry in bd_rows, entries}
    uint32_t n_bundles = 0;
    int64_t bundle_rows = 0;
    // split rows
    std::vector<SplitRow> split;
    std::vector<uint32_t> row_sidx, row_first_partial;
    uint32_t n_partials = 0;
    int64_t split_nnz = 0;
    // tasks, chunks, records
    std::vector<uint32_t> t_beg, t_dst, w_task;  // w_task[c] = first task of chunk c (+ sentinel)
    std::vector<uint2> t_aux;
    RecordVec rec;  // filled in parallel right after it is sized: no zero-fill pass (plan.h)
    uint32_t slice_chunk[kXcds + 1] = {0};  // 2-D: first chunk of each XCD slice

    void lap(const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "plan: %-22s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    }
    // a knob: the caller's value if positive, else the rule's
    static long pick(int32_t given, long rule) { return given > 0 ? given : rule; }
    uint32_t dst_of(uint32_t r) const { return dst_map ? static_cast<uint32_t>(dst_map[r]) : r - static_cast<uint32_t>(r0); }
    int64_t slice_nnz() const { return static_cast<int64_t>(A->rowPtr[r1]) - A->rowPtr[r0]; }

    // G lanes x float4 cover one column tile of 4*G columns; k wider than that runs as several tiles
    // (blockIdx.y, dispatched one after the other).  Widest tile (fewest instructions per byte) for
    // low-degree graphs; high-degree graphs are bound by L2-miss traffic instead -- the B rows touched by
    // the resident waves (waves x records x 16*G bytes) overflow the 4 MiB L2s -- and a narrower tile
    // shrinks that footprint at the price of re-reading the records once per tile.  Measured on MI355X,
    // k=128: reddit-like generator, G=16 vs 32: -7 % at degree 12, +7 % at 24, +15 % at
    // 36..100; amazon shape +12 % (G=16), +16 % (G=8); flickr (degree 11) -7 %, yelp (19.5) -2.5 %.
    void choose_tile_width() {
        avg_deg = m > 0 ? static_cast<double>(A->rowPtr[r1] - A->rowPtr[r0]) / m : 0.0;
        G = 8;
        while (4 * G < k && G < 64) G <<= 1;
        // k <= 16: a B row is at most 64 bytes, and four lanes cover it -- 16 records per step instead of 8 with half of the lanes
        // gathering a column nobody stores (≙ the reference's narrow kernel, flex.cu:81-118: one thread per (row, column), 128 / k rows
        // per block).  Rows are padded to whole steps, so the narrow tile is the rule only where rows are long enough to fill them.
        // (With row bundles -- form_tasks -- short rows sit side by side instead of being padded, and the narrow tile wins at every degree:
        // epinions stand-in, degree 6.4, k=16: 12.2 us against 13.4 on the 8-lane tile, flickr shape 11.6 against 12.6.)
        const bool narrow_ok = k <= 16 && k % 4 == 0;
        if (force_G) {
            G = std::min(G, force_G);
        } else if (const int g_t = tn.lanes_per_nz; g_t == 8 || g_t == 16 || g_t == 32 || g_t == 64 || (g_t == 4 && narrow_ok)) {
            G = g_t == 4 ? 4 : std::min<int>(G, g_t);  // tuning experiments
        } else if (narrow_ok && (avg_deg >= 8.0 || tn.bundle == 1 || (tn.bundle != 2 && bundle_rule()))) {
            G = 4;
        } else {
            G = std::min(G, 32);  // k = 256 as two 128-column tiles beats one 256-column tile on every shape measured
            if (avg_deg >= 24.0) G = std::min(G, 16);
            if (avg_deg >= 128.0) G = std::min(G, 8);
            // With row bundles the 16-lane tile is the better one at LOW degree too: its short rows cost no more instructions than on
            // the wide tile, and two passes over half-rows of B keep more of them in the L2s (k=128, 32-lane tile without bundles ->
            // 16-lane tile with: epinions stand-in 47.9 -> 46.1 us on 316 -> 300 MB, yelp shape 533 -> 519 us on 3.59 -> 3.22 GB,
            // flickr shape 38.2 -> 38.1; profiles/r04_row_bundles.txt).  The 8-lane tile loses again (48.3 / 560 / 38.6).
            if (tn.bundle == 1 || (tn.bundle != 2 && bundle_rule())) G = std::min(G, 16);
        }
        S = 64u / static_cast<uint32_t>(G);
        p->lanes_per_nz = G;
        p->off32 = static_cast<uint64_t>(A->n) * static_cast<uint64_t>(p->ldb) * 4u <= (uint64_t(1) << 32);
    }

    int order_rows() {
        if (have_cache && sched.size() == static_cast<size_t>(m) && m > 0) {
            // computed by an earlier candidate of the same matrix
        } else if (sched.assign(static_cast<size_t>(m), 0u); order != FLEX_ORDER_NATURAL) {
            std::vector<uint32_t> rank;
            const int rc = order == FLEX_ORDER_RCM       ? order_rcm_host(m, A->rowPtr, A->col, rank)
                           : order == FLEX_ORDER_CLUSTER ? order_cluster_host(m, A->rowPtr, A->col, rank, &tn.cluster)
                                                         : order_gorder_host(m, A->rowPtr, A->col, 3, rank);
            if (rc) {
                sched.clear();
                return rc;
            }
            for (int32_t r = 0; r < m; ++r) sched[rank[r]] = static_cast<uint32_t>(r);
        } else {
            std::iota(sched.begin(), sched.end(), static_cast<uint32_t>(r0));
        }
        // position of a column's vertex in the schedule (what "near" means for a reordered square matrix); for a
        // natural-order plan, a mapped plan or a row shard the column ids of A are positions already
        if (order != FLEX_ORDER_NATURAL) {
            colpos.resize(static_cast<size_t>(m));
            for (int32_t i = 0; i < m; ++i) colpos[sched[i]] = static_cast<uint32_t>(i);
        }
        return FLEX_OK;
    }

    // Dense tiles -> MFMA kernel (tuning.mfma: 1 = route tiles of fill >= mfma_fill_pct %, 2 = never; default:
    // route when a sampled look at every 64th row tile finds at least 10 % of the nonzeros in such tiles -- most
    // graphs have none and then pay 1/64 of one pass).  With FLEX_PLAN_STATS the detector looks at every tile, so
    // that its report (share of nonzeros in tiles of fill >= 0.10 / 0.25 / 0.50) is exact.
    // Default threshold 60 %: measured on MI355X at k = 128 (tools/probe_mfma.py, 64-row diagonal blocks + 8 random
    // entries per row, 200 K rows): routing blocks of fill 0.9 takes 297 -> 220 us, fill 0.6 228 -> 219 (break-even),
    // fill 0.3 160 -> 217 (slower: a tile costs the same whatever its fill, and its 32 C rows are read and written
    // once more), DESIGN.md 3.5.
    int route_dense_tiles() {
        const long mode_mfma = tn.mfma;
        const uint32_t fill_pct = static_cast<uint32_t>(std::clamp<long>(pick(tn.mfma_fill_pct, 60), 1, 100));
        p->tuning.mfma = static_cast<int32_t>(mode_mfma);
        p->tuning.mfma_fill_pct = static_cast<int32_t>(fill_pct);
        const uint32_t thr = (1024u * fill_pct + 99u) / 100u;
        const bool report = (flags & FLEX_PLAN_STATS) != 0;
        const int64_t nnz_in = slice_nnz();
        const uint32_t row_bytes32 = static_cast<uint32_t>(p->ldb) * 4u;
        bool route = mode_mfma == 1;
        std::vector<uint8_t> in_tile;
        int rc = FLEX_OK;
        if (mode_mfma != 1 && mode_mfma != 2 && m >= 2048 && nnz_in >= (1 << 16)) {  // the sampled look
            DenseTiles probe;
            if ((rc = detect_dense_tiles(A, r0, m, sched, colpos, col_map, dst_map, p->off32, row_bytes32, 0, 64, in_tile, probe))) return rc;
            const int64_t share = fill_pct <= 10 ? probe.hist_nnz[0] : fill_pct <= 25 ? probe.hist_nnz[1] : probe.hist_nnz[2];  // >= 0.5 also screens for 0.6
            route = share * 64 * 10 >= nnz_in;  // >= 10 % of the nonzeros, extrapolated from the sample (amazon shape: 2.3 % in
                                                //    such tiles; routing them changed nothing, 9.08 vs 9.08 ms, and cost 1.1 s of planning)
        }
        if (route || report) {
            if (route) in_tile.assign(static_cast<size_t>(nnz_in), 0);
            if ((rc = detect_dense_tiles(A, r0, m, sched, colpos, col_map, dst_map, p->off32, row_bytes32, route ? thr : 0, 1, in_tile, tiles))) return rc;
            p->tile_hist[0] = tiles.hist_nnz[0];
            p->tile_hist[1] = tiles.hist_nnz[1];
            p->tile_hist[2] = tiles.hist_nnz[2];
            p->tile_cells = tiles.n_cells;
            p->tile_hist_valid = true;
        }
        if (tiles.nnz == 0) return FLEX_OK;
        // the vector kernel gets A minus the entries that moved into tiles (same rows, same ids)
        keep_unmarked(in_tile);
        return FLEX_OK;
    }

    // A := A minus the entries of rows [r0,r1) whose mask byte is set (index e - rowPtr[r0]); same rows, same ids, order kept.
    void keep_unmarked(const std::vector<uint8_t> &mask) {
        f_rowptr.assign(static_cast<size_t>(A->m) + 1, 0u);
        const uint32_t eb = A->rowPtr[r0];
        // per-row counts, then the prefix, then a parallel copy (the Amazon shape filters 264 M entries)
        parallel_chunks((static_cast<int64_t>(m) + 4095) / 4096, [&](int64_t b) {
            for (int64_t r = r0 + b * 4096; r < std::min<int64_t>(r1, r0 + (b + 1) * 4096); ++r) {
                uint32_t c = 0;
                for (uint32_t e = A->rowPtr[r]; e < A->rowPtr[r + 1]; ++e) c += mask[e - eb] ? 0u : 1u;
                f_rowptr[r + 1] = c;
            }
        });
        for (int32_t r = 0; r < A->m; ++r) f_rowptr[r + 1] += f_rowptr[r];
        f_col.resize(f_rowptr[A->m]);
        f_vals.resize(f_rowptr[A->m]);
        const flex_csr *src = A;
        parallel_chunks((static_cast<int64_t>(m) + 4095) / 4096, [&](int64_t b) {
            for (int64_t r = r0 + b * 4096; r < std::min<int64_t>(r1, r0 + (b + 1) * 4096); ++r) {
                uint32_t o = f_rowptr[r];
                for (uint32_t e = src->rowPtr[r]; e < src->rowPtr[r + 1]; ++e)
                    if (!mask[e - eb]) {
                        f_col[o] = src->col[e];
                        f_vals[o] = src->vals[e];
                        ++o;
                    }
            }
        });
        A_f = flex_csr{A->m, A->n, static_cast<int64_t>(f_rowptr[A->m]), f_rowptr.data(), f_col.data(), f_vals.data()};
        A = &A_f;
    }

    // The hot-block route (a SPLIT of the matrix).  Nonzeros whose column is used at least `thr` times
    // inside their block of schedule-consecutive rows have reuse on chip: they go to the block image (block_plan.cpp; one
    // workgroup per block stages those B rows in LDS).  Every other nonzero is an L2 miss under any schedule, and moving L2
    // misses at the fabric's rate is what the flat kernel does best: it gets A minus the hot entries (same rows, same ids, same
    // schedule), runs first and writes every C row; the hot kernel adds its part afterwards (flex_spmm).
    int route_hot_blocks() {
        BlockKnobs kn;
        const int32_t rd = tn.block_rounds;
        if (rd == 2 || rd == 4 || rd == 8) kn.rounds = static_cast<uint32_t>(rd);
        else if (rd != 0) return FLEX_ERR_INVALID;
        else if (m < 8 * 256 * 480) kn.rounds = m < 8 * 256 * 240 ? 2 : 4;  // smaller blocks while there are fewer than ~8 per CU
        if (tn.block_panel_rows) {
            if (tn.block_panel_rows % 4 != 0 || tn.block_panel_rows > static_cast<int32_t>(kBkPanelMax)) return FLEX_ERR_INVALID;
            kn.panel_rows = static_cast<uint32_t>(tn.block_panel_rows);
        }
        kn.thr = tn.block_thr ? static_cast<uint32_t>(tn.block_thr) : 3u;  // measured: 3 is 1 % ahead of 2 where the route pays (fewer panels, less staging)
        // nonzeros per slot: a longer row is spread over ceil(len / cap) slots, so that its runs are about as long as its neighbours'
        // (a power-law graph keeps a third of its nonzeros in rows several times the average: they are the hottest ones)
        kn.cap = tn.block_cap ? static_cast<uint32_t>(std::min<int32_t>(tn.block_cap, 60000)) : static_cast<uint32_t>(std::clamp(1.5 * avg_deg, 32.0, 4096.0));
        kn.min_last_panel = std::min<uint32_t>(32, kn.panel_rows / 4);
        BlockImage img;
        std::vector<uint8_t> hot_mask;
        const uint32_t row_bytes32 = static_cast<uint32_t>(p->ldb) * 4u;
        int rc;
        if ((rc = build_blocks(A, sched, colpos, col_map, dst_map, r0, row_bytes32, kn, img, hot_mask))) return rc;
        flex_plan_tuning &u = p->tuning;
        u.blocks = 1;
        u.block_rounds = static_cast<int32_t>(kn.rounds);
        u.block_panel_rows = static_cast<int32_t>(kn.panel_rows);
        u.block_thr = static_cast<int32_t>(kn.thr);
        u.block_cap = static_cast<int32_t>(kn.cap);
        if (timing)
            std::fprintf(stderr, "plan: hot blocks %u x %u slots, %lld rows, %lld nnz; candidates %.1f %%, hot %.1f %% (left cold: panel budget %.1f %%, short last panel %.1f %%, runs beyond %u steps %.1f %%), "
                         "u %.2f, records / hot %.2f, panels / block %.1f\n", img.n_blocks, kn.rounds * kBkRowsPerRound, static_cast<long long>(img.rows), static_cast<long long>(img.nnz),
                         100.0 * img.cand_nnz / std::max<int64_t>(img.nnz, 1), 100.0 * img.hot_nnz / std::max<int64_t>(img.nnz, 1), 100.0 * img.lost_panels / std::max<int64_t>(img.nnz, 1),
                         100.0 * img.lost_last / std::max<int64_t>(img.nnz, 1), kn.run_max, 100.0 * img.lost_run / std::max<int64_t>(img.nnz, 1),
                         static_cast<double>(img.hot_nnz) / std::max<int64_t>(img.hot_cols, 1), static_cast<double>(img.rec.size()) / std::max<int64_t>(img.hot_nnz, 1),
                         static_cast<double>(img.panels) / std::max<uint32_t>(img.n_blocks, 1));
        if (img.hot_nnz == 0) return FLEX_OK;  // nothing has reuse: a flat plan
        if ((rc = upload(&p->d_bk_hdr, img.hdr, &p->device_bytes))) return rc;
        if ((rc = upload(&p->d_bk_wstart, img.wstart, &p->device_bytes))) return rc;
        if ((rc = upload(&p->d_bk_cnt, img.cnt, &p->device_bytes))) return rc;
        if ((rc = upload(&p->d_bk_hcol, img.hcol, &p->device_bytes))) return rc;
        if ((rc = upload(&p->d_bk_brow, img.brow, &p->device_bytes))) return rc;
        if ((rc = upload(&p->d_bk_link, img.link, &p->device_bytes))) return rc;
        if ((rc = upload(&p->d_bk_rec, img.rec, &p->device_bytes))) return rc;
        p->bk_blocks = img.n_blocks;
        p->bk_rounds = img.rounds;
        p->bk_panel_rows = img.panel_rows;
        p->bk_rows = img.rows;
        p->bk_nnz = img.nnz;
        p->bk_hot_nnz = img.hot_nnz;
        p->bk_hot_cols = img.hot_cols;
        p->bk_panels = img.panels;
        p->bk_records = static_cast<int64_t>(img.rec.size());
        p->bk_ablate = static_cast<uint32_t>(tn.block_ablate);
        img = BlockImage{};
        keep_unmarked(hot_mask);
        // what stays flat is a different matrix -- by construction the nonzeros WITHOUT reuse nearby: its tile width and chunk budget
        // follow ITS degree (the Amazon shape's 40 % that stay cold: degree 69, so 64-column tiles and two passes over the records
        // instead of 32-column tiles and four)
        choose_tile_width();
        return FLEX_OK;
    }

    void read_knobs() {
        // chunk budget in records: short chunks keep the dispatcher's load balancing fine-grained on low-degree
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
        // (round 3, pubmed.csv k=32 with no row split: 4.8 us at 48-64 records per chunk, 4.9 at 96, 5.1 at 128, 5.6 at 192 -- on the G = 8
        //  tile a graph this small may go down to 64; k=128 is flat from 64 to 128: profiles/r03_small_graph_sweep.txt)
        // (round 4, the 16-lane tile with row bundles -- what small graphs of short rows run at k >= 64: 64 records per chunk against
        //  96: pubmed.csv k=64 6.4 -> 5.6 us, k=128 7.9 -> 7.1; wiki-Vote shape k=64 6.7 -> 6.3, k=128 level;
        //  profiles/r04_row_bundles.txt)
        const bool bundles_expected = tn.two_d != 1 && S >= kBundleMinSlots && m > 0 && (tn.bundle == 1 || (tn.bundle != 2 && bundle_rule()));
        const long n_rec = static_cast<long>(A->rowPtr[r1] - A->rowPtr[r0]);
        auto_budget = std::min(auto_budget, std::max<long>(G <= 8 || bundles_expected ? 64 : lo_budget, n_rec / 2048));
        wave_nnz = static_cast<uint32_t>(pick(tn.chunk_records, auto_budget));
        row_cost = static_cast<uint32_t>(pick(tn.row_cost, 16));
        // Contiguous XCD slices (workgroup ids remapped) keep a community's rows on ONE private L2; they pay when the
        // eighths of the schedule run at different speeds, because a slice cannot borrow an idle XCD.  Community and
        // natural order: remap on (reddit k=128 cluster 676 vs 811 us without, flickr 38 vs 49, yelp 523 vs 688; natural
        // order the same either way).  RCM / Gorder (BFS-like orders: the eighths differ in L2 hit rate, so cost-balanced
        // slices end 20-34 % apart): the hardware's round-robin over XCDs is faster -- reddit RCM 1160 -> 1102 us (k=32:
        // 260 -> 251), yelp RCM 973 -> 911, flickr RCM 63.8 -> 60.9, flickr Gorder 59.1 -> 52.7.  tuning.xcd_slices = 1 / 2 forces.
        const long remap_env = tn.xcd_slices;
        // A reordered loader planned as given says so itself: FLEX_PLAN_XCD_INTERLEAVE.
        const bool interleave = order == FLEX_ORDER_RCM || order == FLEX_ORDER_GORDER || (flags & FLEX_PLAN_XCD_INTERLEAVE) != 0;
        p->xcd_remap = remap_env == 1 || remap_env == 3 || (remap_env != 2 && !interleave);
        xcd_dealt = remap_env == 3;
        p->lds_extra = static_cast<unsigned>(std::max(0, tn.lds_extra)) & ~15u;
        // The record stream is read once per column tile.  Non-temporal loads keep it from displacing B rows in the L2s and
        // the Infinity Cache, but they also come back slower and sit on the header -> records -> gathers chain of every chunk.
        // Measured on MI355X (same box each):
        //   several tiles (k > 4G), stream >= 32 MB:  amazon shape k=128 9.48 -> 9.00 ms, reddit 687 -> 679 us, yelp 524 -> 515 us
        //   one tile, stream of 0.1-0.2 GB:           reddit k=32 178 -> 196 us, yelp k=32 130 -> 148 us   (worse)
        //   one tile, stream of 2.1 GB (8x the Infinity Cache): amazon k=32 2.43 -> 2.29 ms
        //   small streams:                            flickr k=128 37.9 -> 40.4 us                          (worse)
        // hence: on for multi-tile launches from 32 MB, for single-tile launches only from 1 GiB.  tuning.rec_nt = 1 / 2 forces.
        const long nt_env = tn.rec_nt;
        const int ktiles = (k + 4 * G - 1) / (4 * G);
        const uint64_t stream_bytes = static_cast<uint64_t>(A->rowPtr[r1] - A->rowPtr[r0]) * 8u;
        // grouped tile order (tuning.tile_group; the rule is "off" until measured): the records of a group are meant to come back from
        // the Infinity Cache on the group's later tiles, so they are read with ordinary loads there
        p->tile_group = (ktiles >= 2 && tn.tile_group > 1) ? static_cast<uint32_t>(tn.tile_group) : 0u;
        p->rec_nt = nt_env == 1 || (nt_env != 2 && p->tile_group == 0 && stream_bytes >= (ktiles >= 2 ? (32ull << 20) : (1ull << 30)));
        p->unroll = tn.unroll == 8 ? 8 : 0;
        // Rows longer than one budget are cut into pieces of one budget, each a chunk of its own that
        // writes a k-wide partial sum: one wave keeps only U gathers in flight, so a long row is much
        // faster as several concurrent pieces (flickr, MI355X: 88 us with no splitting, 43 us with
        // rows > 192 records split, 40 us with rows > 96 split; reddit is flat from 256 to 512;
        // DESIGN.md 3.3).  Pieces stay in schedule order: moving them to the
        // front of the XCD slices helped flickr by 3 % and cost reddit 12 % (half its chunks are pieces).
        // A graph that does not fill the chip (fewer chunks than half the resident wave slots) gains nothing from cutting a row that is
        // only a few gather blocks long -- the launch ends with its slowest wave either way -- and every split row costs the second
        // launch (spmm_fixup_kernel): such rows stay whole up to 12 blocks of U gathers.  pubmed.csv (6 rows beyond one budget):
        // k=32 7.1 -> 4.8 us, k=128 9.2 -> 7.1 us (profiles/r03_small_graph_sweep.txt; 5.4 / 8.2 with the in-launch sum of round 2).
        long whole_row = wave_nnz;
        if (n_rec / std::max<long>(wave_nnz, 1) <= 4096) whole_row = std::max<long>(wave_nnz, std::min<long>(4L * wave_nnz, 12L * S * (G >= 32 ? 8 : 4)));
        long_row = static_cast<uint32_t>(pick(tn.long_row, whole_row));
        piece_len = std::max<uint32_t>(S, static_cast<uint32_t>(pick(tn.piece_records, wave_nnz)) / S * S);

        // Column panels (the 2-D schedule; ≙ the column spans of csr2_DiagTiling's rounds 2-3, mat.cu:680-942, and
        // csr2seg_Cmajor, mat.cu:1192-1269, re-thought for eight private 4 MiB L2s).  An XCD walks ONE contiguous slice of
        // the rows; in 1-D that slice is walked row by row and the B rows it needs within +-w communities (megabytes)
        // are evicted between uses.  In 2-D the slice is walked PHASE by PHASE: phase q holds, for every row of the
        // slice, the records whose column lies in panel q of B (P rows = `panel_bytes` of one column tile, about half an
        // L2), so whatever the resident waves gather at one time comes from one or two panels and hits the L2 by
        // construction; the price is that a row with records in several phases is summed from several pieces (a k-wide
        // partial sum written and read once per piece).  Only runs of >= `seg_min` records of a row in one panel become a
        // piece; the rest of the row (its scattered columns, which miss either way) is ONE more piece in a last phase.
        // The rule is "off": decided by measurement; tuning.two_d = 1 forces it on (tests, tuning), any size.
        two_d = tn.two_d == 1 && m > 0;
        const uint64_t tile_bytes = 16ull * static_cast<uint64_t>(G);  // one B row of one column tile
        const uint64_t panel_bytes = static_cast<uint64_t>(pick(tn.panel_kb, 2048)) << 10;
        pshift = 0;
        while ((2ull << pshift) * tile_bytes <= panel_bytes) ++pshift;  // P = 2^pshift rows of B per panel
        seg_min = static_cast<uint32_t>(pick(tn.seg_min, 4));
        // Row bundles (form_tasks): only on the tiles of 4 or more slots per step (the kernels of the wide tiles have no code for them),
        // and not on 2-D plans (their tasks are runs of a row, not rows)
        bundles_on = bundles_expected && !two_d;
        bundle_len = static_cast<uint32_t>(pick(tn.bundle_len, S >= 8 ? 12 : 16));
        far_window = two_d ? 0u : static_cast<uint32_t>(std::max(0, tn.far_first));  // (2-D pieces are cut by column panel already)
        p->tuning.far_first = static_cast<int32_t>(far_window);
        // what this plan was built with (flex_plan_get_tuning)
        flex_plan_tuning &u = p->tuning;
        u.lanes_per_nz = G;
        u.chunk_records = static_cast<int32_t>(wave_nnz);
        u.long_row = static_cast<int32_t>(long_row);
        u.piece_records = static_cast<int32_t>(piece_len);
        u.row_cost = static_cast<int32_t>(row_cost);
        u.xcd_slices = xcd_dealt ? 3 : p->xcd_remap ? 1 : 2;
        u.rec_nt = p->rec_nt ? 1 : 2;
        u.tile_group = p->tile_group ? static_cast<int32_t>(p->tile_group) : 1;
        u.unroll = p->unroll;
        u.two_d = two_d ? 1 : 0;
        u.panel_kb = static_cast<int32_t>(panel_bytes >> 10);
        u.seg_min = static_cast<int32_t>(seg_min);
        u.lds_extra = static_cast<int32_t>(p->lds_extra);
        u.host_threads = host_threads();
        u.cluster = tn.cluster;
    }

    // Would the plan, cut the usual way, hold at least one chunk for every wave slot of the card (256 CUs x 32)?  Below that a launch is
    // bound by the latency of ONE wave's chain -- header, descriptors, records, gathers, store -- and not by instruction or memory
    // throughput.
    bool fills_the_chip() const {
        const double budget = std::clamp(16.0 * avg_deg, 128.0, 512.0);
        return (static_cast<double>(slice_nnz()) + 16.0 * m) / budget >= 8192.0;
    }
    // The rule for row bundles, measured on MI355X (profiles/r04_row_bundles.txt; kernel time per launch, bundles off -> on):
    //   k=32 (8 slots per step): epinions stand-in (131 828 rows of average degree 6.4) 19.3 -> 14.8 us, flickr shape (degree 11)
    //   16.1 -> 12.9, yelp shape (degree 19) 130.5 -> 120.2, reddit shape (degree 50, few short rows) 161.4 -> 160.9;
    //   k=64 (4 slots): 27.3 -> 25.2, 24.3 -> 21.8;  k=16 (16 slots): 18.1 -> 12.2, 14.8 -> 11.6;
    //   wave instructions per 64 multiply-adds on the epinions stand-in at k=32: 10.4 VALU + 10.5 SALU -> 3.9 + 3.7.
    // Candidate length: 12-16 records is the optimum on all three tiles (24-32: +2-7 %, longer chains and more padding inside a
    // bundle; 8: +0-6 %): 12 on the tiles of 8 and 16 slots (against 16: wiki-Vote shape -5 %, pubmed.csv k=16 -3 %, the rest
    // within 1 % either way), 16 on the 4-slot tile.  Graphs that do NOT fill the chip run one wave's chain long, and a bundle walks its rows' records one step
    // after the other where the plain form spreads a row over the slots of one step: with candidates of up to 32 records bundles
    // lost there (pubmed.csv k=32 5.14 -> 5.37 us), with 16 they win where the rows are short -- pubmed.csv (degree 5.5) k=16 / 32 /
    // 64 / 128: 5.10 -> 4.44, 5.17 -> 4.64, 6.68 -> 6.47, 8.14 -> 7.56 us; wiki-Vote shape (degree 12): 5.41 -> 5.56, 5.79 -> 5.80,
    // 7.05 -> 6.77, 8.31 -> 7.77 -- and lose where few rows are (ppi shape, degree 28: k=32 8.25 -> 8.41, k=128 18.8 -> 20.5).
    // Hence: on when the plan fills the chip, or the average degree is below 16.
    bool bundle_rule() const { return fills_the_chip() || avg_deg < 16.0; }

    // A run of `len` records as pieces: one, or (longer than a budget) several of about one budget.  The last piece to
    // arrive sums all of them with ONE wave, so a hub of 10^6 nonzeros cut into 10^4 budget-sized pieces spent 0.9 ms
    // in that sum alone (tools/probe_hub.py: 1263 us against 358 us without the hub): at most 256 pieces per run,
    // longer ones instead: 564 us (774 pieces: 601 us).
    void cut_run(std::vector<Piece> &out, uint32_t spos, uint32_t b, uint32_t e, uint32_t phase) const {
        const uint32_t len = e - b;
        if (len <= long_row) {
            out.push_back({spos, b, e, phase, 0u});
            return;
        }
        constexpr uint32_t kMaxPieces = 256;
        const uint32_t nchunk = std::min<uint32_t>((len + piece_len - 1) / piece_len, kMaxPieces);
        const uint32_t per = ((len + nchunk - 1) / nchunk + S - 1) / S * S;  // whole steps
        for (uint32_t c0 = b; c0 < e; c0 += per) out.push_back({spos, c0, std::min(e, c0 + per), phase, 1u});
    }

    int cut_rows_into_pieces() {
        row_first_piece.assign(static_cast<size_t>(m) + 1, 0u);
        rcol = A->col;
        rval = A->vals;
        if (two_d) return cut_by_column_panel();
        pieces.reserve(static_cast<size_t>(m) + 1024);
        for (int32_t i = 0; i < m; ++i) {
            const uint32_t r = sched[i];
            row_first_piece[i] = static_cast<uint32_t>(pieces.size());
            cut_run(pieces, static_cast<uint32_t>(i), A->rowPtr[r], A->rowPtr[r + 1], 0u);
        }
        row_first_piece[m] = static_cast<uint32_t>(pieces.size());
        return FLEX_OK;
    }

    // 2-D: the records of every row are re-grouped by column panel into pcol/pval (index e - e_base); piece ranges
    // point into those.  Parallel over blocks of schedule positions, block results concatenated in order.
    int cut_by_column_panel() {
        const uint32_t e_base = A->rowPtr[r0];
        pcol.resize(static_cast<size_t>(slice_nnz()));
        pval.resize(static_cast<size_t>(slice_nnz()));
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
                    auto copy_run = [&](size_t q) {
                        for (uint32_t z = run_beg[q]; z < run_beg[q + 1]; ++z) {
                            const uint32_t e = e0 + static_cast<uint32_t>(key[z] & 0xFFFFFFFFu);
                            pcol[o] = A->col[e];
                            pval[o] = A->vals[e];
                            ++o;
                        }
                    };
                    for (size_t q = 0; q + 1 < run_beg.size(); ++q) {
                        const uint32_t cnt = run_beg[q + 1] - run_beg[q];
                        if (cnt < seg_min) {
                            n_far += cnt;
                            continue;
                        }
                        copy_run(q);
                        ++n_kept;
                        cut_run(out, static_cast<uint32_t>(i), o - cnt, o, run_pan[q] + 1);
                    }
                    const uint32_t far_beg = o;
                    if (n_far) {
                        for (size_t q = 0; q + 1 < run_beg.size(); ++q)
                            if (run_beg[q + 1] - run_beg[q] < seg_min) copy_run(q);
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
        rcol = pcol.data();  // piece ranges are relative to e_base
        rval = pval.data();
        return FLEX_OK;
    }

    // 2-D: XCD slices of the rows, cut by records + row overhead (a slice's phases must stay on one XCD).
    void slice_rows_for_xcds() {
        std::fill(slice_row, slice_row + kXcds, 0u);
        slice_row[kXcds] = static_cast<uint32_t>(m);
        if (!two_d) return;
        std::vector<uint64_t> cum(static_cast<size_t>(m) + 1, 0);
        for (int32_t i = 0; i < m; ++i) {
            const uint32_t r = sched[i];
            cum[i + 1] = cum[i] + (A->rowPtr[r + 1] - A->rowPtr[r]) + row_cost;
        }
        for (int x = 1; x < kXcds; ++x)
            slice_row[x] = std::max<uint32_t>(slice_row[x - 1], static_cast<uint32_t>(std::lower_bound(cum.begin(), cum.end(), cum[m] * x / kXcds) - cum.begin()));
    }

    // emission order: 1-D = schedule order; 2-D = per slice, phase by phase (rows in schedule order inside a phase)
    void order_pieces() {
        emit.resize(pieces.size());
        std::iota(emit.begin(), emit.end(), 0u);
        if (!two_d) return;
        parallel_chunks(kXcds, [&](int64_t x) {
            auto b = emit.begin() + row_first_piece[slice_row[x]], e = emit.begin() + row_first_piece[slice_row[x + 1]];
            std::stable_sort(b, e, [&](uint32_t u, uint32_t v) { return pieces[u].phase < pieces[v].phase; });
        });
    }

    // rows summed from several pieces: partial slots (consecutive per row, in piece order) + arrival bookkeeping
    void number_split_rows() {
        row_sidx.assign(static_cast<size_t>(m), 0u);
        row_first_partial.assign(static_cast<size_t>(m), 0u);
        for (int32_t i = 0; i < m; ++i) {
            const uint32_t np = row_first_piece[i + 1] - row_first_piece[i];
            if (np <= 1) continue;
            const uint32_t r = sched[i];
            row_sidx[i] = static_cast<uint32_t>(split.size());
            row_first_partial[i] = n_partials;
            split.push_back({dst_of(r), n_partials, np});
            n_partials += np;
            split_nnz += A->rowPtr[r + 1] - A->rowPtr[r];
        }
    }

    // Which pieces become tasks of their own and which short rows share a bundle.  Candidates -- whole rows of at most `bundle_len`
    // records -- are collected while the pieces go by (everything else keeps its place), 8 x S at a time, sorted by length and cut into
    // groups of S: a group becomes a bundle when that takes fewer steps than its rows one after the other, a row by itself costing
    // its steps (padded to S records each) plus a reduction across the slots, a store and the scalar bookkeeping of a task end
    // (about 2.5 steps' worth of instructions on the G = 8 tile; a bundle's end about 1.5: one cross-lane read, one store).
    void form_tasks() {
        const uint32_t n_pieces = static_cast<uint32_t>(pieces.size());
        tasks.clear();
        tasks.reserve(n_pieces);
        if (n_partials >= kBundleFlag) bundles_on = false;  // t_dst keeps partial-slot ids below the bundle flag
        p->tuning.bundle = bundles_on ? 1 : 2;
        p->tuning.bundle_len = bundles_on ? static_cast<int32_t>(bundle_len) : 0;
        if (!bundles_on) {
            for (uint32_t t = 0; t < n_pieces; ++t) tasks.push_back({t, 0u, 0u});
            return;
        }
        constexpr uint32_t kRowEnd2 = 5, kBundleEnd2 = 3;  // in half steps
        auto len_of = [&](uint32_t e) { return pieces[emit[e]].end - pieces[emit[e]].beg; };
        std::vector<uint32_t> buf;
        auto flush_buf = [&]() {
            std::stable_sort(buf.begin(), buf.end(), [&](uint32_t a, uint32_t b) { return len_of(a) > len_of(b); });
            for (size_t g = 0; g < buf.size(); g += S) {
                const uint32_t n = static_cast<uint32_t>(std::min<size_t>(S, buf.size() - g));
                const uint32_t steps = len_of(buf[g]);
                uint32_t alone2 = 0;
                for (uint32_t s = 0; s < n; ++s) alone2 += 2 * ((len_of(buf[g + s]) + S - 1) / S) + kRowEnd2;
                if (n >= 2 && 2 * steps + kBundleEnd2 <= alone2) {
                    tasks.push_back({static_cast<uint32_t>(bundle_piece.size()), n, steps});
                    for (uint32_t s = 0; s < n; ++s) bundle_piece.push_back(buf[g + s]);
                    ++n_bundles;
                    bundle_rows += n;
                } else {
                    for (uint32_t s = 0; s < n; ++s) tasks.push_back({buf[g + s], 0u, 0u});
                }
            }
            buf.clear();
        };
        for (uint32_t e = 0; e < n_pieces; ++e) {
            const Piece &pc = pieces[emit[e]];
            if (pc.own_chunk == 0 && pc.end - pc.beg <= bundle_len) {  // 1-D: a piece that is not a slice of a long row is a whole row
                buf.push_back(e);
                if (buf.size() == 8u * S) flush_buf();
            } else {
                tasks.push_back({e, 0u, 0u});
            }
        }
        flush_buf();
    }

    // the chunks (one wave each) the tasks are packed into
    int pack_tasks_into_chunks() {
        const uint32_t n_tasks = static_cast<uint32_t>(tasks.size());
        t_beg.resize(static_cast<size_t>(n_tasks) + 1);
        t_dst.resize(n_tasks);
        t_aux.resize(n_tasks);
        uint64_t pos = 0;
        uint32_t wave_cost = 0, cur_phase = 0, prev_own = 0, bd_first = 0;
        int cur_slice = 0;
        for (uint32_t t = 0; t < n_tasks; ++t) {
            const TaskSpec &ts = tasks[t];
            const Piece &pc = pieces[emit[ts.rows ? bundle_piece[ts.ref] : ts.ref]];  // a bundle: its first (longest) row
            const uint32_t own = ts.rows ? 0u : pc.own_chunk;
            const uint32_t len = ts.rows ? ts.steps * S : pc.end - pc.beg;
            bool fresh = w_task.empty() || wave_cost >= wave_nnz || t - w_task.back() >= kMaxTasksPerWave || own || prev_own;
            if (ts.rows && static_cast<uint32_t>(bd_rows.size()) - bd_first + S > kBundleRowsPerChunk) fresh = true;
            if (two_d) {
                while (pc.spos >= slice_row[cur_slice + 1]) {  // first task of the next XCD slice
                    slice_chunk[++cur_slice] = static_cast<uint32_t>(w_task.size());
                    fresh = true;
                }
                if (pc.phase != cur_phase) fresh = true;  // a chunk never straddles two panels
                cur_phase = pc.phase;
            }
            if (fresh) {
                if (!w_task.empty()) chunk_bd.push_back(make_uint2(bd_first, static_cast<uint32_t>(bd_rows.size()) - bd_first));
                bd_first = static_cast<uint32_t>(bd_rows.size());
                w_task.push_back(t);
                wave_cost = 0;
            }
            wave_cost += len + row_cost;
            prev_own = own;
            t_beg[t] = static_cast<uint32_t>(pos);
            if (ts.rows) {
                t_dst[t] = kPartialFlag | kBundleFlag | (static_cast<uint32_t>(bd_rows.size()) - bd_first);
                t_aux[t] = make_uint2(static_cast<uint32_t>(bd_rows.size()), ts.steps);
                for (uint32_t s = 0; s < S; ++s) {
                    if (s < ts.rows) {
                        const Piece &row = pieces[emit[bundle_piece[ts.ref + s]]];
                        bd_rows.push_back(dst_of(sched[row.spos]) | (row.end == row.beg ? kBundleZero : 0u));
                    } else {
                        bd_rows.push_back(kBundleNoRow);
                    }
                }
                pos += len;
            } else {
                pos += (len + S - 1) / S * S;  // padded to a whole number of steps
            }
            if (pos >= (uint64_t(1) << 32)) return FLEX_ERR_UNSUPPORTED;  // 32-bit record offsets
        }
        t_beg[n_tasks] = static_cast<uint32_t>(pos);
        if (!w_task.empty()) chunk_bd.push_back(make_uint2(bd_first, static_cast<uint32_t>(bd_rows.size()) - bd_first));
        w_task.push_back(n_tasks);
        if (m == 0) w_task.assign(1, 0u);
        if (two_d)
            while (cur_slice < kXcds) slice_chunk[++cur_slice] = static_cast<uint32_t>(w_task.size() - 1);
        rec.resize(static_cast<size_t>(pos));
        return FLEX_OK;
    }

    void fill_records() {
        const uint32_t n_tasks = static_cast<uint32_t>(tasks.size());
        const uint32_t row_bytes32 = static_cast<uint32_t>(p->ldb) * 4u;
        constexpr int64_t kTaskBlk = 4096;
        auto record_of = [&](uint32_t e) {
            uint32_t c = rcol[e];
            if (col_map) c = static_cast<uint32_t>(col_map[c]);
            uint32_t bits;
            std::memcpy(&bits, &rval[e], 4);
            return make_uint2(p->off32 ? c * row_bytes32 : c, bits);
        };
        // Padding behind the last real record `last` of a row: n_pad more records, `stride` apart.  B row = the last real one (always a
        // valid address).  The padding does not carry value 0 -- 0 x inf would turn a row's +-inf into NaN (the oracle and the reference
        // have no padding) -- but SHARES the last real record's value: v = v/2 + v/4 + ... + v/2^p + v/2^p, every part exact
        // (power-of-two scaling), so a non-finite B value contributes what v itself would and a finite one the same product up to the
        // last rounding.  Values too small to be halved p times without leaving the normal range keep the plain zero padding.
        auto pad_row = [](uint2 *last, uint32_t n_pad, uint32_t stride) {
            if (n_pad == 0) return;
            const uint32_t ex = (last->y >> 23) & 0xFFu;  // biased exponent of v
            if (ex > n_pad + 1 && ex < 0xFFu) {
                float part;
                std::memcpy(&part, &last->y, 4);
                uint2 *q = last;  // the last real record takes v/2, the paddings v/4 ... v/2^p, v/2^p
                for (uint32_t i = 0; i < n_pad; ++i, q += stride) {
                    part *= 0.5f;
                    uint32_t bits;
                    std::memcpy(&bits, &part, 4);
                    q->y = bits;
                    q[stride] = make_uint2(q->x, bits);
                }
            } else {
                for (uint32_t i = 1; i <= n_pad; ++i) last[static_cast<size_t>(i) * stride] = make_uint2(last->x, 0u);
            }
        };
        parallel_chunks((static_cast<int64_t>(n_tasks) + kTaskBlk - 1) / kTaskBlk, [&](int64_t b) {
            for (int64_t t = b * kTaskBlk; t < std::min<int64_t>(n_tasks, (b + 1) * kTaskBlk); ++t) {
                const TaskSpec &ts = tasks[t];
                uint2 *const base = rec.data() + t_beg[t];
                if (ts.rows) {
                    // a bundle: record j of row s at [j][s]; slot 0 holds the longest row, so every step has a real record there
                    // whose B row the slots without a row (and the rows without a nonzero) gather with value 0 -- they store
                    // nothing (zeros), so what 0 x B makes of it is never seen
                    for (uint32_t s = 0; s < S; ++s) {
                        uint32_t len = 0;
                        if (s < ts.rows) {
                            const Piece &pc = pieces[emit[bundle_piece[ts.ref + s]]];
                            len = pc.end - pc.beg;
                            for (uint32_t j = 0; j < len; ++j) base[static_cast<size_t>(j) * S + s] = record_of(pc.beg + j);
                        }
                        if (len > 0) pad_row(base + static_cast<size_t>(len - 1) * S + s, ts.steps - len, S);
                        else
                            for (uint32_t j = 0; j < ts.steps; ++j) base[static_cast<size_t>(j) * S + s] = make_uint2(base[static_cast<size_t>(j) * S].x, 0u);
                    }
                    continue;
                }
                const uint32_t pi = emit[ts.ref];
                const Piece &pc = pieces[pi];
                const uint32_t i = pc.spos, r = sched[i];
                const uint32_t np = row_first_piece[i + 1] - row_first_piece[i];
                if (np > 1) {
                    t_dst[t] = kPartialFlag | (row_first_partial[i] + (pi - row_first_piece[i]));
                    t_aux[t] = make_uint2(row_sidx[i], np);
                } else {
                    t_dst[t] = dst_of(r);
                    t_aux[t] = make_uint2(0u, 0u);
                }
                uint2 *o = base;
                if (far_window == 0) {
                    for (uint32_t e = pc.beg; e < pc.end; ++e) *o++ = record_of(e);
                } else {
                    // FAR records first (tuning.far_first): a column whose vertex sits far from the row in the schedule is a likely L2
                    // miss, a near one a likely hit.  A wave's gathers return in issue order, so a group of U gathers waits for its
                    // slowest: with misses and hits interleaved nearly every group waits for the fabric, with the misses issued together
                    // only their groups do.  The order inside each class is kept; the sum order of a row changes, its value within
                    // rounding, reproducibly.
                    const auto is_far = [&](uint32_t e) {
                        const uint32_t c = rcol[e];
                        const int64_t cp = colpos.empty() ? static_cast<int64_t>(c) : static_cast<int64_t>(colpos[c]);
                        const int64_t d = cp - static_cast<int64_t>(colpos.empty() ? r : i);
                        return (d < 0 ? -d : d) > static_cast<int64_t>(far_window);
                    };
                    for (uint32_t e = pc.beg; e < pc.end; ++e)
                        if (is_far(e)) *o++ = record_of(e);
                    for (uint32_t e = pc.beg; e < pc.end; ++e)
                        if (!is_far(e)) *o++ = record_of(e);
                }
                // pad to a whole number of steps
                uint2 *const end = rec.data() + t_beg[t + 1];
                if (o > base) pad_row(o - 1, static_cast<uint32_t>(end - o), 1u);
            }
        });
        pcol = std::vector<uint32_t>();
        pval = std::vector<float>();
    }

    int upload_tasks() {
        p->n_tasks = static_cast<uint32_t>(tasks.size());
        p->n_records = rec.size();
        p->n_bundles = n_bundles;
        p->bundle_rows = bundle_rows;
        p->n_bd_rows = static_cast<uint32_t>(bd_rows.size());
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
        if (n_bundles && (rc = upload(&p->d_bd_rows, bd_rows, &p->device_bytes))) return rc;
        p->n_tiles = static_cast<uint32_t>(tiles.boff.size() / 32);
        p->n_row_tiles = tiles.rt_ptr.empty() ? 0u : static_cast<uint32_t>(tiles.rt_ptr.size() - 1);
        p->tile_nnz = tiles.nnz;
        if (p->n_tiles) {
            if ((rc = upload(&p->d_tile_a, tiles.a, &p->device_bytes))) return rc;
            if ((rc = upload(&p->d_tile_boff, tiles.boff, &p->device_bytes))) return rc;
            if ((rc = upload(&p->d_tile_mask, tiles.mask, &p->device_bytes))) return rc;
            if ((rc = upload(&p->d_rt_ptr, tiles.rt_ptr, &p->device_bytes))) return rc;
            if ((rc = upload(&p->d_rt_rows, tiles.rt_rows, &p->device_bytes))) return rc;
        }
        return FLEX_OK;
    }

    // Chunk table in launch order: the kernel gives XCD x the x-th eighth of it.  The eighths are cut
    // by COST (records + per-row and per-chunk overhead), not by chunk count, and padded with empty
    // chunks to a common length: schedules that put the heavy rows at one end (degree order, RCM,
    // Gorder) otherwise leave one XCD with up to 1.9x the mean work (flickr shape).
    // (2-D: the eighths are the row slices cut above -- a slice's phases must stay on one XCD.)
    int build_chunk_table() {
        const uint32_t n_real = static_cast<uint32_t>(w_task.size() - 1);
        auto header = [&](uint32_t c) {
            return make_uint4(w_task[c], w_task[c + 1] - w_task[c], t_beg[w_task[c]], t_beg[w_task[c + 1]]);
        };
        std::vector<uint4> chunk;
        std::vector<uint2> cbd;  // the chunks' bundle tables, placed like the headers (left empty when the plan has no bundle)
        auto place = [&](size_t at, uint32_t c) {
            chunk[at] = header(c);
            if (n_bundles) cbd[at] = chunk_bd[c];
        };
        p->tuning.xcd_balance = tn.xcd_balance == 2 ? 2 : 1;
        if (xcd_dealt && !two_d && n_real >= 8u * kXcds * kWavesPerBlock) {
            // Stretches of the schedule dealt to the XCDs in turn: XCD x walks stretches x, x + 8, x + 16, ...  Each XCD still has a
            // stretch (a community, or a part of one) to itself -- its L2 keeps what the stretch reuses -- while the eight of them
            // are on ADJACENT stretches at any time, so what neighbouring stretches share (the near ring of a community order) is
            // fetched by one XCD and found in the Infinity Cache by the others.  Chunks carry about the same number of records each,
            // so dealing by count balances the slices to within a stretch.
            const uint32_t per = static_cast<uint32_t>(pick(tn.xcd_stretch, 256)) * kWavesPerBlock;
            p->tuning.xcd_stretch = static_cast<int32_t>(per / kWavesPerBlock);
            const uint32_t n_st = (n_real + per - 1) / per;
            uint32_t len[kXcds] = {};
            for (uint32_t s = 0; s < n_st; ++s) len[s % kXcds] += std::min(per, n_real - s * per);
            uint32_t longest = *std::max_element(len, len + kXcds);
            longest = (longest + kWavesPerBlock - 1) / kWavesPerBlock * kWavesPerBlock;
            chunk.assign(static_cast<size_t>(longest) * kXcds, make_uint4(0u, 0u, 0u, 0u));
            if (n_bundles) cbd.assign(chunk.size(), make_uint2(0u, 0u));
            uint32_t at[kXcds] = {};
            for (uint32_t s = 0; s < n_st; ++s) {
                const uint32_t x = s % kXcds, c0 = s * per, c1 = std::min(n_real, c0 + per);
                for (uint32_t c = c0; c < c1; ++c) place(static_cast<size_t>(x) * longest + at[x]++, c);
            }
        } else if (two_d || (p->xcd_remap && n_real >= 8u * kXcds * kWavesPerBlock && tn.xcd_balance != 2)) {
            uint32_t cut[kXcds + 1];
            cut[0] = 0;
            cut[kXcds] = n_real;
            if (two_d) {
                for (uint32_t x = 1; x < kXcds; ++x) cut[x] = slice_chunk[x];
            } else {
                // cost of a chunk in units of one 512-byte gather (a record at k = 128): measured per-XCD times
                // on the flickr shape fit  t = a * records + ~20 a * chunks  with rows nearly free
                const uint64_t chunk_cost = static_cast<uint64_t>(pick(tn.chunk_cost, 16)) * 32u;
                const uint64_t task_cost = static_cast<uint64_t>(pick(tn.task_cost, 2)) * 32u;
                p->tuning.chunk_cost = static_cast<int32_t>(chunk_cost / 32u);
                p->tuning.task_cost = static_cast<int32_t>(task_cost / 32u);
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
            if (n_bundles) cbd.assign(chunk.size(), make_uint2(0u, 0u));
            for (uint32_t x = 0; x < kXcds; ++x)
                for (uint32_t c = cut[x]; c < cut[x + 1]; ++c) place(static_cast<size_t>(x) * longest + (c - cut[x]), c);
        } else {
            chunk.resize(n_real);
            if (n_bundles) cbd.resize(n_real);
            for (uint32_t c = 0; c < n_real; ++c) place(c, c);
        }
        p->n_chunks = n_real;
        p->n_slots = static_cast<uint32_t>(chunk.size());
        int rc;
        if ((rc = upload(&p->d_chunk, chunk, &p->device_bytes))) return rc;
        if (n_bundles && (rc = upload(&p->d_chunk_bd, cbd, &p->device_bytes))) return rc;
        if (flags & FLEX_PLAN_STATS) collect_stats(p, rec, chunk, split_nnz);
        // split-row workspace: the rows, one arrival counter per (row, column tile) -- zero between launches -- and the partial sums
        if ((rc = upload(&p->d_split, split, &p->device_bytes))) return rc;
        const size_t ktiles = (static_cast<size_t>(k) + 4 * G - 1) / (4 * G);
        std::vector<uint32_t> zeros(std::max<size_t>(1, split.size() * ktiles), 0u);
        if ((rc = upload(&p->d_split_cnt, zeros, &p->device_bytes))) return rc;
        const size_t pbytes = std::max<size_t>(1, static_cast<size_t>(n_partials) * k) * sizeof(float);
        // Large launches: split rows are summed by spmm_fixup_kernel after the main launch: stream order is all that needs, and it measured
        // FASTER than the in-launch form on the large shapes (MI355X, profiles/r03_fixup_in_launch_vs_two_launch.txt: reddit k=128
        // 638 vs 642 us, amazon 8.22 vs 8.27 ms; flickr 38.5 vs 36.7 us and reddit k=32 151 vs 150 the other way: one kernel
        // boundary).  tuning.split_rows = 1 asks for the in-launch form (relaxed agent atomics + sc1 hand-off: measured on
        // gfx950, not an architectural guarantee; spmm_kernels.hip).
        // Round 4: which form, by SIZE.  A launch of a few tens of microseconds pays the kernel boundary of the second launch in
        // full (flickr shape k=128: 36.5-36.7 us in-launch against 38.5-40.0 with spmm_fixup_kernel; reddit k=32, 150 us: level),
        // a launch of milliseconds gains from having no arrival atomics and reducer tails inside its 2 M waves -- so the in-launch
        // form is the rule up to 4e8 multiply-adds per launch and the two-launch form above.  tuning.split_rows = 1 / 2 forces.
        const double madds = static_cast<double>(slice_nnz()) * k;
        p->fused_fixup = tn.split_rows == 1 || (tn.split_rows != 2 && madds <= 4e8);
        p->tuning.split_rows = p->fused_fixup ? 1 : 2;
        FLEX_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_partial), pbytes));
        p->device_bytes += static_cast<int64_t>(pbytes);
        return FLEX_OK;
    }
};

}  // namespace

int build_plan(flex_plan *p, const flex_csr *A, int32_t r0, int32_t r1, const int32_t *col_map, const int32_t *dst_map, unsigned flags,
               const flex_plan_tuning &tuning, std::vector<uint32_t> *sched_cache, int force_G) try {
    // Hot blocks need the float4 path's shapes, 32-bit B offsets, whole 64-column tiles' worth of k, and no forced tile width
    // (autotune candidates stay flat).
    const bool block_shapes = p->k % 4 == 0 && p->k >= 64 && p->ldb % 4 == 0 && p->ldc % 4 == 0 &&
                              static_cast<uint64_t>(A->n) * static_cast<uint64_t>(p->ldb) * 4u <= (uint64_t(1) << 32);
    const bool can_block = block_shapes && force_G == 0 && r1 > r0 && tuning.mfma != 1 && tuning.two_d != 1 && tuning.blocks != 2;
    bool want_blocks = tuning.blocks == 1 && can_block;
    std::vector<uint32_t> local_cache;
    // The rule (tuning.blocks = 0), from what was measured on MI355X.  What stays flat after the split is, record for
    // record, an L2 miss (its launch time is cold nonzeros x 4k bytes at the fabric's ~7.3 TB/s), and the flat kernel ALONE already
    // serves the hot nonzeros out of its L2s underneath its own misses: the split pays only where the cold share is small and the
    // launch long enough for a second kernel -- the Amazon shape without uniformly random edges (29 % cold) 7.62 -> 6.51 ms; its
    // preset (39 % cold) 8.25 -> 9.5, the Reddit shapes (a tenth of the rows) 0.76-0.97x.  The look costs one sort of every 16th
    // block's columns; the schedule it needs is computed once and handed on.
    const int64_t nnz_in = static_cast<int64_t>(A->rowPtr[r1]) - A->rowPtr[r0];
    if (tuning.blocks == 0 && can_block && static_cast<int64_t>(r1 - r0) >= 480 * 2048 && nnz_in >= 48ll * (r1 - r0)) {
        if (!sched_cache) sched_cache = &local_cache;
        std::vector<uint32_t> *sched = nullptr, *colpos = nullptr;
        int rc;
        {
            PlanBuilder pre(p, A, r0, r1, col_map, dst_map, flags, tuning, sched_cache, 0);
            if ((rc = pre.schedule_only(&sched, &colpos))) return rc;
        }
        const double share = estimate_hot_share(A, *sched_cache, 480, 3, 16);
        if (plan_timing_enabled()) std::fprintf(stderr, "plan: hot share of 480-row blocks (thr 3, every 16th) %.3f\n", share);
        want_blocks = share >= 0.72;
    }
    return PlanBuilder(p, A, r0, r1, col_map, dst_map, flags, tuning, sched_cache, force_G, want_blocks).run();
} catch (const std::bad_alloc &) {  // any host allocation of any stage
    return FLEX_ERR_NOMEM;
}

}  // namespace flex
