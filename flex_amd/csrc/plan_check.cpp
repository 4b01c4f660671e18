// plan_check.cpp -- what is read back from a finished plan: the B-reuse / imbalance statistics of the planner's
// arrays, the self-check of the DEVICE image, and the measured per-CU imbalance (stamped twin of the kernel).
#include <algorithm>
#include <new>

#include "plan.h"

using namespace flex;

namespace flex {

// ≙ alpha_stats_collect (mat.cu:944-1065): distinct B rows per chunk / workgroup / XCD slice by
// stamping, and how evenly records are cut.  Padding records repeat the row's last column, so they
// change no distinct count.
void collect_stats(flex_plan *p, const RecordVec &rec, const std::vector<uint4> &chunk, int64_t split_nnz) {
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
    st.lds_hot_pct_2 = p->lds_hot[0];
    st.lds_hot_pct_4 = p->lds_hot[1];
    st.lds_u_2 = p->lds_u[0];
    st.lds_u_4 = p->lds_u[1];
    p->has_stats = true;
}

}  // namespace flex

extern "C" {

int flex_plan_measure_imbalance(flex_plan *p, const float *dB, float *dC, flex_stream_t stream, flex_imbalance *out) try {
    if (!p || !out || !dC || (!dB && p->nnz > 0)) return FLEX_ERR_INVALID;
    *out = flex_imbalance{};
    if (p->m == 0 || p->n_slots == 0) return FLEX_OK;
    if (!operands_vec4(p, dB, dC) || p->bk_blocks) return FLEX_ERR_UNSUPPORTED;  // the stamped twin exists for the vector kernel only (not for row blocks)
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
        rc = launch_spmm_stamped(plan_view(p, p->fused_fixup, d_log), p->lanes_per_nz, p->off32, dB, dC, s);
        if (rc == FLEX_OK && !p->fused_fixup) rc = launch_fixup(p->d_partial, p->d_split, p->n_split, p->k, p->ldc, dC, s);
        if (rc == FLEX_OK && p->n_tiles) {
            rc = launch_tiles(tile_view(p), p->off32, dB, dC, p->k, p->ldb, p->ldc, s);
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
    std::vector<uint2> chunk_bd(p->n_bundles ? p->n_slots : 0u);
    std::vector<uint32_t> bd_rows(p->n_bd_rows);
    std::vector<SplitRow> split(p->n_split);
    auto down = [&](void *dst, const void *src, size_t bytes) { return bytes == 0 || hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess; };
    const bool ok_copy = down(rec.data(), p->d_rec, rec.size() * sizeof(uint2)) && down(t_beg.data(), p->d_t_beg, t_beg.size() * 4) &&
                         down(t_dst.data(), p->d_t_dst, t_dst.size() * 4) && down(chunk.data(), p->d_chunk, chunk.size() * sizeof(uint4)) &&
                         down(split.data(), p->d_split, split.size() * sizeof(SplitRow)) && down(t_aux.data(), p->d_t_aux, t_aux.size() * sizeof(uint2)) &&
                         down(chunk_bd.data(), p->d_chunk_bd, chunk_bd.size() * sizeof(uint2)) && down(bd_rows.data(), p->d_bd_rows, bd_rows.size() * 4);
    if (cur != p->device) (void)hipSetDevice(cur);
    if (!ok_copy) return FLEX_ERR_HIP;
    if ((p->n_bundles != 0) != (p->d_bd_rows != nullptr) || (p->n_bundles != 0) != (p->d_chunk_bd != nullptr)) return FLEX_ERR_FORMAT;

    // tasks tile the record stream
    if (t_beg[0] != 0 || t_beg[p->n_tasks] != p->n_records) return FLEX_ERR_FORMAT;
    for (uint32_t t = 0; t < p->n_tasks; ++t)
        if (t_beg[t] > t_beg[t + 1]) return FLEX_ERR_FORMAT;
    // chunks tile the tasks (in table order, skipping the empty padding entries), each within the kernel's limits
    // ... and a chunk's bundles name consecutive groups of S entries of ITS part of bd_rows (what the wave holds in two registers per
    // lane), each bundle's records are its steps x S, and the parts of the chunks tile bd_rows
    const uint32_t S = 64u / static_cast<uint32_t>(p->lanes_per_nz);
    uint32_t next_task = 0, real = 0, bundles = 0;
    uint64_t bd_total = 0;
    std::vector<std::pair<uint32_t, uint32_t>> seen;  // real chunks as (first task, #tasks)
    std::vector<std::pair<uint32_t, uint32_t>> bd_parts;
    for (size_t ci = 0; ci < chunk.size(); ++ci) {
        const uint4 &c = chunk[ci];
        const uint2 cb = chunk_bd.empty() ? make_uint2(0u, 0u) : chunk_bd[ci];
        if (c.y == 0) {
            if (c.x | c.z | c.w | cb.x | cb.y) return FLEX_ERR_FORMAT;
            continue;
        }
        if (c.y > 63 || c.x + c.y > p->n_tasks || c.z != t_beg[c.x] || c.w != t_beg[c.x + c.y]) return FLEX_ERR_FORMAT;
        if (cb.y > kBundleRowsPerChunk || cb.y % S != 0 || static_cast<uint64_t>(cb.x) + cb.y > bd_rows.size()) return FLEX_ERR_FORMAT;
        uint32_t at = 0;
        for (uint32_t t = c.x; t < c.x + c.y; ++t) {
            if ((t_dst[t] & (kPartialFlag | kBundleFlag)) != (kPartialFlag | kBundleFlag)) continue;
            if ((t_dst[t] & ~(kPartialFlag | kBundleFlag)) != at || t_aux[t].x != cb.x + at) return FLEX_ERR_FORMAT;
            if (static_cast<uint64_t>(t_aux[t].y) * S != t_beg[t + 1] - t_beg[t]) return FLEX_ERR_FORMAT;
            at += S;
            ++bundles;
        }
        if (at != cb.y) return FLEX_ERR_FORMAT;
        if (cb.y) bd_parts.emplace_back(cb.x, cb.y);
        bd_total += cb.y;
        seen.emplace_back(c.x, c.y);
        ++real;
    }
    if (bundles != p->n_bundles || bd_total != bd_rows.size() || (S < kBundleMinSlots && bundles)) return FLEX_ERR_FORMAT;
    std::sort(bd_parts.begin(), bd_parts.end());
    uint32_t bd_next = 0;
    for (const auto &part : bd_parts) {
        if (part.first != bd_next) return FLEX_ERR_FORMAT;
        bd_next += part.second;
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
    int64_t in_bundles = 0;
    for (uint32_t t = 0; t < p->n_tasks; ++t) {
        const uint32_t d = t_dst[t];
        if ((d & (kPartialFlag | kBundleFlag)) == (kPartialFlag | kBundleFlag)) {  // a bundle: every slot a row of its own, or none
            int64_t rows_here = 0;
            for (uint32_t s2 = 0; s2 < S; ++s2) {
                const uint32_t row = bd_rows[t_aux[t].x + s2];
                if (row == kBundleNoRow || (row & kBundleZero)) {  // nothing of what this slot sums is stored: it must sum zeros
                    for (uint32_t j = 0; j < t_aux[t].y; ++j)
                        if (rec[t_beg[t] + static_cast<size_t>(j) * S + s2].y != 0) return FLEX_ERR_FORMAT;
                }
                if (row == kBundleNoRow) continue;
                const uint32_t dr = row & ~kBundleZero;
                if (dr >= p->c_rows || written[dr]++) return FLEX_ERR_FORMAT;
                ++rows_here;
            }
            if (rows_here == 0) return FLEX_ERR_FORMAT;
            in_bundles += rows_here;
        } else if (d & kPartialFlag) {
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
    if (in_bundles != p->bundle_rows) return FLEX_ERR_FORMAT;
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
    // hot blocks: the block table tiles hcol / cnt / the record streams, every wave's run counts add up to its stream and no run is
    // longer than one DPP row, every record names a row of its panel buffer (padding = the row of zeros with value 0), every staged
    // B row is valid, and every C row that a block adds to is a row the flat plan writes and is named by one slot only
    if (p->bk_blocks) {
        const uint32_t nb = p->bk_blocks, rounds = p->bk_rounds, P = p->bk_panel_rows, RB = rounds * kBkRowsPerRound;
        if ((rounds != 2 && rounds != 4 && rounds != 8) || P == 0 || P % 4 != 0 || P > kBkPanelMax || !p->off32) return FLEX_ERR_FORMAT;
        std::vector<uint4> hdr(nb);
        std::vector<uint2> wstart(static_cast<size_t>(nb) * kBkWaves), brec(static_cast<size_t>(p->bk_records));
        std::vector<uint32_t> brow(static_cast<size_t>(nb) * RB), link(static_cast<size_t>(nb) * RB);
        if (cur != p->device) FLEX_HIP_TRY(hipSetDevice(p->device));
        bool ok_b = down(hdr.data(), p->d_bk_hdr, hdr.size() * sizeof(uint4)) && down(wstart.data(), p->d_bk_wstart, wstart.size() * sizeof(uint2)) &&
                    down(brec.data(), p->d_bk_rec, brec.size() * sizeof(uint2)) && down(brow.data(), p->d_bk_brow, brow.size() * 4) &&
                    down(link.data(), p->d_bk_link, link.size() * 4);
        uint64_t n_cnt = 0, n_hcol = 0;
        for (const uint4 &h : hdr) n_cnt += static_cast<uint64_t>(h.w) * kBkWaves, n_hcol += static_cast<uint64_t>(h.x & 0x7FFFFFFFu) * P;
        std::vector<uint32_t> cnt(static_cast<size_t>(n_cnt)), hcol(static_cast<size_t>(n_hcol));
        ok_b = ok_b && down(cnt.data(), p->d_bk_cnt, cnt.size() * 4) && down(hcol.data(), p->d_bk_hcol, hcol.size() * 4);
        if (cur != p->device) (void)hipSetDevice(cur);
        if (!ok_b) return FLEX_ERR_HIP;
        for (uint32_t o : hcol)
            if (o % row_bytes != 0 || o / row_bytes >= static_cast<uint64_t>(p->n)) return FLEX_ERR_FORMAT;
        uint64_t at_hcol = 0, at_cnt = 0, at_step = 0;
        int64_t panels = 0, real = 0;
        std::vector<uint8_t> added(static_cast<size_t>(p->c_rows), 0);
        for (uint32_t b = 0; b < nb; ++b) {
            uint4 h = hdr[b];
            const bool chains = (h.x >> 31) != 0;
            h.x &= 0x7FFFFFFFu;
            if (h.x == 0) {
                if (h.w != 0 || chains) return FLEX_ERR_FORMAT;
            } else if (h.y != at_hcol || h.z != at_cnt || h.w != 2 * h.x || h.x > kBkMaxPanels || h.y % 4 != 0) {
                return FLEX_ERR_FORMAT;
            }
            at_hcol += static_cast<uint64_t>(h.x) * P;
            at_cnt += static_cast<uint64_t>(h.w) * kBkWaves;
            panels += h.x;
            for (uint32_t w = 0; w < kBkWaves; ++w) {
                const uint2 ws = wstart[static_cast<size_t>(b) * kBkWaves + w];
                if (ws.x != at_step) return FLEX_ERR_FORMAT;
                uint64_t pos = static_cast<uint64_t>(ws.x) * kBkSlots, steps = 0;
                for (uint32_t idx = 0; idx < h.x * kBkMaxRounds; ++idx) {  // panel-major, eight byte counts per panel (rounds beyond the block's: 0)
                    const uint32_t ph = idx / kBkMaxRounds, rd = idx % kBkMaxRounds;
                    const uint32_t n = (cnt[h.z + static_cast<size_t>(w) * h.w + 2 * ph + rd / 4] >> (8 * (rd & 3))) & 0xFFu;
                    if (rd >= rounds && n != 0) return FLEX_ERR_FORMAT;
                    if (n > kBkRunMax || pos + static_cast<uint64_t>(n) * kBkSlots > brec.size()) return FLEX_ERR_FORMAT;
                    for (uint64_t q = 0; q < static_cast<uint64_t>(n) * kBkSlots; ++q) {
                        const uint2 r = brec[pos + q];
                        if (r.x == kBkZeroRow) {
                            if (r.y != 0) return FLEX_ERR_FORMAT;
                        } else if (r.x % kBkRowBytes != 0 || r.x / kBkRowBytes >= P) {
                            return FLEX_ERR_FORMAT;
                        } else {
                            ++real;
                        }
                    }
                    pos += static_cast<uint64_t>(n) * kBkSlots;
                    steps += n;
                }
                if (steps != ws.y) return FLEX_ERR_FORMAT;
                at_step += steps;
            }
            // slots: an owner names a C row the flat plan writes, once over all blocks; the chain of a multi-part row starts at its
            // owner, runs through slots that are marked as parts and hold no row, and every such part is on exactly one chain
            std::vector<uint8_t> on_chain(RB, 0);
            bool any_link = false;
            for (uint32_t s = 0; s < RB; ++s) {
                const uint32_t row = brow[static_cast<size_t>(b) * RB + s], l = link[static_cast<size_t>(b) * RB + s];
                any_link = any_link || l != 0;
                if (l & ~(kBkLinkOwner | kBkLinkPart | 0xFFFFu)) return FLEX_ERR_FORMAT;
                if ((l & kBkLinkOwner) && ((l & kBkLinkPart) || (l & 0xFFFFu) == 0 || row == kBkEmptyRow)) return FLEX_ERR_FORMAT;
                if ((l & kBkLinkPart) && row != kBkEmptyRow) return FLEX_ERR_FORMAT;
                if (!(l & (kBkLinkOwner | kBkLinkPart)) && l != 0) return FLEX_ERR_FORMAT;
                if (row == kBkEmptyRow) continue;
                if (row >= p->c_rows || !written[row] || added[row]++) return FLEX_ERR_FORMAT;
                if (l & kBkLinkOwner) {
                    uint32_t hops = 0;
                    for (uint32_t n = l & 0xFFFFu; n != 0; n = link[static_cast<size_t>(b) * RB + n - 1] & 0xFFFFu) {
                        if (n > RB || ++hops > RB || !(link[static_cast<size_t>(b) * RB + n - 1] & kBkLinkPart) || on_chain[n - 1]++) return FLEX_ERR_FORMAT;
                    }
                }
            }
            for (uint32_t s = 0; s < RB; ++s)
                if (((link[static_cast<size_t>(b) * RB + s] & kBkLinkPart) != 0) != (on_chain[s] == 1)) return FLEX_ERR_FORMAT;
            if (any_link != chains) return FLEX_ERR_FORMAT;
        }
        if (at_step * kBkSlots != brec.size() || at_hcol != hcol.size() || at_cnt != cnt.size() || panels != p->bk_panels || real != p->bk_hot_nnz) return FLEX_ERR_FORMAT;
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

}  // extern "C"
