// block_plan.cpp -- the planner of the row-block path (kernel: block_kernels.hip; layout: internal.h, BlockView).
//
// ≙ what csr2_DiagTiling's rounds and csr2seg_Cmajor do in the reference (mat.cu:680-942, 1192-1269): confine a unit of
// work to a set of B rows small enough to stay on chip, so that a fetched B row is used `u` times (flex.cu:5513-5528).
// Re-thought for a CU with 160 KiB of LDS: the unit is a BLOCK of schedule-consecutive rows owned by one workgroup, "on
// chip" is LDS, and the set is not a column span but whatever columns the block's own nonzeros use at least `thr` times
// (its HOT columns: the members of the block's community, hubs) -- staged panel by panel, `panel_rows` B rows at a time.
//
//   walk the schedule       rows -> items (a row of up to `cap` records per slot takes 1 slot, longer ones 2 / 4 / 8 aligned
//                           slots; empty rows and rows beyond 8 x cap stay with the flat kernel), items -> blocks of R slots
//   per block (parallel)    count column uses -> hot columns -> panels; every nonzero becomes a COLD record {byte offset of
//                           its B row} or a HOT record {byte offset inside the panel buffer} of (slot, phase);
//                           slots of similar length are grouped 8 to a (wave, round), groups dealt to the 15 waves by
//                           longest-processing-time; a (wave, phase, round) group is as long as its longest slot
//   emit                    per wave ONE record stream [step][slot], phase-major; 16-bit step counts per (phase, round)
//
// The image does not depend on the number of host threads (blocks are independent and concatenated in order).
#include <algorithm>
#include <cstring>
#include <numeric>

#include "host_parallel.h"
#include "plan.h"

namespace flex {

namespace {

struct Item {
    uint32_t spos;  // position of the row in the schedule
    uint32_t v;     // slots it takes: 1, 2, 4 or 8; a HUB row takes 8 g slots = g whole groups on g different waves (2 <= g <= 15)
};

struct Entry {  // one nonzero of the block
    uint32_t col;   // column as stored in A
    uint32_t item;  // index of its row's item inside the block
    uint32_t seq;   // its position in the row
    float val;
};

struct BlockOut {
    uint4 hdr{};
    std::vector<uint2> wstart;  // 15: {first step (block-relative), steps}
    std::vector<uint32_t> cnt, hcol, brow, grp;
    std::vector<uint2> rec;
    int64_t hot_nnz = 0, nnz = 0, hot_cols = 0;
    bool failed = false;
};

}  // namespace

double estimate_hot_share(const flex_csr *A, const std::vector<uint32_t> &sched, uint32_t rows, uint32_t thr, int64_t stride, double *u) {
    const int64_t m = static_cast<int64_t>(sched.size());
    const int64_t nb = (m + rows - 1) / rows, ns = (nb + stride - 1) / stride;
    std::vector<int64_t> hot(static_cast<size_t>(ns), 0), all(static_cast<size_t>(ns), 0), staged(static_cast<size_t>(ns), 0);
    parallel_chunks(ns, [&](int64_t q) {
        const int64_t b = q * stride;
        std::vector<uint32_t> cols;
        for (int64_t i = b * rows; i < std::min<int64_t>(m, (b + 1) * rows); ++i) {
            const uint32_t r = sched[i];
            cols.insert(cols.end(), A->col + A->rowPtr[r], A->col + A->rowPtr[r + 1]);
        }
        std::sort(cols.begin(), cols.end());
        for (size_t z = 0; z < cols.size();) {
            size_t z1 = z;
            while (z1 < cols.size() && cols[z1] == cols[z]) ++z1;
            if (z1 - z >= thr) hot[q] += static_cast<int64_t>(z1 - z), ++staged[q];
            z = z1;
        }
        all[q] = static_cast<int64_t>(cols.size());
    });
    int64_t h = 0, a = 0, st = 0;
    for (int64_t q = 0; q < ns; ++q) h += hot[q], a += all[q], st += staged[q];
    if (u) *u = st > 0 ? static_cast<double>(h) / static_cast<double>(st) : 0.0;
    return a > 0 ? static_cast<double>(h) / static_cast<double>(a) : 0.0;
}

int build_blocks(const flex_csr *A, const std::vector<uint32_t> &sched, const std::vector<uint32_t> &colpos, const int32_t *col_map,
                 const int32_t *dst_map, int32_t r0, uint32_t row_bytes32, const BlockKnobs &kn, BlockImage &img, std::vector<uint32_t> &rest) {
    const int64_t m = static_cast<int64_t>(sched.size());
    const uint32_t rounds = kn.rounds, P = kn.panel_rows, RB = rounds * kBkRowsPerRound;
    const uint32_t max_panels = std::min<uint32_t>(kn.max_panels, kBkMaxCounts / rounds - 1);
    img = BlockImage{};
    img.rounds = rounds;
    img.panel_rows = P;
    rest.clear();
    // ---- rows -> items -> blocks (sequential: one pass over the row lengths)
    std::vector<Item> items;
    std::vector<uint32_t> blk_first;  // first item of each block (+ sentinel)
    uint32_t used = RB;               // forces the first block open
    for (int64_t i = 0; i < m; ++i) {
        const uint32_t r = sched[i];
        const uint32_t len = A->rowPtr[r + 1] - A->rowPtr[r];
        uint32_t v = 1;
        while (v < 8 && len > v * kn.cap) v <<= 1;
        if (len > v * kn.cap) {  // a hub: g whole groups, summed through LDS at the end of the tile
            const uint64_t g = (static_cast<uint64_t>(len) + 8ull * kn.cap - 1) / (8ull * kn.cap);
            v = g <= static_cast<uint64_t>(std::min<uint32_t>(kBkWaves, RB / kBkSlots)) ? static_cast<uint32_t>(8 * g) : 0u;
        }
        if (len == 0 || v == 0) {
            rest.push_back(static_cast<uint32_t>(i));
            continue;
        }
        if (used + v > RB) {
            blk_first.push_back(static_cast<uint32_t>(items.size()));
            used = 0;
        }
        used += v;
        items.push_back({static_cast<uint32_t>(i), v});
    }
    blk_first.push_back(static_cast<uint32_t>(items.size()));
    const int64_t nb = static_cast<int64_t>(blk_first.size()) - 1;
    if (nb <= 0) return FLEX_OK;
    if (nb >= (int64_t(1) << 31)) return FLEX_ERR_UNSUPPORTED;

    std::vector<BlockOut> out(static_cast<size_t>(nb));
    parallel_chunks(nb, [&](int64_t b) {
        BlockOut &o = out[static_cast<size_t>(b)];
        const Item *it = items.data() + blk_first[b];
        const uint32_t n_it = blk_first[b + 1] - blk_first[b];
        // ---- the block's nonzeros, and how often each column is used
        std::vector<Entry> ent;
        for (uint32_t x = 0; x < n_it; ++x) {
            const uint32_t r = sched[it[x].spos];
            for (uint32_t e = A->rowPtr[r]; e < A->rowPtr[r + 1]; ++e) ent.push_back({A->col[e], x, e - A->rowPtr[r], A->vals[e]});
        }
        o.nnz = static_cast<int64_t>(ent.size());
        std::vector<uint32_t> by_col(ent.size());
        std::iota(by_col.begin(), by_col.end(), 0u);
        std::sort(by_col.begin(), by_col.end(), [&](uint32_t a, uint32_t c) { return ent[a].col != ent[c].col ? ent[a].col < ent[c].col : a < c; });
        struct Hot {
            uint32_t col, uses, pos;
        };
        std::vector<Hot> hot;
        for (size_t z = 0; z < by_col.size();) {
            size_t z1 = z;
            while (z1 < by_col.size() && ent[by_col[z1]].col == ent[by_col[z]].col) ++z1;
            const uint32_t c = ent[by_col[z]].col;
            if (z1 - z >= kn.thr) hot.push_back({c, static_cast<uint32_t>(z1 - z), colpos.empty() ? c : colpos[c]});
            z = z1;
        }
        if (hot.size() > static_cast<size_t>(max_panels) * P) {  // more than fits the phase counts: the most used ones
            std::nth_element(hot.begin(), hot.begin() + static_cast<size_t>(max_panels) * P, hot.end(),
                             [](const Hot &a, const Hot &c) { return a.uses != c.uses ? a.uses > c.uses : a.col < c.col; });
            hot.resize(static_cast<size_t>(max_panels) * P);
        }
        // a last panel that would hold only a few rows costs a barrier and a DMA round for little: leave those columns cold
        if (hot.size() % P != 0 && hot.size() % P < kn.min_last_panel && hot.size() > P) {
            std::sort(hot.begin(), hot.end(), [](const Hot &a, const Hot &c) { return a.uses != c.uses ? a.uses > c.uses : a.col < c.col; });
            hot.resize(hot.size() / P * P);
        }
        std::sort(hot.begin(), hot.end(), [](const Hot &a, const Hot &c) { return a.pos != c.pos ? a.pos < c.pos : a.col < c.col; });
        const uint32_t np = static_cast<uint32_t>((hot.size() + P - 1) / P);
        const uint32_t n_ph = np + 1;
        o.hot_cols = static_cast<int64_t>(hot.size());
        // hcol: np x P byte offsets, the tail of the last panel padded with its last row (a valid address)
        o.hcol.resize(static_cast<size_t>(np) * P);
        for (size_t h = 0; h < o.hcol.size(); ++h) {
            const uint32_t c = hot[std::min(h, hot.size() - 1)].col;
            o.hcol[h] = (col_map ? static_cast<uint32_t>(col_map[c]) : c) * row_bytes32;
        }
        // ---- slots: an item of v slots deals its records round-robin (per phase) over them
        // items sorted by (v desc, records per slot desc): aligned placement for free, similar lengths side by side
        std::vector<uint32_t> ord(n_it);
        std::iota(ord.begin(), ord.end(), 0u);
        auto len_of = [&](uint32_t x) {
            const uint32_t r = sched[it[x].spos];
            return A->rowPtr[r + 1] - A->rowPtr[r];
        };
        std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t c) {
            if (it[a].v != it[c].v) return it[a].v > it[c].v;
            return (len_of(a) + it[a].v - 1) / it[a].v > (len_of(c) + it[c].v - 1) / it[c].v;
        });
        std::vector<uint32_t> first_slot(n_it);
        uint32_t n_slots = 0;
        for (uint32_t x : ord) {
            first_slot[x] = n_slots;
            n_slots += it[x].v;
        }
        const uint32_t n_groups = rounds * kBkWaves;  // RB / 8
        // hot lookup per entry: walk the column-sorted entries against the column-sorted hot list
        std::vector<uint32_t> ent_phase(ent.size(), 0u), ent_off(ent.size());
        {
            std::vector<std::pair<uint32_t, uint32_t>> hot_by_col(hot.size());  // (col, index in panel order)
            for (size_t h = 0; h < hot.size(); ++h) hot_by_col[h] = {hot[h].col, static_cast<uint32_t>(h)};
            std::sort(hot_by_col.begin(), hot_by_col.end());
            size_t h = 0;
            for (size_t z = 0; z < by_col.size(); ++z) {
                const uint32_t e = by_col[z], c = ent[e].col;
                while (h < hot_by_col.size() && hot_by_col[h].first < c) ++h;
                if (h < hot_by_col.size() && hot_by_col[h].first == c) {
                    const uint32_t idx = hot_by_col[h].second;
                    ent_phase[e] = 1 + idx / P;
                    ent_off[e] = (idx % P) * kBkRowBytes;
                    ++o.hot_nnz;
                } else {
                    ent_off[e] = (col_map ? static_cast<uint32_t>(col_map[c]) : c) * row_bytes32;
                }
            }
        }
        // records of every (slot, phase): entries are in (item, seq) order already; deal them out
        std::vector<uint32_t> cnt_sp(static_cast<size_t>(RB) * n_ph, 0u);  // [slot][phase]
        std::vector<uint32_t> ent_slot(ent.size());
        {
            std::vector<uint32_t> rr(n_ph);
            size_t e = 0;
            for (uint32_t x = 0; x < n_it; ++x) {
                std::fill(rr.begin(), rr.end(), 0u);
                const uint32_t len = len_of(x), v = it[x].v, s0 = first_slot[x];
                for (uint32_t q = 0; q < len; ++q, ++e) {
                    const uint32_t ph = ent_phase[e], s = s0 + (rr[ph]++ % v);
                    ent_slot[e] = s;
                    ++cnt_sp[static_cast<size_t>(s) * n_ph + ph];
                }
            }
        }
        std::vector<uint32_t> beg_sp(static_cast<size_t>(RB) * n_ph + 1, 0u);
        for (size_t q = 0; q < cnt_sp.size(); ++q) beg_sp[q + 1] = beg_sp[q] + cnt_sp[q];
        std::vector<uint2> rec_sp(ent.size());
        {
            std::vector<uint32_t> fill(beg_sp.begin(), beg_sp.end() - 1);
            for (size_t e = 0; e < ent.size(); ++e) {
                uint32_t bits;
                std::memcpy(&bits, &ent[e].val, 4);
                rec_sp[fill[static_cast<size_t>(ent_slot[e]) * n_ph + ent_phase[e]]++] = make_uint2(ent_off[e], bits);
            }
        }
        // what a padding record of the cold phase points at: a column the slot's row uses anyway
        std::vector<uint32_t> pad_off(RB, 0u);
        {
            size_t e = 0;
            for (uint32_t x = 0; x < n_it; ++x) {
                const uint32_t c = ent[e].col;  // len >= 1
                const uint32_t off = (col_map ? static_cast<uint32_t>(col_map[c]) : c) * row_bytes32;
                for (uint32_t s = 0; s < it[x].v; ++s) pad_off[first_slot[x] + s] = off;
                e += len_of(x);
            }
        }
        // ---- groups of 8 slots -> (wave, round): longest processing time first
        std::vector<uint64_t> g_cost(n_groups, 0);
        for (uint32_t g = 0; g < n_groups; ++g)
            for (uint32_t ph = 0; ph < n_ph; ++ph) {
                uint32_t mx = 0;
                for (uint32_t s = 0; s < kBkSlots; ++s) mx = std::max(mx, cnt_sp[static_cast<size_t>(g * kBkSlots + s) * n_ph + ph]);
                g_cost[g] += mx;
            }
        std::vector<uint32_t> g_ord(n_groups);
        std::iota(g_ord.begin(), g_ord.end(), 0u);
        std::stable_sort(g_ord.begin(), g_ord.end(), [&](uint32_t a, uint32_t c) { return g_cost[a] > g_cost[c]; });
        std::vector<uint64_t> w_load(kBkWaves, 0);
        std::vector<uint32_t> w_n(kBkWaves, 0), grp_of(static_cast<size_t>(kBkWaves) * rounds, 0xFFFFFFFFu);  // [wave][round]
        // hub rows first: the g groups of one row go to g DIFFERENT waves (each wave sums its part; the parts meet in LDS)
        std::vector<uint32_t> g_info(n_groups, 0u);  // what the kernel reads per group: part | g << 8 | scratch slot << 16
        std::vector<uint8_t> g_placed(n_groups, 0);
        uint32_t scratch_next = 0;
        for (uint32_t x : ord) {
            if (it[x].v <= 8) break;  // ord: widest first
            const uint32_t ng = it[x].v / 8, g0 = first_slot[x] / kBkSlots;
            std::vector<uint32_t> waves(kBkWaves);
            std::iota(waves.begin(), waves.end(), 0u);
            // most free rounds first (keeps the free rounds even, so that later hubs still find enough different waves), then least loaded
            std::stable_sort(waves.begin(), waves.end(), [&](uint32_t a, uint32_t c) { return w_n[a] != w_n[c] ? w_n[a] < w_n[c] : w_load[a] < w_load[c]; });
            uint32_t part = 0;
            for (uint32_t w : waves) {
                if (part == ng) break;
                if (w_n[w] >= rounds) continue;
                const uint32_t g = g0 + part;
                grp_of[static_cast<size_t>(w) * rounds + w_n[w]++] = g;
                w_load[w] += g_cost[g];
                g_info[g] = part | (ng << 8) | (scratch_next << 16);
                g_placed[g] = 1;
                ++part;
            }
            if (part != ng) {  // cannot happen: hubs are placed first and ng <= 15 waves with a free round each
                o.failed = true;
                return;
            }
            scratch_next += ng - 1;
        }
        for (uint32_t g : g_ord) {
            if (g_placed[g]) continue;
            uint32_t best = kBkWaves;
            for (uint32_t w = 0; w < kBkWaves; ++w)
                if (w_n[w] < rounds && (best == kBkWaves || w_load[w] < w_load[best])) best = w;
            grp_of[static_cast<size_t>(best) * rounds + w_n[best]++] = g;
            w_load[best] += g_cost[g];
        }
        // ---- emit
        // brow[round][wave][slot]
        std::vector<uint32_t> slot_row(RB, kBkEmptyRow);
        for (uint32_t x = 0; x < n_it; ++x) {
            const uint32_t r = sched[it[x].spos];
            const uint32_t dst = dst_map ? static_cast<uint32_t>(dst_map[r]) : r - static_cast<uint32_t>(r0);
            uint32_t vcode = 0;
            while ((1u << vcode) < std::min<uint32_t>(it[x].v, 8)) ++vcode;  // a hub's groups are 8-slot rows to the butterfly
            for (uint32_t s = 0; s < it[x].v; ++s) slot_row[first_slot[x] + s] = dst | (vcode << 29);
        }
        o.brow.assign(RB, kBkEmptyRow);
        o.grp.assign(static_cast<size_t>(rounds) * kBkWaves, 0u);
        const uint32_t cw = (n_ph * rounds + 1) / 2;
        o.cnt.assign(static_cast<size_t>(kBkWaves) * cw, 0u);
        o.wstart.resize(kBkWaves);
        uint32_t step_pos = 0;
        for (uint32_t w = 0; w < kBkWaves; ++w) {
            const uint32_t w_first = step_pos;
            for (uint32_t rd = 0; rd < rounds; ++rd) {
                const uint32_t g = grp_of[static_cast<size_t>(w) * rounds + rd];
                for (uint32_t s = 0; s < kBkSlots; ++s) o.brow[(static_cast<size_t>(rd) * kBkWaves + w) * kBkSlots + s] = slot_row[g * kBkSlots + s];
                o.grp[static_cast<size_t>(rd) * kBkWaves + w] = g_info[g];
            }
            for (uint32_t ph = 0; ph < n_ph; ++ph)
                for (uint32_t rd = 0; rd < rounds; ++rd) {
                    const uint32_t g = grp_of[static_cast<size_t>(w) * rounds + rd];
                    uint32_t steps = 0;
                    for (uint32_t s = 0; s < kBkSlots; ++s) steps = std::max(steps, cnt_sp[static_cast<size_t>(g * kBkSlots + s) * n_ph + ph]);
                    const uint32_t idx = ph * rounds + rd;
                    o.cnt[static_cast<size_t>(w) * cw + idx / 2] |= steps << (16 * (idx & 1));
                    const size_t base = o.rec.size();
                    o.rec.resize(base + static_cast<size_t>(steps) * kBkSlots);
                    if (ph == 0) {
                        for (uint32_t s = 0; s < kBkSlots; ++s) {
                            const size_t sp = static_cast<size_t>(g * kBkSlots + s) * n_ph;
                            const uint32_t have = cnt_sp[sp];
                            const uint2 pad = make_uint2(pad_off[g * kBkSlots + s], 0u);
                            for (uint32_t q = 0; q < steps; ++q) o.rec[base + static_cast<size_t>(q) * kBkSlots + s] = q < have ? rec_sp[beg_sp[sp] + q] : pad;
                        }
                    } else {
                        // LDS banks.  One ds_read_b128 of the wave serves its 64 lanes in four groups of 16 (MI355X_MICROARCH.md, LDS);
                        // with 8 lanes per slot a group holds a 64-byte half of FOUR slots' rows, and the halves of slots (0,3), (1,2),
                        // (4,7), (5,6) land on the same 16 banks whenever the two panel rows have the same parity (a row is 128 bytes = 32
                        // of the 64 banks): a 2-way conflict in half of the steps, measured as 7 instead of 4 cycles per instruction --
                        // the panel phases ARE LDS-bound.  The order of a slot's records inside a (panel, round) group is free, so the
                        // first slot of each pair takes its even rows first and the second its odd rows first; a padding record takes
                        // the zero row of the parity its partner does not use.
                        static const uint8_t kPartner[kBkSlots] = {3, 2, 1, 0, 7, 6, 5, 4};
                        static const uint8_t kEvenFirst[kBkSlots] = {1, 1, 0, 0, 1, 1, 0, 0};
                        std::vector<uint2> lst[kBkSlots];
                        for (uint32_t s = 0; s < kBkSlots; ++s) {
                            const size_t sp = static_cast<size_t>(g * kBkSlots + s) * n_ph + ph;
                            lst[s].assign(rec_sp.begin() + beg_sp[sp], rec_sp.begin() + beg_sp[sp] + cnt_sp[sp]);
                            const uint32_t want = kEvenFirst[s] ? 0u : 1u;  // parity of the rows that go first
                            std::stable_partition(lst[s].begin(), lst[s].end(), [&](const uint2 &r) { return ((r.x / kBkRowBytes) & 1u) == want; });
                        }
                        for (uint32_t q = 0; q < steps; ++q)
                            for (uint32_t s = 0; s < kBkSlots; ++s) {
                                uint2 r;
                                if (q < lst[s].size()) {
                                    r = lst[s][q];
                                } else {
                                    const uint32_t t = kPartner[s];
                                    const uint32_t partner_parity = q < lst[t].size() ? (lst[t][q].x / kBkRowBytes) & 1u : 1u;
                                    r = make_uint2(kBkZeroRow + (partner_parity ? 0u : kBkRowBytes), 0u);  // zero rows at panel rows kBkPanelMax (even) and +1 (odd)
                                }
                                o.rec[base + static_cast<size_t>(q) * kBkSlots + s] = r;
                            }
                    }
                    step_pos += steps;
                }
            o.wstart[w] = make_uint2(w_first, step_pos - w_first);
        }
        o.hdr = make_uint4(np | (scratch_next ? 0x80000000u : 0u), 0u, 0u, cw);
    });

    // ---- concatenate (offsets are sequential; the copies run in parallel)
    std::vector<uint64_t> rec_at(static_cast<size_t>(nb) + 1, 0), cnt_at(static_cast<size_t>(nb) + 1, 0), hcol_at(static_cast<size_t>(nb) + 1, 0);
    for (int64_t b = 0; b < nb; ++b) {
        const BlockOut &o = out[static_cast<size_t>(b)];
        if (o.failed || o.hdr.w * 2 > kBkMaxCounts || (o.hdr.x & 0x7FFFFFFFu) > max_panels) return FLEX_ERR_UNSUPPORTED;  // cannot happen: max_panels bounds both
        for (const uint2 &ws : o.wstart)
            if (ws.y > 0xFFFFFFFFu / 8u) return FLEX_ERR_UNSUPPORTED;
        rec_at[b + 1] = rec_at[b] + o.rec.size();
        cnt_at[b + 1] = cnt_at[b] + o.cnt.size();
        hcol_at[b + 1] = hcol_at[b] + o.hcol.size();
        img.nnz += o.nnz;
        img.hot_nnz += o.hot_nnz;
        img.hot_cols += o.hot_cols;
        img.panels += o.hdr.x & 0x7FFFFFFFu;
    }
    if (rec_at[nb] / kBkSlots >= (uint64_t(1) << 32) || cnt_at[nb] >= (uint64_t(1) << 32) || hcol_at[nb] >= (uint64_t(1) << 32)) return FLEX_ERR_UNSUPPORTED;
    img.n_blocks = static_cast<uint32_t>(nb);
    img.rows = static_cast<int64_t>(items.size());
    img.hdr.resize(static_cast<size_t>(nb));
    img.wstart.resize(static_cast<size_t>(nb) * kBkWaves);
    img.brow.resize(static_cast<size_t>(nb) * RB);
    img.grp.resize(static_cast<size_t>(nb) * rounds * kBkWaves);
    img.cnt.resize(static_cast<size_t>(cnt_at[nb]));
    img.hcol.resize(static_cast<size_t>(hcol_at[nb]));
    img.rec.resize(static_cast<size_t>(rec_at[nb]));
    parallel_chunks(nb, [&](int64_t b) {
        BlockOut &o = out[static_cast<size_t>(b)];
        img.hdr[b] = make_uint4(o.hdr.x, static_cast<uint32_t>(hcol_at[b]), static_cast<uint32_t>(cnt_at[b]), o.hdr.w);
        const uint32_t step0 = static_cast<uint32_t>(rec_at[b] / kBkSlots);
        for (uint32_t w = 0; w < kBkWaves; ++w) img.wstart[static_cast<size_t>(b) * kBkWaves + w] = make_uint2(step0 + o.wstart[w].x, o.wstart[w].y);
        std::copy(o.brow.begin(), o.brow.end(), img.brow.begin() + static_cast<size_t>(b) * RB);
        std::copy(o.grp.begin(), o.grp.end(), img.grp.begin() + static_cast<size_t>(b) * rounds * kBkWaves);
        std::copy(o.cnt.begin(), o.cnt.end(), img.cnt.begin() + cnt_at[b]);
        std::copy(o.hcol.begin(), o.hcol.end(), img.hcol.begin() + hcol_at[b]);
        std::copy(o.rec.begin(), o.rec.end(), img.rec.begin() + rec_at[b]);
        o = BlockOut{};  // free as we go
    });
    return FLEX_OK;
}

}  // namespace flex
