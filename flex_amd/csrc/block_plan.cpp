// block_plan.cpp -- the planner of the hot-block path (kernel: block_kernels.hip; layout: internal.h, BlockView).
//
// ≙ what csr2_DiagTiling's rounds and csr2seg_Cmajor do in the reference (mat.cu:680-942, 1192-1269): confine a unit of
// work to a set of B rows small enough to stay on chip, so that a fetched B row is used `u` times (flex.cu:5513-5528).
// Re-thought for a CU with 160 KiB of LDS, and (round 4) as a SPLIT of the matrix rather than a second way of doing all of it:
// the unit is a BLOCK of R = rounds x 60 schedule-consecutive rows owned by one workgroup, "on chip" is LDS, and the set is
// whatever columns the block's own nonzeros use at least `thr` times (its HOT columns: the members of the block's community,
// hubs) -- staged panel by panel, `panel_rows` B rows at a time.  Only the nonzeros in those columns go into the block image;
// the others (by construction the ones without reuse on chip: the L2 misses of any schedule) are handed back to the flat
// planner through `hot_mask`, and so is everything awkward: rows too long for one slot, the 17th and later nonzeros of one
// row in one panel, columns beyond the panel budget.
//
//   walk the schedule       a row of len nonzeros takes ceil(len / cap) slots (its PARTS: each accumulates a share of the row's hot
//                           nonzeros, the first one -- the OWNER -- collects the others through LDS at the end of the tile and
//                           writes the row); R slots to a block; empty rows and rows of more than R slots take none
//   per block (parallel)    count column uses -> hot columns -> panels (in schedule order of the columns); every hot nonzero
//                           becomes a record {byte offset inside the panel buffer, value} of (slot, panel);
//                           slots with similar panel profiles are grouped 4 to a (wave, round), groups dealt to the 15 waves by
//                           longest-processing-time; a RUN (wave, panel, round) is as long as its longest slot (<= 16 steps)
//   emit                    per wave ONE record stream [step][slot], panel-major; per (wave, panel) the 8-bit step counts of its runs
//
// The image does not depend on the number of host threads (blocks are independent and concatenated in order).
#include <algorithm>
#include <cstring>
#include <numeric>

#include "host_parallel.h"
#include "plan.h"

namespace flex {

namespace {

struct Entry {  // one nonzero of the block
    uint32_t col;   // column as stored in A
    uint32_t item;  // index of its row inside the block
    uint32_t e;     // its index in A's arrays, relative to the first entry of the planned row range
    float val;
};

struct BlockOut {
    uint4 hdr{};
    std::vector<uint2> wstart;  // 15: {first step (block-relative), steps}
    std::vector<uint32_t> cnt, hcol, brow, link;
    std::vector<uint2> rec;
    int64_t hot_nnz = 0, nnz = 0, hot_cols = 0, rows = 0;
    int64_t cand_nnz = 0, lost_panels = 0, lost_last = 0, lost_run = 0;  // FLEX_PLAN_TIMING: where candidates (column uses >= thr) were left cold
    bool multi = false;  // some row of the block has several parts
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
                 const int32_t *dst_map, int32_t r0, uint32_t row_bytes32, const BlockKnobs &kn, BlockImage &img, std::vector<uint8_t> &hot_mask) {
    const int64_t m = static_cast<int64_t>(sched.size());
    const uint32_t rounds = kn.rounds, P = kn.panel_rows, RB = rounds * kBkRowsPerRound;
    const uint32_t max_panels = std::min<uint32_t>(kn.max_panels, kBkMaxPanels);
    const uint32_t run_max = std::min<uint32_t>(std::max<uint32_t>(kn.run_max, 1u), kBkRunMax);
    img = BlockImage{};
    img.rounds = rounds;
    img.panel_rows = P;
    const uint32_t e_base = m > 0 ? A->rowPtr[r0] : 0u;
    const size_t nnz_in = m > 0 ? static_cast<size_t>(A->rowPtr[r0 + m] - e_base) : 0;
    hot_mask.assign(nnz_in, 0);
    // ---- rows -> items -> blocks (sequential: one pass over the row lengths)
    struct Item {
        uint32_t spos, v;  // schedule position of the row, slots it takes
    };
    std::vector<Item> items;
    std::vector<uint32_t> blk_first;  // first item of each block (+ sentinel)
    uint32_t used = RB;               // forces the first block open
    for (int64_t i = 0; i < m; ++i) {
        const uint32_t r = sched[i];
        const uint32_t len = A->rowPtr[r + 1] - A->rowPtr[r];
        if (len == 0) continue;
        const uint64_t v = (static_cast<uint64_t>(len) + kn.cap - 1) / kn.cap;
        if (v > RB) continue;  // longer than a whole block of slots: stays flat
        if (used + v > RB) {
            blk_first.push_back(static_cast<uint32_t>(items.size()));
            used = 0;
        }
        used += static_cast<uint32_t>(v);
        items.push_back({static_cast<uint32_t>(i), static_cast<uint32_t>(v)});
    }
    blk_first.push_back(static_cast<uint32_t>(items.size()));
    const int64_t nb = static_cast<int64_t>(blk_first.size()) - 1;
    if (nb <= 0 || items.empty()) return FLEX_OK;
    if (nb >= (int64_t(1) << 31)) return FLEX_ERR_UNSUPPORTED;

    std::vector<BlockOut> out(static_cast<size_t>(nb));
    parallel_chunks(nb, [&](int64_t b) {
        BlockOut &o = out[static_cast<size_t>(b)];
        const Item *it = items.data() + blk_first[b];
        const uint32_t n_it = blk_first[b + 1] - blk_first[b];
        o.rows = n_it;
        // slots: item x holds slots [first_slot[x], first_slot[x] + v): its parts
        std::vector<uint32_t> first_slot(n_it);
        uint32_t n_sl = 0;
        for (uint32_t x = 0; x < n_it; ++x) {
            first_slot[x] = n_sl;
            n_sl += it[x].v;
            if (it[x].v > 1) o.multi = true;
        }
        // ---- the block's nonzeros, and how often each column is used
        std::vector<Entry> ent;
        for (uint32_t x = 0; x < n_it; ++x) {
            const uint32_t r = sched[it[x].spos];
            for (uint32_t e = A->rowPtr[r]; e < A->rowPtr[r + 1]; ++e) ent.push_back({A->col[e], x, e - e_base, A->vals[e]});
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
        for (const Hot &h : hot) o.cand_nnz += h.uses;
        if (hot.size() > static_cast<size_t>(max_panels) * P) {  // more than fits the run counts: the most used ones
            std::nth_element(hot.begin(), hot.begin() + static_cast<size_t>(max_panels) * P, hot.end(),
                             [](const Hot &a, const Hot &c) { return a.uses != c.uses ? a.uses > c.uses : a.col < c.col; });
            for (size_t h = static_cast<size_t>(max_panels) * P; h < hot.size(); ++h) o.lost_panels += hot[h].uses;
            hot.resize(static_cast<size_t>(max_panels) * P);
        }
        // a last panel that would hold only a few rows costs a barrier and a DMA round for little: those columns stay with the flat kernel
        if (hot.size() % P != 0 && hot.size() % P < kn.min_last_panel) {
            std::sort(hot.begin(), hot.end(), [](const Hot &a, const Hot &c) { return a.uses != c.uses ? a.uses > c.uses : a.col < c.col; });
            for (size_t h = hot.size() / P * P; h < hot.size(); ++h) o.lost_last += hot[h].uses;
            hot.resize(hot.size() / P * P);
        }
        std::sort(hot.begin(), hot.end(), [](const Hot &a, const Hot &c) { return a.pos != c.pos ? a.pos < c.pos : a.col < c.col; });
        const uint32_t np = static_cast<uint32_t>((hot.size() + P - 1) / P);
        o.hot_cols = static_cast<int64_t>(hot.size());
        o.brow.assign(RB, kBkEmptyRow);
        o.link.assign(RB, 0u);
        o.wstart.assign(kBkWaves, make_uint2(0u, 0u));
        if (np == 0) {
            o.hdr = make_uint4(0u, 0u, 0u, 0u);
            return;
        }
        // hcol: np x P byte offsets, the tail of the last panel padded with its last row (a valid address)
        o.hcol.resize(static_cast<size_t>(np) * P);
        for (size_t h = 0; h < o.hcol.size(); ++h) {
            const uint32_t c = hot[std::min(h, hot.size() - 1)].col;
            o.hcol[h] = (col_map ? static_cast<uint32_t>(col_map[c]) : c) * row_bytes32;
        }
        // ---- the hot entries: (part, panel) -> records, at most run_max of them (the rest of such a run stays with the flat kernel);
        // a row's hot nonzeros of one panel are dealt round-robin over its parts
        std::vector<uint32_t> cnt_ip(static_cast<size_t>(n_sl) * np, 0u);  // [part][panel]
        std::vector<uint32_t> ent_panel(ent.size(), 0xFFFFFFFFu), ent_off(ent.size(), 0u), ent_part(ent.size(), 0u);
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
                    ent_panel[e] = idx / P;
                    ent_off[e] = (idx % P) * kBkRowBytes;
                }
            }
            // in row order (ent is in (item, position in the row) order), so that WHICH entries of an over-long run stay hot is the
            // same whatever the sort above did with equal keys
            std::vector<uint32_t> rr(np, 0u);
            uint32_t cur_item = 0xFFFFFFFFu;
            for (size_t e = 0; e < ent.size(); ++e) {
                if (ent[e].item != cur_item) {
                    cur_item = ent[e].item;
                    std::fill(rr.begin(), rr.end(), 0u);
                }
                if (ent_panel[e] == 0xFFFFFFFFu) continue;
                const uint32_t v = it[cur_item].v, part = first_slot[cur_item] + (rr[ent_panel[e]]++ % v);
                uint32_t &c = cnt_ip[static_cast<size_t>(part) * np + ent_panel[e]];
                if (c >= run_max) {
                    ent_panel[e] = 0xFFFFFFFFu;
                    ++o.lost_run;
                } else {
                    ++c;
                    ent_part[e] = part;
                }
            }
        }
        // ---- slots: parts with similar panel profiles side by side (a run is as long as the longest of its 4 slots): by the
        // panel that holds most of the part's hot nonzeros, then by how many it has there, then by the total
        std::vector<uint32_t> tot(n_sl, 0u), top(n_sl, 0u), top_cnt(n_sl, 0u);
        for (uint32_t x = 0; x < n_sl; ++x)
            for (uint32_t ph = 0; ph < np; ++ph) {
                const uint32_t c = cnt_ip[static_cast<size_t>(x) * np + ph];
                tot[x] += c;
                if (c > top_cnt[x]) top_cnt[x] = c, top[x] = ph;
            }
        std::vector<uint32_t> ord(n_sl);
        std::iota(ord.begin(), ord.end(), 0u);
        std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t c) {
            if ((tot[a] == 0) != (tot[c] == 0)) return tot[a] != 0;  // parts without a hot nonzero last: their slots stay empty
            if (top[a] != top[c]) return top[a] < top[c];
            if (top_cnt[a] != top_cnt[c]) return top_cnt[a] > top_cnt[c];
            return tot[a] > tot[c];
        });
        std::vector<uint32_t> slot_of(n_sl);
        for (uint32_t s = 0; s < n_sl; ++s) slot_of[ord[s]] = s;
        const uint32_t n_groups = rounds * kBkWaves;  // RB / 4
        auto cnt_sp = [&](uint32_t s, uint32_t ph) -> uint32_t { return s < n_sl ? cnt_ip[static_cast<size_t>(ord[s]) * np + ph] : 0u; };
        // records of every (slot, panel), in row order
        std::vector<uint32_t> beg_sp(static_cast<size_t>(RB) * np + 1, 0u);
        for (uint32_t s = 0; s < RB; ++s)
            for (uint32_t ph = 0; ph < np; ++ph) beg_sp[static_cast<size_t>(s) * np + ph + 1] = beg_sp[static_cast<size_t>(s) * np + ph] + cnt_sp(s, ph);
        std::vector<uint2> rec_sp(beg_sp.back());
        {
            std::vector<uint32_t> fill(beg_sp.begin(), beg_sp.end() - 1);
            for (size_t e = 0; e < ent.size(); ++e) {
                if (ent_panel[e] == 0xFFFFFFFFu) continue;
                uint32_t bits;
                std::memcpy(&bits, &ent[e].val, 4);
                rec_sp[fill[static_cast<size_t>(slot_of[ent_part[e]]) * np + ent_panel[e]]++] = make_uint2(ent_off[e], bits);
                hot_mask[ent[e].e] = 1;
                ++o.hot_nnz;
            }
        }
        // ---- groups of 4 slots -> (wave, round): longest processing time first
        std::vector<uint64_t> g_cost(n_groups, 0);
        for (uint32_t g = 0; g < n_groups; ++g)
            for (uint32_t ph = 0; ph < np; ++ph) {
                uint32_t mx = 0;
                for (uint32_t s = 0; s < kBkSlots; ++s) mx = std::max(mx, cnt_sp(g * kBkSlots + s, ph));
                g_cost[g] += mx;
            }
        std::vector<uint32_t> g_ord(n_groups);
        std::iota(g_ord.begin(), g_ord.end(), 0u);
        std::stable_sort(g_ord.begin(), g_ord.end(), [&](uint32_t a, uint32_t c) { return g_cost[a] > g_cost[c]; });
        std::vector<uint64_t> w_load(kBkWaves, 0);
        std::vector<uint32_t> w_n(kBkWaves, 0), grp_of(static_cast<size_t>(kBkWaves) * rounds, 0u);  // [wave][round]
        for (uint32_t g : g_ord) {
            uint32_t best = kBkWaves;
            for (uint32_t w = 0; w < kBkWaves; ++w)
                if (w_n[w] < rounds && (best == kBkWaves || w_load[w] < w_load[best])) best = w;
            grp_of[static_cast<size_t>(best) * rounds + w_n[best]++] = g;
            w_load[best] += g_cost[g];
        }
        // ---- emit
        // where every sorted slot ended up in the kernel's [round][wave][slot] numbering (the index of brow / link, and of the
        // slot's 256 bytes of LDS when the parts of a row meet)
        std::vector<uint32_t> place(RB, 0u);
        for (uint32_t w = 0; w < kBkWaves; ++w)
            for (uint32_t rd = 0; rd < rounds; ++rd) {
                const uint32_t g = grp_of[static_cast<size_t>(w) * rounds + rd];
                for (uint32_t s = 0; s < kBkSlots; ++s) place[g * kBkSlots + s] = (rd * kBkWaves + w) * kBkSlots + s;
            }
        // brow: the C row of an OWNER slot (part 0 of a row that has a hot nonzero anywhere), kBkEmptyRow for every other slot.
        // link: owner of a multi-part row: kBkLinkOwner | place + 1 of its next part with a hot nonzero; such a part: kBkLinkPart |
        // place + 1 of the next one (0 = the last).  Parts without a hot nonzero are left out of the chain.
        o.link.assign(RB, 0u);
        for (uint32_t x = 0; x < n_it; ++x) {
            uint32_t any = 0;
            for (uint32_t q = 0; q < it[x].v; ++q) any += tot[first_slot[x] + q];
            if (any == 0) continue;
            const uint32_t r = sched[it[x].spos];
            const uint32_t own = place[slot_of[first_slot[x]]];
            o.brow[own] = dst_map ? static_cast<uint32_t>(dst_map[r]) : r - static_cast<uint32_t>(r0);
            uint32_t prev = own, n_chain = 0;
            for (uint32_t q = 1; q < it[x].v; ++q) {
                if (tot[first_slot[x] + q] == 0) continue;
                const uint32_t pl = place[slot_of[first_slot[x] + q]];
                o.link[prev] |= pl + 1;
                o.link[pl] = kBkLinkPart;
                prev = pl;
                ++n_chain;
            }
            if (n_chain) o.link[own] |= kBkLinkOwner;
        }
        const uint32_t cw = 2 * np;  // per wave and panel one 64-bit word: byte r = the steps of the run (panel, round r)
        o.cnt.assign(static_cast<size_t>(kBkWaves) * cw, 0u);
        uint32_t step_pos = 0;
        for (uint32_t w = 0; w < kBkWaves; ++w) {
            const uint32_t w_first = step_pos;
            for (uint32_t ph = 0; ph < np; ++ph)
                for (uint32_t rd = 0; rd < rounds; ++rd) {
                    const uint32_t g = grp_of[static_cast<size_t>(w) * rounds + rd];
                    uint32_t steps = 0;
                    for (uint32_t s = 0; s < kBkSlots; ++s) steps = std::max(steps, cnt_sp(g * kBkSlots + s, ph));
                    o.cnt[static_cast<size_t>(w) * cw + 2 * ph + rd / 4] |= steps << (8 * (rd & 3));
                    const size_t base = o.rec.size();
                    o.rec.resize(base + static_cast<size_t>(steps) * kBkSlots);
                    for (uint32_t s = 0; s < kBkSlots; ++s) {
                        const uint32_t sl = g * kBkSlots + s, have = cnt_sp(sl, ph);
                        const uint32_t first = have ? beg_sp[static_cast<size_t>(sl) * np + ph] : 0u;
                        // a slot shorter than its run is padded with value 0 at the buffer's row of zeros: a non-finite B value
                        // reaches only the rows that reference it
                        for (uint32_t q = 0; q < steps; ++q) o.rec[base + static_cast<size_t>(q) * kBkSlots + s] = q < have ? rec_sp[first + q] : make_uint2(kBkZeroRow, 0u);
                    }
                    step_pos += steps;
                }
            o.wstart[w] = make_uint2(w_first, step_pos - w_first);
        }
        bool chains = false;
        for (uint32_t l : o.link) chains = chains || l != 0;
        o.multi = chains;
        o.hdr = make_uint4(np | (chains ? 0x80000000u : 0u), 0u, 0u, cw);
    });

    // ---- concatenate (offsets are sequential; the copies run in parallel)
    std::vector<uint64_t> rec_at(static_cast<size_t>(nb) + 1, 0), cnt_at(static_cast<size_t>(nb) + 1, 0), hcol_at(static_cast<size_t>(nb) + 1, 0);
    for (int64_t b = 0; b < nb; ++b) {
        const BlockOut &o = out[static_cast<size_t>(b)];
        if ((o.hdr.x & 0x7FFFFFFFu) > max_panels) return FLEX_ERR_UNSUPPORTED;  // cannot happen: max_panels bounds both
        rec_at[b + 1] = rec_at[b] + o.rec.size();
        cnt_at[b + 1] = cnt_at[b] + o.cnt.size();
        hcol_at[b + 1] = hcol_at[b] + o.hcol.size();
        img.nnz += o.nnz;
        img.hot_nnz += o.hot_nnz;
        img.hot_cols += o.hot_cols;
        img.panels += o.hdr.x & 0x7FFFFFFFu;
        img.rows += o.rows;
        img.cand_nnz += o.cand_nnz;
        img.lost_panels += o.lost_panels;
        img.lost_last += o.lost_last;
        img.lost_run += o.lost_run;
    }
    if (rec_at[nb] / kBkSlots >= (uint64_t(1) << 32) || cnt_at[nb] >= (uint64_t(1) << 32) || hcol_at[nb] >= (uint64_t(1) << 32)) return FLEX_ERR_UNSUPPORTED;
    img.n_blocks = static_cast<uint32_t>(nb);
    img.hdr.resize(static_cast<size_t>(nb));
    img.wstart.resize(static_cast<size_t>(nb) * kBkWaves);
    img.brow.resize(static_cast<size_t>(nb) * RB);
    img.link.resize(static_cast<size_t>(nb) * RB);
    img.cnt.resize(static_cast<size_t>(cnt_at[nb]));
    img.hcol.resize(static_cast<size_t>(hcol_at[nb]));
    img.rec.resize(static_cast<size_t>(rec_at[nb]));
    parallel_chunks(nb, [&](int64_t b) {
        BlockOut &o = out[static_cast<size_t>(b)];
        img.hdr[b] = make_uint4(o.hdr.x, static_cast<uint32_t>(hcol_at[b]), static_cast<uint32_t>(cnt_at[b]), o.hdr.w);
        const uint32_t step0 = static_cast<uint32_t>(rec_at[b] / kBkSlots);
        for (uint32_t w = 0; w < kBkWaves; ++w) img.wstart[static_cast<size_t>(b) * kBkWaves + w] = make_uint2(step0 + o.wstart[w].x, o.wstart[w].y);
        std::copy(o.brow.begin(), o.brow.end(), img.brow.begin() + static_cast<size_t>(b) * RB);
        std::copy(o.link.begin(), o.link.end(), img.link.begin() + static_cast<size_t>(b) * RB);
        std::copy(o.cnt.begin(), o.cnt.end(), img.cnt.begin() + cnt_at[b]);
        std::copy(o.hcol.begin(), o.hcol.end(), img.hcol.begin() + hcol_at[b]);
        std::copy(o.rec.begin(), o.rec.end(), img.rec.begin() + rec_at[b]);
        o = BlockOut{};  // free as we go
    });
    return FLEX_OK;
}

}  // namespace flex
