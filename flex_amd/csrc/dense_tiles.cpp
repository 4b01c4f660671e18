// dense_tiles.cpp -- block-density detector and dense-tile store of the MFMA route (north_star: "MFMA only where
// RCM/Gorder reordering yields dense block-sparse tiles").
//
// The matrix is looked at in SCHEDULE coordinates: row tile = 32 consecutive rows of the schedule, column tile = 32
// consecutive column positions.  Every (row tile, column tile) pair with at least one entry is counted; `hist_nnz`
// reports which share of the nonzeros sits in tiles of fill >= 0.10 / 0.25 / 0.50 (the verdict `flex ... --stats`
// prints for every graph), and -- when `thr` > 0 -- tiles holding >= thr entries are taken OUT of the record stream
// (`in_tile[e - e_base] = 1`) and stored as dense fp32 blocks in the A-operand order of v_mfma_f32_32x32x2_f32.
#include <algorithm>
#include <atomic>
#include <new>

#include "host_parallel.h"
#include "plan.h"

namespace flex {

int detect_dense_tiles(const flex_csr *A, int32_t r0, int32_t m, const std::vector<uint32_t> &sched, const std::vector<uint32_t> &colpos,
                       const int32_t *col_map, const int32_t *dst_map, bool off32, uint32_t row_bytes32, uint32_t thr, int64_t stride,
                       std::vector<uint8_t> &in_tile, DenseTiles &out) {
    const uint32_t e_base = A->rowPtr[r0];
    const int64_t n_rt = (static_cast<int64_t>(m) + 31) / 32;
    constexpr int64_t kBlk = 64;  // row tiles per work item
    const int64_t nblk = (n_rt + kBlk - 1) / kBlk;
    struct Found {
        uint32_t rt, ct;
        std::vector<float> a;  // 1024, operand order
        uint32_t mask[32];     // bit j of word i: an entry at (row i, column j)
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
                        Found f{static_cast<uint32_t>(rt), static_cast<uint32_t>(key[z] >> 32), std::vector<float>(1024, 0.f), {}};
                        uint8_t taken[32][32] = {};
                        for (size_t y = z; y < z1; ++y) {
                            const int i = static_cast<int>((key[y] >> 27) & 31);
                            const uint32_t e = ebeg[i] + static_cast<uint32_t>(key[y] & ((1u << 27) - 1));
                            const uint32_t c = A->col[e];
                            const int j = static_cast<int>((colpos.empty() ? c : colpos[c]) & 31);
                            if (taken[i][j]) continue;  // a duplicate (row, col) entry stays with the vector kernel
                            taken[i][j] = 1;
                            f.mask[i] |= 1u << j;
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
                out.mask.insert(out.mask.end(), f.mask, f.mask + 32);
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

}  // namespace flex
