// reorder.cpp -- vertex reordering on the host (RCM) and "apply a permutation to a CSR".
//
// ≙ DataLoaderRcm (DataLoader.cu:723-787) -> order_rcm (order_rcm.cu:15-33) ->
// order_deg (order_deg.cu:19-45), Dadjlist (adjlist.cu:127-150), algo_bfs
// (algo_bfs.cu:11-39), rank_from_order (tools.cu:31-43).  Produces the same rank
// as the reference (ties broken the same way) without its 16-byte-per-edge
// edge list, std::function ranker or per-row comparison sorts: degrees come
// straight from the CSR, the degree order is a counting sort, and neighbour
// lists are built already sorted by scattering in relabelled order.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <numeric>
#include <stdexcept>
#include <vector>

#include "host_parallel.h"
#include "internal.h"

namespace flex {

int order_rcm_host(int64_t n, const uint32_t *rowPtr, const uint32_t *col, std::vector<uint32_t> &rank) {
    rank.assign(static_cast<size_t>(n), 0u);
    if (n == 0) return FLEX_OK;
    const uint32_t nnz = rowPtr[n];
    // deg = out-degree + in-degree (edgelist.cu:86-103)
    std::vector<uint32_t> deg(static_cast<size_t>(n), 0u);
    for (int64_t u = 0; u < n; ++u) deg[u] = rowPtr[u + 1] - rowPtr[u];
    for (uint32_t e = 0; e < nnz; ++e) {
        if (col[e] >= n) return FLEX_ERR_INVALID;
        ++deg[col[e]];
    }
    // rank_deg: position in (degree ASC, id ASC) order == stable counting sort by degree
    const uint32_t maxdeg = *std::max_element(deg.begin(), deg.end());
    std::vector<uint32_t> bucket(static_cast<size_t>(maxdeg) + 2, 0u);
    for (int64_t u = 0; u < n; ++u) ++bucket[deg[u] + 1];
    for (size_t d = 1; d < bucket.size(); ++d) bucket[d] += bucket[d - 1];
    std::vector<uint32_t> rank_deg(static_cast<size_t>(n)), by_deg(static_cast<size_t>(n));
    for (int64_t u = 0; u < n; ++u) {
        const uint32_t pos = bucket[deg[u]]++;
        rank_deg[u] = pos;
        by_deg[pos] = static_cast<uint32_t>(u);
    }
    // out-adjacency in relabelled ids with ascending neighbours.  Scatter targets in
    // ascending relabelled id: visiting v' = 0..n-1 and appending v' to every source
    // u' with an edge u -> v would need the transpose; instead build lists then sort
    // only when a list is out of order (short lists dominate).
    std::vector<uint32_t> cd(static_cast<size_t>(n) + 1, 0u);
    for (int64_t u = 0; u < n; ++u) cd[rank_deg[u] + 1] = rowPtr[u + 1] - rowPtr[u];
    for (int64_t u = 0; u < n; ++u) cd[u + 1] += cd[u];
    std::vector<uint32_t> adj(nnz);
    for (int64_t u = 0; u < n; ++u) {
        uint32_t *o = adj.data() + cd[rank_deg[u]];
        const uint32_t len = rowPtr[u + 1] - rowPtr[u];
        for (uint32_t i = 0; i < len; ++i) o[i] = rank_deg[col[rowPtr[u] + i]];
        if (!std::is_sorted(o, o + len)) std::sort(o, o + len);
    }
    // BFS over all components, roots in relabelled-id order, from node 0 (algo_bfs.cu:21-36)
    std::vector<uint32_t> order;
    order.reserve(static_cast<size_t>(n));
    std::vector<uint8_t> placed(static_cast<size_t>(n), 0);
    size_t head = 0;
    for (int64_t c = 0; c < n; ++c) {
        if (placed[c]) continue;
        placed[c] = 1;
        order.push_back(static_cast<uint32_t>(c));
        while (head < order.size()) {
            const uint32_t w = order[head++];
            for (uint32_t a = cd[w]; a < cd[w + 1]; ++a) {
                const uint32_t v = adj[a];
                if (!placed[v]) {
                    placed[v] = 1;
                    order.push_back(v);
                }
            }
        }
    }
    // rank[u] = n-1 - rank_bfs[rank_deg[u]]   (order_rcm.cu:28-31)
    std::vector<uint32_t> &rank_bfs = deg;  // reuse
    for (int64_t i = 0; i < n; ++i) rank_bfs[order[i]] = static_cast<uint32_t>(i);
    for (int64_t u = 0; u < n; ++u) rank[u] = static_cast<uint32_t>(n - 1 - rank_bfs[rank_deg[u]]);
    return FLEX_OK;
}

}  // namespace flex

extern "C" {

int flex_order_rcm(const flex_csr *A, uint32_t *rank) try {
    if (!rank) return FLEX_ERR_INVALID;
    int rc = flex::validate_csr(A);
    if (rc) return rc;
    if (A->m != A->n) return FLEX_ERR_INVALID;
    std::vector<uint32_t> r;
    rc = flex::order_rcm_host(A->m, A->rowPtr, A->col, r);
    if (rc) return rc;
    std::copy(r.begin(), r.end(), rank);
    return FLEX_OK;
} catch (const std::bad_alloc &) {
    return FLEX_ERR_NOMEM;
} catch (const std::length_error &) {
    return FLEX_ERR_NOMEM;
} catch (...) {  // nothing crosses the C ABI as an exception
    return FLEX_ERR_INVALID;
}

int flex_order_dfs(const flex_csr *A, uint32_t *rank) try {
    // ≙ DataLoaderDFS (DataLoader.cu:324-395): pre-order numbering of an iterative depth-first search
    // that starts at vertex 0, follows out-edges in CSR order and restarts at the lowest unvisited vertex.
    if (!rank) return FLEX_ERR_INVALID;
    int rc = flex::validate_csr(A);
    if (rc) return rc;
    if (A->m != A->n) return FLEX_ERR_INVALID;
    const uint32_t n = static_cast<uint32_t>(A->m), unseen = 0xFFFFFFFFu;
    std::fill(rank, rank + n, unseen);
    std::vector<std::pair<uint32_t, uint32_t>> stack;  // (vertex, next edge)
    uint32_t next_id = 0;
    for (uint32_t root = 0; root < n; ++root) {
        if (rank[root] != unseen) continue;
        rank[root] = next_id++;
        stack.emplace_back(root, A->rowPtr[root]);
        while (!stack.empty()) {
            auto &top = stack.back();
            if (top.second == A->rowPtr[top.first + 1]) {
                stack.pop_back();
                continue;
            }
            const uint32_t v = A->col[top.second++];
            if (rank[v] != unseen) continue;
            rank[v] = next_id++;
            stack.emplace_back(v, A->rowPtr[v]);
        }
    }
    return FLEX_OK;
} catch (const std::bad_alloc &) {
    return FLEX_ERR_NOMEM;
} catch (const std::length_error &) {
    return FLEX_ERR_NOMEM;
} catch (...) {  // nothing crosses the C ABI as an exception
    return FLEX_ERR_INVALID;
}

int flex_order_deg(const flex_csr *A, int descending, uint32_t *rank) try {
    if (!rank) return FLEX_ERR_INVALID;
    int rc = flex::validate_csr(A);
    if (rc) return rc;
    if (A->m != A->n) return FLEX_ERR_INVALID;
    const int32_t n = A->m;
    std::vector<uint32_t> deg(static_cast<size_t>(n), 0u), by(static_cast<size_t>(n));
    for (int32_t u = 0; u < n; ++u) deg[u] = A->rowPtr[u + 1] - A->rowPtr[u];
    for (int64_t e = 0; e < A->nnz; ++e) ++deg[A->col[e]];
    std::iota(by.begin(), by.end(), 0u);
    std::stable_sort(by.begin(), by.end(), [&](uint32_t a, uint32_t b) {
        return descending ? deg[a] > deg[b] : deg[a] < deg[b];
    });
    for (int32_t i = 0; i < n; ++i) rank[by[i]] = static_cast<uint32_t>(i);
    return FLEX_OK;
} catch (const std::bad_alloc &) {
    return FLEX_ERR_NOMEM;
} catch (const std::length_error &) {
    return FLEX_ERR_NOMEM;
} catch (...) {  // nothing crosses the C ABI as an exception
    return FLEX_ERR_INVALID;
}

int flex_perm_csr(const flex_csr *A, const uint32_t *rank, int32_t *vo_mp, uint32_t *rowPtr2, uint32_t *col2,
                  float *vals2) try {
    if (!rank || !vo_mp || !rowPtr2 || (A && A->nnz > 0 && (!col2 || !vals2))) return FLEX_ERR_INVALID;
    int rc = flex::validate_csr(A);
    if (rc) return rc;
    if (A->m != A->n) return FLEX_ERR_INVALID;
    const int32_t n = A->m;
    std::vector<uint8_t> seen(static_cast<size_t>(n), 0);
    for (int32_t u = 0; u < n; ++u) {
        if (rank[u] >= static_cast<uint32_t>(n) || seen[rank[u]]) return FLEX_ERR_INVALID;  // not a permutation
        seen[rank[u]] = 1;
        vo_mp[rank[u]] = u;  // DataLoader.cu:747-750
    }
    rowPtr2[0] = 0;
    for (int32_t i = 0; i < n; ++i) rowPtr2[i + 1] = rowPtr2[i] + (A->rowPtr[vo_mp[i] + 1] - A->rowPtr[vo_mp[i]]);
    // DataLoader.cu:758-779: map columns, sort ascending.  Rows are independent (each owns its output range): blocks of rows
    // in parallel -- the amazon shape spent 6.5 s here on one thread, every rank of a row-sharded run pays it.
    constexpr int64_t kBlk = 4096;
    std::atomic<int> failed{0};
    flex::parallel_chunks((static_cast<int64_t>(n) + kBlk - 1) / kBlk, [&](int64_t b) {
        try {
            std::vector<std::pair<uint32_t, float>> row;
            for (int64_t s = b * kBlk; s < std::min<int64_t>(n, (b + 1) * kBlk); ++s) {
                row.clear();
                for (uint32_t e = A->rowPtr[s]; e < A->rowPtr[s + 1]; ++e) row.emplace_back(rank[A->col[e]], A->vals[e]);
                std::sort(row.begin(), row.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
                uint32_t o = rowPtr2[rank[s]];
                for (const auto &cv : row) {
                    col2[o] = cv.first;
                    vals2[o++] = cv.second;
                }
            }
        } catch (...) {
            failed.store(1);
        }
    });
    return failed.load() ? FLEX_ERR_NOMEM : FLEX_OK;
} catch (const std::bad_alloc &) {
    return FLEX_ERR_NOMEM;
} catch (const std::length_error &) {
    return FLEX_ERR_NOMEM;
} catch (...) {  // nothing crosses the C ABI as an exception
    return FLEX_ERR_INVALID;
}

}  // extern "C"
