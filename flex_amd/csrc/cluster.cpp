// cluster.cpp -- community ("cluster") vertex ordering on the host.
//
// Counterpart of DataLoaderRabbit (DataLoader.cu:453-655): agglomerative,
// modularity-driven clustering in rounds (lowest-degree community first, merge into
// the neighbour with the largest modularity gain while the gain is positive), then a
// depth-first walk of the merge forest so that vertices of one community -- and of
// its sub-communities -- become consecutive.  Own design, not a restatement: the
// reference keeps a std::map per vertex and rewrites both endpoints' maps on every
// merge; here a community's adjacency is an append-only list that is compacted
// lazily through a union-find when the community is next examined, and weights are
// accumulated in one dense scratch array (memory ~ 8 B per nonzero, no per-edge
// allocation), which is what makes Amazon-size inputs (2.6e8 nnz) practical.
//
// In this engine an ordering is a SCHEDULE (which rows a wave / an XCD works on
// next), never a data permutation: see plan.cpp.
#include <algorithm>
#include <cstring>
#include <numeric>
#include <stdexcept>
#include <vector>

#include "internal.h"

namespace flex {

namespace {

struct Clusterer {
    const int64_t n;
    const uint32_t *rowPtr, *col;
    std::vector<uint32_t> parent;                                 // union-find over communities
    std::vector<double> cdeg;                                     // community degree (sum of member degrees)
    std::vector<std::vector<std::pair<uint32_t, float>>> adj;     // lazily compacted community adjacency
    std::vector<uint8_t> raw;                                     // adjacency still lives in the CSR
    std::vector<std::vector<uint32_t>> children;                  // merge forest
    std::vector<float> acc;                                       // dense scratch: weight per neighbour root
    std::vector<uint32_t> touched;
    double M = 0;                                                 // total degree (2m)

    Clusterer(int64_t n_, const uint32_t *rp, const uint32_t *c)
        : n(n_), rowPtr(rp), col(c), parent(n_), cdeg(n_, 0.0), adj(n_), raw(n_, 1), children(n_), acc(n_, 0.f) {
        std::iota(parent.begin(), parent.end(), 0u);
        for (int64_t u = 0; u < n; ++u) {
            uint32_t d = 0;
            for (uint32_t e = rowPtr[u]; e < rowPtr[u + 1]; ++e) d += (col[e] != u);
            cdeg[u] = d;
            M += d;
        }
    }

    uint32_t find(uint32_t x) {
        while (parent[x] != x) {
            parent[x] = parent[parent[x]];
            x = parent[x];
        }
        return x;
    }

    // Rebuild u's adjacency keyed by CURRENT roots; returns the best merge target or u itself.
    uint32_t compact_and_pick(uint32_t u) {
        touched.clear();
        auto add = [&](uint32_t v, float w) {
            const uint32_t r = find(v);
            if (r == u) return;
            if (acc[r] == 0.f) touched.push_back(r);
            acc[r] += w;
        };
        if (raw[u]) {
            for (uint32_t e = rowPtr[u]; e < rowPtr[u + 1]; ++e) add(col[e], 1.f);
            raw[u] = 0;
        }
        for (const auto &vw : adj[u]) add(vw.first, vw.second);
        auto &list = adj[u];
        list.clear();
        list.reserve(touched.size());
        uint32_t best = u;
        double best_gain = 0.0;
        const double du_over_M = cdeg[u] / M;
        for (uint32_t r : touched) {
            const float w = acc[r];
            acc[r] = 0.f;
            list.emplace_back(r, w);
            const double gain = static_cast<double>(w) - cdeg[r] * du_over_M;  // ~ delta modularity * M
            // ties go to the smaller community id, whatever order the neighbours were met in
            if (gain > best_gain || (gain == best_gain && gain > 0.0 && r < best)) {
                best_gain = gain;
                best = r;
            }
        }
        return best;
    }

    void run(int max_rounds) {
        if (M <= 0) return;
        std::vector<uint32_t> cur(static_cast<size_t>(n)), next;
        std::iota(cur.begin(), cur.end(), 0u);
        std::vector<uint32_t> stamp(static_cast<size_t>(n), 0u);
        for (int round = 1; round <= max_rounds && !cur.empty(); ++round) {
            std::stable_sort(cur.begin(), cur.end(), [&](uint32_t a, uint32_t b) { return cdeg[a] < cdeg[b]; });
            next.clear();
            for (uint32_t u : cur) {
                if (parent[u] != u) continue;        // merged earlier in this round
                if (stamp[u] == static_cast<uint32_t>(round)) continue;  // just absorbed something: next round
                const uint32_t v = compact_and_pick(u);
                if (v == u) continue;
                parent[u] = v;  // u joins v
                cdeg[v] += cdeg[u];
                children[v].push_back(u);
                if (raw[v]) {  // materialise v's own edges before appending foreign ones
                    for (uint32_t e = rowPtr[v]; e < rowPtr[v + 1]; ++e)
                        if (col[e] != v) adj[v].emplace_back(col[e], 1.f);
                    raw[v] = 0;
                }
                adj[v].insert(adj[v].end(), adj[u].begin(), adj[u].end());
                std::vector<std::pair<uint32_t, float>>().swap(adj[u]);
                if (stamp[v] != static_cast<uint32_t>(round)) {
                    stamp[v] = static_cast<uint32_t>(round);
                    next.push_back(v);
                }
            }
            cur.swap(next);
        }
    }

    // depth-first over the merge forest: a community's members become consecutive
    void order(std::vector<uint32_t> &rank) {
        rank.assign(static_cast<size_t>(n), 0u);
        std::vector<uint32_t> roots;
        for (int64_t v = 0; v < n; ++v)
            if (parent[v] == v) roots.push_back(static_cast<uint32_t>(v));
        // large communities first, ties by id: deterministic, and keeps the long tail of
        // singletons (isolated / unmerged vertices) together at the end
        std::stable_sort(roots.begin(), roots.end(), [&](uint32_t a, uint32_t b) { return cdeg[a] > cdeg[b]; });
        uint32_t next_id = 0;
        std::vector<std::pair<uint32_t, uint32_t>> stack;  // (vertex, next child index)
        for (uint32_t r : roots) {
            stack.emplace_back(r, 0u);
            rank[r] = next_id++;
            while (!stack.empty()) {
                auto &top = stack.back();
                if (top.second < children[top.first].size()) {
                    // most recently merged sub-community first: it is the one most tied to the parent
                    const auto &ch = children[top.first];
                    const uint32_t c = ch[ch.size() - 1 - top.second++];
                    rank[c] = next_id++;
                    stack.emplace_back(c, 0u);
                } else {
                    stack.pop_back();
                }
            }
        }
    }
};

}  // namespace

int order_cluster_host(int64_t n, const uint32_t *rowPtr, const uint32_t *col, std::vector<uint32_t> &rank) {
    if (n == 0) {
        rank.clear();
        return FLEX_OK;
    }
    for (uint32_t e = 0; e < rowPtr[n]; ++e)
        if (col[e] >= n) return FLEX_ERR_INVALID;
    try {
        Clusterer c(n, rowPtr, col);
        c.run(32);
        c.order(rank);
    } catch (const std::bad_alloc &) {
        return FLEX_ERR_NOMEM;
    }
    return FLEX_OK;
}

}  // namespace flex

extern "C" int flex_order_cluster(const flex_csr *A, uint32_t *rank) try {
    if (!rank) return FLEX_ERR_INVALID;
    int rc = flex::validate_csr(A);
    if (rc) return rc;
    if (A->m != A->n) return FLEX_ERR_INVALID;
    std::vector<uint32_t> r;
    rc = flex::order_cluster_host(A->m, A->rowPtr, A->col, r);
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
