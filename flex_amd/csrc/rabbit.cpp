// rabbit.cpp -- flex_order_rabbit: the reference's Rabbit vertex order (DataLoaderRabbit, DataLoader.cu:455-655) with the
// options it is compiled with (iterative rounds in degree order, no hub grouping, cluster shyness 1), so that the
// reference-flow loader "RBT" of the host mirror orders a graph the way the reference does: same merges, same dendrogram
// walk, hence the same vo_mp.  (The engine's own schedule, FLEX_ORDER_CLUSTER / flex_order_cluster in cluster.cpp, is a
// different, parallel algorithm built for Amazon-size inputs; this one keeps the reference's per-vertex weight maps and is
// meant for the graph sizes the reference runs it on.)
//
// Data structure: where the reference keeps a std::map<int,int> per vertex, a vertex here owns one sorted vector of
// (neighbour, weight) -- the same ascending iteration order, one allocation per vertex instead of one per edge.
// One liberty, shared with the oracle: a round's vertices are sorted by degree with a STABLE sort (the reference's
// ranges::sort leaves the order of equal degrees to the library).
#include <algorithm>
#include <cstdint>
#include <vector>

#include "internal.h"

namespace {

struct Nbr {
    uint32_t v;
    int32_t w;
};
using NbrList = std::vector<Nbr>;

inline NbrList::iterator seek(NbrList &l, uint32_t v) {
    return std::lower_bound(l.begin(), l.end(), v, [](const Nbr &a, uint32_t key) { return a.v < key; });
}
inline void add_weight(NbrList &l, uint32_t v, int32_t w) {
    auto it = seek(l, v);
    if (it != l.end() && it->v == v) it->w += w;
    else l.insert(it, Nbr{v, w});
}

}  // namespace

extern "C" int flex_order_rabbit(const flex_csr *A, int is_directed, uint32_t *rank) try {
    if (!rank) return FLEX_ERR_INVALID;
    int rc = flex::validate_csr(A);
    if (rc) return rc;
    if (A->m != A->n) return FLEX_ERR_INVALID;
    const uint32_t n = static_cast<uint32_t>(A->m);
    if (n == 0) return FLEX_OK;
    std::vector<NbrList> g(n);
    std::vector<int64_t> deg(n);
    int64_t n_edges = 0;
    for (uint32_t v = 0; v < n; ++v) {
        for (uint32_t e = A->rowPtr[v]; e < A->rowPtr[v + 1]; ++e) {
            const uint32_t d = A->col[e];
            if (d == v) continue;
            auto it = seek(g[v], d);
            if (it == g[v].end() || it->v != d) g[v].insert(it, Nbr{d, 1});
            if (is_directed) {  // cluster on the undirected version of the graph (DataLoader.cu:516, 527)
                auto jt = seek(g[d], v);
                if (jt == g[d].end() || jt->v != v) g[d].insert(jt, Nbr{v, 1});
            }
        }
        // the reference takes a vertex's degree at the END OF ITS OWN TURN here (DataLoader.cu:529-530): its out-edges plus the
        // reverse edges inserted by the vertices before it; reverse edges that later vertices add reach the map, not deg / n_edges
        n_edges += (deg[v] = static_cast<int64_t>(g[v].size()));
    }
    const double two_m_inv = 1.0 / static_cast<double>(2 * n_edges);
    // dendrogram: ids < n are leaves; n + u is the node created when u was merged away: (tree of its target, tree of u)
    std::vector<int64_t> left(2 * static_cast<size_t>(n), -1), right(2 * static_cast<size_t>(n), -1), tree(n);
    std::vector<int32_t> last_round(n, 0);
    std::vector<uint32_t> active(n), upcoming;
    for (uint32_t v = 0; v < n; ++v) tree[v] = active[v] = v;
    for (int32_t round = 1; !active.empty(); ++round) {
        std::stable_sort(active.begin(), active.end(), [&](uint32_t a, uint32_t b) { return deg[a] < deg[b]; });
        upcoming.clear();
        for (const uint32_t u : active) {
            if (last_round[u] == round) continue;  // it absorbed a vertex in this round
            double best = -1.0;
            int64_t v = -1;
            const double du_2m = static_cast<double>(deg[u]) * two_m_inv;
            for (const Nbr &x : g[u]) {  // ascending neighbour id; strictly larger gain wins
                const double gain = static_cast<double>(x.w) - static_cast<double>(deg[x.v]) * du_2m;
                if (gain > best) best = gain, v = x.v;
            }
            if (best <= 0.0) continue;
            deg[v] += deg[u];
            for (const Nbr &x : g[u]) {
                if (x.v == static_cast<uint32_t>(v)) continue;
                add_weight(g[v], x.v, x.w);
                NbrList &other = g[x.v];
                auto it = seek(other, u);
                if (it == other.end() || it->v != u) continue;
                const int32_t w_u = it->w;
                other.erase(it);
                add_weight(other, static_cast<uint32_t>(v), w_u);
            }
            auto it = seek(g[v], u);
            if (it != g[v].end() && it->v == u) g[v].erase(it);
            left[n + u] = tree[v];
            right[n + u] = tree[u];
            tree[u] = -1;
            tree[v] = static_cast<int64_t>(n) + u;
            if (last_round[v] == round) continue;
            last_round[v] = round;
            upcoming.push_back(static_cast<uint32_t>(v));
        }
        active.swap(upcoming);
    }
    uint32_t next_id = 0;
    std::vector<int64_t> todo;
    for (uint32_t v = 0; v < n; ++v) {
        if (tree[v] < 0) continue;
        todo.assign(1, tree[v]);
        while (!todo.empty()) {  // leaves of the dendrogram, left to right
            const int64_t t = todo.back();
            todo.pop_back();
            if (t < static_cast<int64_t>(n)) {
                rank[t] = next_id++;
                continue;
            }
            todo.push_back(right[t]);
            todo.push_back(left[t]);
        }
    }
    return FLEX_OK;
} catch (const std::bad_alloc &) {
    return FLEX_ERR_NOMEM;
} catch (...) {  // nothing crosses the C ABI as an exception
    return FLEX_ERR_INVALID;
}
