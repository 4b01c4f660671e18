// gorder.cpp -- Gorder vertex ordering (window w) on top of RCM, on the host.
//
// ≙ DataLoaderGorder (DataLoader.cu:789-857) -> complete_gorder (order_gorder.cu:13-31):
// relabel by RCM, build in- and out-adjacency in the new ids, then greedily append the vertex
// that shares the most in-neighbours / edges with the last w placed vertices
// (order_gorder.cu:35-143), with scores kept in the "unit heap" of Wei et al. (unitheap.cu): a
// doubly linked list sorted by key with per-key head/tail cursors and lazily applied decrements.
// The rank is identical to the reference's, tie for tie (tests compare with the oracle's literal
// restatement); the container is a flat structure-of-arrays rather than the reference's classes.
// Vertices whose out-degree exceeds sqrt(n) do not propagate scores, as in the reference.
#include <algorithm>
#include <cmath>
#include <numeric>
#include <stdexcept>
#include <vector>

#include "internal.h"

namespace flex {
namespace {

constexpr int kInf = 0x7fffffff / 2;

// keys only ever move by +-1 (unit steps), which is what makes O(1) list maintenance possible
struct UnitList {
    uint32_t nil;
    std::vector<int> key, pending;          // pending <= 0: decrements not yet applied; kInf: removed
    std::vector<uint32_t> prev, next;       // list in key-descending order
    std::vector<uint32_t> first, last;      // per key: first / last element holding it
    uint32_t top = 0;
    size_t live = 0;
    bool broken = false;

    explicit UnitList(uint32_t n) : nil(n + 2), key(n, kInf), pending(n, kInf), prev(n, nil), next(n, nil) {}

    void grow_levels(size_t want) {
        if (want > first.size()) {
            first.resize(want, nil);
            last.resize(want, nil);
        }
    }
    void add(uint32_t v, int k) {
        key[v] = k;
        pending[v] = -k;
        ++live;
    }
    void link_sorted() {
        std::vector<uint32_t> ids(live);
        std::iota(ids.begin(), ids.end(), 0u);
        std::sort(ids.begin(), ids.end(), [&](uint32_t a, uint32_t b) { return key[a] != key[b] ? key[a] > key[b] : a < b; });
        top = ids[0];
        int cur = key[top];
        grow_levels(static_cast<size_t>(10) * cur + 1);
        first[cur] = top;
        for (size_t i = 0; i < ids.size(); ++i) {
            const uint32_t v = ids[i];
            prev[v] = i ? ids[i - 1] : nil;
            next[v] = i + 1 < ids.size() ? ids[i + 1] : nil;
            if (key[v] != cur) {
                last[cur] = ids[i - 1];
                first[key[v]] = v;
                cur = key[v];
            }
        }
        last[cur] = ids.back();
    }
    void leave_level(uint32_t v, uint32_t nx, uint32_t pv) {
        const int k = key[v];
        if (first[k] == last[k]) first[k] = last[k] = nil;
        else if (first[k] == v) first[k] = nx;
        else if (last[k] == v) last[k] = pv;
    }
    void remove(uint32_t v) {
        pending[v] = kInf;
        const uint32_t pv = prev[v], nx = next[v];
        if (pv != nil) next[pv] = nx;
        if (nx != nil) prev[nx] = pv;
        leave_level(v, nx, pv);
        if (top == v) top = nx;
        prev[v] = next[v] = nil;
        --live;
    }
    // apply half of the top's pending decrement and let it sink behind every element with key >= new key
    void sink_top() {
        const uint32_t t = top, nx = next[t];
        if (nx == nil) return;
        const int k = key[t], keep = pending[t] / 2, nk = k + pending[t] - keep;
        if (-pending[t] > k) { broken = true; return; }
        if (nk >= key[nx]) return;
        pending[t] = keep;
        uint32_t tail = last[k], after = next[tail];
        while (after != nil && key[after] >= nk) {
            tail = last[key[after]];
            after = next[tail];
        }
        prev[nx] = nil;
        prev[t] = tail;
        next[t] = after;
        next[tail] = t;
        if (after != nil) prev[after] = t;
        leave_level(t, nx, nil);
        if (nk < 0) { broken = true; return; }
        key[t] = nk;
        last[nk] = t;
        if (first[nk] == nil) first[nk] = t;
        top = nx;
    }
    uint32_t pop_max() {
        uint32_t seen;
        do {
            seen = top;
            if (pending[top] < 0) sink_top();
            if (broken) return nil;
        } while (top != seen);
        remove(seen);
        return seen;
    }
    void raise(uint32_t v) {  // key += 1: hop in front of its level
        const uint32_t head = first[key[v]], pv = prev[v], nx = next[v];
        if (head != v) {
            next[pv] = nx;
            if (nx != nil) prev[nx] = pv;
            const uint32_t before = prev[head];
            prev[v] = before;
            next[v] = head;
            prev[head] = v;
            if (before != nil) next[before] = v;
        }
        leave_level(v, nx, pv);
        const int k = ++key[v];
        last[k] = v;
        if (first[k] == nil) {
            first[k] = v;
            if (k > key[top]) top = v;
        }
        if (static_cast<size_t>(k) + 4 >= first.size()) grow_levels(static_cast<size_t>(first.size() * 1.5));
    }
    void bump(uint32_t v, int by) {
        if (pending[v] == kInf) return;
        if (pending[v] == 0 && by > 0) raise(v);
        else {
            pending[v] += by;
            if (-pending[v] > key[v]) broken = true;
        }
    }
};

struct BiAdj {  // out lists in [0,n), in lists in [n,2n), neighbours ascending
    uint32_t n;
    std::vector<uint32_t> ptr, adj;
    uint32_t out_deg(uint32_t u) const { return ptr[u + 1] - ptr[u]; }
};

void slide(const BiAdj &g, UnitList &h, uint32_t huge, uint32_t in_node, uint32_t out_node,
           std::vector<uint32_t> &lost, std::vector<uint32_t> &won) {
    const uint32_t n = g.n;
    const uint32_t *op = &g.adj[0] + g.ptr[out_node + n], *oe = &g.adj[0] + g.ptr[out_node + n + 1];
    const uint32_t *np = &g.adj[0] + g.ptr[in_node + n], *ne = &g.adj[0] + g.ptr[in_node + n + 1];
    if (out_node == in_node) op = oe;  // nothing leaves the window yet
    else if (g.out_deg(out_node) <= huge)
        for (uint32_t a = g.ptr[out_node]; a < g.ptr[out_node + 1]; ++a) h.bump(g.adj[a], -1);
    lost.clear();
    won.clear();
    while (op < oe || np < ne) {  // symmetric difference of the two sorted parent lists
        if (op < oe && np < ne && *op == *np) { ++op; ++np; continue; }
        if (np < ne && (op >= oe || *np < *op)) {
            if (g.out_deg(*np) <= huge) won.push_back(*np);
            ++np;
        } else {
            if (g.out_deg(*op) <= huge) lost.push_back(*op);
            ++op;
        }
    }
    for (uint32_t p : lost) {
        h.bump(p, -1);
        for (uint32_t a = g.ptr[p]; a < g.ptr[p + 1]; ++a)
            if (g.adj[a] != out_node) h.bump(g.adj[a], -1);
    }
    if (g.out_deg(in_node) <= huge)
        for (uint32_t a = g.ptr[in_node]; a < g.ptr[in_node + 1]; ++a) h.bump(g.adj[a], +1);
    for (uint32_t p : won) {
        h.bump(p, +1);
        for (uint32_t a = g.ptr[p]; a < g.ptr[p + 1]; ++a)
            if (g.adj[a] != in_node) h.bump(g.adj[a], +1);
    }
}

}  // namespace

int order_gorder_host(int64_t n64, const uint32_t *rowPtr, const uint32_t *col, uint32_t window,
                      std::vector<uint32_t> &rank) {
    rank.assign(static_cast<size_t>(n64), 0u);
    if (n64 == 0) return FLEX_OK;
    const uint32_t n = static_cast<uint32_t>(n64);
    std::vector<uint32_t> rcm;
    int rc = order_rcm_host(n64, rowPtr, col, rcm);
    if (rc) return rc;
    const uint32_t nnz = rowPtr[n];
    BiAdj g{n, std::vector<uint32_t>(2 * static_cast<size_t>(n) + 1, 0u), std::vector<uint32_t>(2 * static_cast<size_t>(nnz))};
    for (uint32_t u = 0; u < n; ++u)
        for (uint32_t e = rowPtr[u]; e < rowPtr[u + 1]; ++e) {
            ++g.ptr[rcm[u] + 1];
            ++g.ptr[rcm[col[e]] + n + 1];
        }
    for (size_t i = 0; i < 2 * static_cast<size_t>(n); ++i) g.ptr[i + 1] += g.ptr[i];
    {
        std::vector<uint32_t> cur(g.ptr.begin(), g.ptr.end() - 1);
        for (uint32_t u = 0; u < n; ++u)
            for (uint32_t e = rowPtr[u]; e < rowPtr[u + 1]; ++e) {
                const uint32_t a = rcm[u], b = rcm[col[e]];
                g.adj[cur[a]++] = b;
                g.adj[cur[b + n]++] = a;
            }
    }
    for (size_t i = 0; i < 2 * static_cast<size_t>(n); ++i) std::sort(g.adj.begin() + g.ptr[i], g.adj.begin() + g.ptr[i + 1]);

    UnitList h(n);
    const uint32_t huge = static_cast<uint32_t>(std::sqrt(static_cast<double>(n)));
    for (uint32_t u = 0; u < n; ++u) {
        const uint32_t din = g.ptr[u + n + 1] - g.ptr[u + n];
        // the reference cannot order a graph with an isolated vertex (unitheap.cu:35-38 links the
        // wrong ids and sizes its key table from INT_MAX/2): refuse instead of imitating the crash
        if (din + g.out_deg(u) == 0) return FLEX_ERR_UNSUPPORTED;
        h.add(u, static_cast<int>(din));
    }
    h.link_sorted();
    std::vector<uint32_t> order, lost, won;
    order.reserve(n);
    const uint32_t hub = h.top;
    order.push_back(hub);
    h.remove(hub);
    slide(g, h, huge, hub, hub, lost, won);
    while (h.live > 0) {
        const uint32_t v = h.pop_max();
        if (h.broken || v >= n) return FLEX_ERR_INVALID;
        order.push_back(v);
        const uint32_t leaving = order.size() > window ? order[order.size() - window - 1] : v;
        slide(g, h, huge, v, leaving, lost, won);
        if (h.broken) return FLEX_ERR_INVALID;
    }
    std::vector<uint32_t> pos(n);
    for (uint32_t i = 0; i < n; ++i) pos[order[i]] = i;
    for (uint32_t u = 0; u < n; ++u) rank[u] = pos[rcm[u]];
    return FLEX_OK;
}

}  // namespace flex

extern "C" int flex_order_gorder(const flex_csr *A, uint32_t window, uint32_t *rank) try {
    if (!rank || window == 0) return FLEX_ERR_INVALID;
    int rc = flex::validate_csr(A);
    if (rc) return rc;
    if (A->m != A->n) return FLEX_ERR_INVALID;
    std::vector<uint32_t> r;
    rc = flex::order_gorder_host(A->m, A->rowPtr, A->col, window, r);
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
