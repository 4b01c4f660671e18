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
// next), never a data permutation: see plan_build.cpp.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <stdexcept>
#include <vector>

#include "host_parallel.h"
#include "internal.h"

namespace flex {

namespace {

struct Clusterer {
    const int64_t n;
    const uint32_t *rowPtr, *col;
    std::vector<uint32_t> parent;                                 // union-find over communities
    std::vector<double> cdeg;                                     // community degree (sum of member degrees)
    std::vector<uint32_t> deg0;                                   // a vertex's own degree, self loop not counted (kept for the second stage)
    using Adj = std::vector<std::pair<uint32_t, float>>;
    std::vector<Adj> adj;                                          // a community's adjacency as of its last compaction
    std::vector<std::vector<Adj>> segs;                           // + the lists of the communities it has absorbed since (moved, not copied)
    std::vector<uint8_t> raw;                                     // own adjacency still lives in the CSR (never copied out while it does)
    std::vector<std::vector<uint32_t>> rawm;                      // + the vertices absorbed in that state: their CSR rows are read in place
    std::vector<std::vector<uint32_t>> children;                  // merge forest
    double M = 0;                                                 // total degree (2m)
    int64_t batch = 4096;                                         // proposals per batch (flex_cluster_tuning.batch)

    Clusterer(int64_t n_, const uint32_t *rp, const uint32_t *c)
        : n(n_), rowPtr(rp), col(c), parent(n_), cdeg(n_, 0.0), deg0(n_), adj(n_), segs(n_), raw(n_, 1), rawm(n_), children(n_) {
        std::iota(parent.begin(), parent.end(), 0u);
        constexpr int64_t kBlk = 1 << 14;
        const int64_t nblk = (n + kBlk - 1) / kBlk;
        std::vector<double> part(static_cast<size_t>(nblk), 0.0);
        parallel_chunks(nblk, [&](int64_t b) {
            double m = 0;
            for (int64_t u = b * kBlk; u < std::min(n, (b + 1) * kBlk); ++u) {
                uint32_t d = 0;
                for (uint32_t e = rowPtr[u]; e < rowPtr[u + 1]; ++e) d += (col[e] != u);
                cdeg[u] = d;
                deg0[u] = d;
                m += d;
            }
            part[b] = m;
        });
        for (double m : part) M += m;  // block order: the same sum whatever the thread count
    }

    uint32_t find(uint32_t x) {
        while (parent[x] != x) {
            parent[x] = parent[parent[x]];
            x = parent[x];
        }
        return x;
    }

    // Rebuild u's adjacency keyed by the roots of the state the round started from (`parent` is flat then: parent[v]
    // IS v's root); returns the best merge target or u itself, and the weight of the edge bundle to it.
    // Touches only adj[u] and the caller's scratch, so distinct communities can be examined concurrently.
    uint32_t compact_and_pick(uint32_t u, std::vector<float> &acc, std::vector<uint32_t> &touched, float *w_best) {
        touched.clear();
        auto add = [&](uint32_t v, float w) {
            uint32_t r = v;  // read-only find: no merge happens while proposals are computed
            while (parent[r] != r) r = parent[r];
            if (r == u) return;
            if (acc[r] == 0.f) touched.push_back(r);
            acc[r] += w;  // weights are edge counts: exact in fp32, so the order the neighbours are met in does not matter
        };
        // (a 16 KB per-thread hash table for the small communities instead of the dense array of n floats was tried:
        //  round 1 of the Amazon shape 1.15 -> 1.4 s on 8 cores -- the dense array's misses overlap, the probing does not)
        // a vertex that has absorbed nothing keeps reading its CSR row: writing every row out as a list in round 1 and
        // reading it back in round 2 was 2 x 8 B per nonzero of first-touched heap (0.6 s of a cold Amazon-size plan)
        const bool leaf = raw[u] && rawm[u].empty() && segs[u].empty();
        if (raw[u])
            for (uint32_t e = rowPtr[u]; e < rowPtr[u + 1]; ++e) add(col[e], 1.f);
        for (uint32_t m : rawm[u])
            for (uint32_t e = rowPtr[m]; e < rowPtr[m + 1]; ++e) add(col[e], 1.f);
        for (const auto &vw : adj[u]) add(vw.first, vw.second);
        for (const Adj &sg : segs[u])
            for (const auto &vw : sg) add(vw.first, vw.second);
        auto &list = adj[u];
        if (!leaf) {
            raw[u] = 0;
            std::vector<uint32_t>().swap(rawm[u]);
            std::vector<Adj>().swap(segs[u]);
            list.clear();
            list.reserve(touched.size());
        }
        uint32_t best = u;
        double best_gain = 0.0;
        *w_best = 0.f;
        const double du_over_M = cdeg[u] / M;
        for (uint32_t r : touched) {
            const float w = acc[r];
            acc[r] = 0.f;
            if (!leaf) list.emplace_back(r, w);
            const double gain = static_cast<double>(w) - cdeg[r] * du_over_M;  // ~ delta modularity * M
            // ties go to the smaller community id, whatever order the neighbours were met in
            if (gain > best_gain || (gain == best_gain && gain > 0.0 && r < best)) {
                best_gain = gain;
                best = r;
                *w_best = w;
            }
        }
        return best;
    }

    // Rounds.  Each round has three steps, and its result does not depend on the number of threads:
    //  A (parallel)   every active community compacts its adjacency and PROPOSES the neighbour with the largest
    //                 modularity gain, all from the state the round started with;
    //  B (sequential, O(#active)) the proposals are applied lowest-degree community first: u joins the current root r
    //                 of its choice unless u has absorbed something this round, or the gain -- re-priced with r's
    //                 degree as it stands now, so that a popular community does not swallow a whole round of
    //                 proposers -- is no longer positive (then u proposes again next round);
    //  C (sequential, O(#merges)) what the absorbed communities held is HANDED to their new roots -- vertex ids for rows still in
    //                 the CSR, whole lists (moved, never copied) otherwise -- and read when the root is next examined.
    // (Round 1 of this project examined and merged one community at a time, each seeing every earlier merge: the
    //  Amazon shape spent 8.4 s of its 9.3 s plan there, single-threaded.)
    void run(int max_rounds) {
        if (M <= 0) return;
        const int nthr = host_threads();
        WorkerPool pool(nthr);  // one set of threads for the few hundred small parallel sections below
        std::vector<std::vector<float>> acc(static_cast<size_t>(nthr));
        std::vector<std::vector<uint32_t>> touched(static_cast<size_t>(nthr));
        std::vector<uint32_t> cur(static_cast<size_t>(n)), next, pick(static_cast<size_t>(n));
        std::vector<float> pick_w(static_cast<size_t>(n));
        std::iota(cur.begin(), cur.end(), 0u);
        std::vector<uint32_t> stamp(static_cast<size_t>(n), 0u);
        std::vector<std::pair<uint32_t, uint32_t>> moved;  // (new root, absorbed community) of the current batch, in the order decided
        const int64_t env_batch = batch;
        const bool timing = plan_timing_enabled();
        double t_sort = 0, t_a = 0, t_b = 0, t_c = 0;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto secs = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
        for (int round = 1; round <= max_rounds && !cur.empty(); ++round) {
            const auto ts0 = now();
            // lowest community degree first, ties in list order.  Degrees are sums of integer degrees: when their range is
            // small against the list (round 1: every vertex, degrees < n) a stable counting sort replaces the comparison sort
            {
                double dmax = 0;
                for (uint32_t u : cur) dmax = std::max(dmax, cdeg[u]);
                if (cur.size() >= (1u << 16) && dmax < 4.0 * static_cast<double>(cur.size())) {
                    const size_t nb = static_cast<size_t>(dmax) + 2;
                    std::vector<uint32_t> count(nb, 0u), sorted(cur.size());
                    for (uint32_t u : cur) ++count[static_cast<size_t>(cdeg[u]) + 1];
                    for (size_t i = 1; i < nb; ++i) count[i] += count[i - 1];
                    for (uint32_t u : cur) sorted[count[static_cast<size_t>(cdeg[u])]++] = u;
                    cur.swap(sorted);
                } else {
                    std::stable_sort(cur.begin(), cur.end(), [&](uint32_t a, uint32_t b) { return cdeg[a] < cdeg[b]; });
                }
            }
            t_sort += secs(ts0, now());
            next.clear();
            moved.clear();
            const int64_t ncur = static_cast<int64_t>(cur.size());
            const int64_t batch = std::max<int64_t>(256, env_batch);
            for (int64_t b0 = 0; b0 < ncur; b0 += batch) {
                const int64_t b1 = std::min(ncur, b0 + batch);
                // A: proposals of this batch, from the state left by the batches before it
                const auto ta0 = now();
                // late rounds: a few hundred communities that hold the whole graph between them -- one per work item
                const int64_t kBlk = b1 - b0 >= 4096 ? 64 : 1;
                pool.run((b1 - b0 + kBlk - 1) / kBlk, [&](int64_t b, int tid) {
                    if (acc[tid].empty()) acc[tid].assign(static_cast<size_t>(n), 0.f);
                    for (int64_t i = b0 + b * kBlk; i < std::min(b1, b0 + (b + 1) * kBlk); ++i) {
                        const uint32_t u = cur[i];
                        if (parent[u] != u || stamp[u] == static_cast<uint32_t>(round)) {
                            pick[u] = u;  // merged away or just absorbed something in an earlier batch: not its turn
                            continue;
                        }
                        pick[u] = compact_and_pick(u, acc[tid], touched[tid], &pick_w[u]);
                    }
                });
                // B
                const auto tb0 = now();
                t_a += secs(ta0, tb0);
                for (int64_t i = b0; i < b1; ++i) {
                    const uint32_t u = cur[i];
                    const uint32_t v = pick[u];
                    if (v == u) continue;                                     // no neighbour worth joining (or not its turn)
                    if (stamp[u] == static_cast<uint32_t>(round)) continue;  // absorbed something inside this batch
                    const uint32_t r = find(v);
                    const bool gain_left = r != u && static_cast<double>(pick_w[u]) - cdeg[r] * cdeg[u] / M > 0.0;
                    if (!gain_left) {  // its choice moved on or filled up: look again next round
                        stamp[u] = static_cast<uint32_t>(round);
                        next.push_back(u);
                        continue;
                    }
                    parent[u] = r;  // u joins r
                    cdeg[r] += cdeg[u];
                    children[r].push_back(u);
                    moved.emplace_back(r, u);
                    if (stamp[r] != static_cast<uint32_t>(round)) {
                        stamp[r] = static_cast<uint32_t>(round);
                        next.push_back(r);
                    }
                }
                // C: the absorbed communities' adjacency lists go to their new roots -- handed over, not copied (copying
                // them was 12.8 s of the Amazon shape's 17 s): the root reads them when it is next compacted
                const auto tc0 = now();
                t_b += secs(tb0, tc0);
                for (const auto &ru : moved) {
                    const uint32_t r = ru.first, u = ru.second;
                    if (raw[u]) rawm[r].push_back(u);
                    if (!rawm[u].empty()) {
                        rawm[r].insert(rawm[r].end(), rawm[u].begin(), rawm[u].end());
                        std::vector<uint32_t>().swap(rawm[u]);
                    }
                    auto &dst = segs[r];
                    if (!adj[u].empty()) dst.push_back(std::move(adj[u]));
                    for (Adj &sg : segs[u]) dst.push_back(std::move(sg));
                    Adj().swap(adj[u]);
                    std::vector<Adj>().swap(segs[u]);
                }
                moved.clear();
                t_c += secs(tc0, now());
            }
            if (timing) std::fprintf(stderr, "cluster: round %d active %zu -> %zu  sort %.2f A %.2f B %.2f C %.2f s (cumulative)\n", round, cur.size(), next.size(), t_sort, t_a, t_b, t_c);
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

// Second stage of the community order: vertex moves between neighbouring stretches of the order.
//
// The merge forest's depth-first order puts a community's members next to each other, but agglomeration never undoes a merge:
// a vertex that joined a neighbour early stays with it even when most of its edges later turn out to lead elsewhere, and
// two communities tied by many edges end up interleaved.  Here the order is cut into stretches of kStretch positions (labels),
// and for a few synchronous sweeps (at most 8; until fewer than 3 % of the vertices move: 4 on the amazon shape, 6 on reddit)
// every vertex moves to the label that holds most of its neighbours, priced like a modularity
// move: gain(l) = w(v,l) - deg(v) * D(l) / M with D the label's total degree (its own degree taken out of the label it is in).
// Labels empty out or grow to the size of the communities the graph really has; the new order is label by label in the order
// of the labels' mean old position, members in their old order.
// Measured (MI355X, k = 128, community order before -> after, tools/probe_refine.py): amazon shape 8.78 -> 8.25 ms (order of the
// generator's own communities: 8.13), reddit shape 672 -> 641 us (632; k = 32: 157 -> 152), flickr 38.0 -> 36.9, yelp 518 -> 517;
// share of the edges within 2048 positions 0.20 -> 0.30 (0.34) on the amazon shape, 0.32 -> 0.45 (0.48) on reddit.  Stretch
// length: 256 / 512 / 1024 / 2048 are the same on reddit (639-641 us) and amazon (8.26 / 8.25 ms at 512 / 1024); on the low-degree
// shapes short stretches are noisy (yelp 538 / 524 / 517 / 513 us, flickr 38.0 / 37.5 / 36.9 / 36.9): 1024.  The passes
// over the edges read long rows sampled (below): the same locality figures to three digits with every 1st, 2nd, 4th or 8th
// neighbour asked.  Every decision of a sweep reads the labels of the sweep before: the result does not depend on the thread count.
// How many (sampled) edges join two vertices at most `window` positions apart: the yardstick the second stage is held to.
int64_t edges_within(int64_t n, const uint32_t *rowPtr, const uint32_t *col, const std::vector<uint32_t> &rank, uint32_t window) {
    constexpr int64_t kBlk = 4096;
    const int64_t nblk = (n + kBlk - 1) / kBlk;
    std::vector<int64_t> part(static_cast<size_t>(nblk), 0);
    parallel_chunks(nblk, [&](int64_t b) {
        int64_t c = 0;
        for (int64_t v = b * kBlk; v < std::min(n, (b + 1) * kBlk); ++v) {
            const uint32_t stride = std::clamp<uint32_t>((rowPtr[v + 1] - rowPtr[v]) / 32u, 1u, 8u);
            const uint32_t rv = rank[v];
            for (uint32_t e = rowPtr[v]; e < rowPtr[v + 1]; e += stride) {
                const uint32_t ru = rank[col[e]];
                c += (ru > rv ? ru - rv : rv - ru) <= window;
            }
        }
        part[static_cast<size_t>(b)] = c;
    });
    int64_t total = 0;
    for (int64_t c : part) total += c;
    return total;
}

void refine_by_label_moves(int64_t n, const uint32_t *rowPtr, const uint32_t *col, const std::vector<uint32_t> &deg, std::vector<uint32_t> &rank,
                           const flex_cluster_tuning &tn) {
    const int64_t kStretch = tn.stretch >= 16 ? tn.stretch : 1024;  // tuning experiments
    const int kSweeps = tn.sweeps > 0 ? tn.sweeps : 8;
    if (n < 4 * kStretch) return;
    const uint32_t L = static_cast<uint32_t>((n + kStretch - 1) / kStretch);
    std::vector<uint32_t> lab(static_cast<size_t>(n)), next(static_cast<size_t>(n));
    constexpr int64_t kBlk = 2048;  // vertices per work item
    constexpr uint32_t kSampleFrom = 32;  // a row is sampled so that about this many of its entries (at least) are read
    const uint32_t max_stride = tn.stride > 0 ? static_cast<uint32_t>(tn.stride) : 4u;
    const int64_t nblk = (n + kBlk - 1) / kBlk;
    for (int64_t v = 0; v < n; ++v) lab[v] = static_cast<uint32_t>(rank[v] / kStretch);
    double M = 0;
    for (int64_t v = 0; v < n; ++v) M += deg[v];
    if (M <= 0) return;
    std::vector<double> D(L);
    std::vector<std::vector<float>> acc(static_cast<size_t>(host_threads()));
    std::vector<std::vector<uint32_t>> touched(acc.size());
    std::vector<int64_t> moved_blk(static_cast<size_t>(nblk));
    int64_t prev_moved = 0;
    for (int sweep = 0; sweep < kSweeps; ++sweep) {
        std::fill(D.begin(), D.end(), 0.0);
        for (int64_t v = 0; v < n; ++v) D[lab[v]] += deg[v];  // vertex order: the same sums whatever the thread count
        parallel_chunks_tid(nblk, [&](int64_t b, int tid) {
            std::vector<float> &a = acc[static_cast<size_t>(tid)];
            std::vector<uint32_t> &t = touched[static_cast<size_t>(tid)];
            if (a.empty()) a.assign(L, 0.f);
            int64_t moved = 0;
            for (int64_t v = b * kBlk; v < std::min(n, (b + 1) * kBlk); ++v) {
                const uint32_t cur = lab[v];
                next[v] = cur;
                if (deg[v] == 0) continue;
                t.clear();
                uint32_t same = 0;  // neighbours in v's own label: most of them, counted in a register (a[cur] += 1 per edge
                                    // was one store-to-load chain through the whole row)
                // A long row says the same thing many times over: every `stride`-th neighbour is asked (a different
                // residue each sweep), and the counts are scaled back.  Rows of < 2 * kSampleFrom entries are read whole.
                const uint32_t e_beg = rowPtr[v], e_end = rowPtr[v + 1];
                const uint32_t stride = std::clamp<uint32_t>((e_end - e_beg) / kSampleFrom, 1u, max_stride);
                for (uint32_t e = e_beg + static_cast<uint32_t>(sweep) % stride; e < e_end; e += stride) {
                    const uint32_t u = col[e];
                    const uint32_t l = lab[u];
                    if (l == cur) {
                        same += (u != v);
                        continue;
                    }
                    if (a[l] == 0.f) t.push_back(l);
                    a[l] += 1.f;  // edge counts: exact in fp32
                }
                const double scale = stride;
                const double dv = deg[v], dv_over_M = dv / M;
                uint32_t best = cur;
                double best_gain = scale * same - (D[cur] - dv) * dv_over_M;  // staying put
                for (uint32_t l : t) {
                    const double g = scale * a[l] - D[l] * dv_over_M;
                    // ties: stay; between two other labels, the smaller id (the neighbours are met in CSR order, but so that nothing depends on it)
                    if (g > best_gain || (g == best_gain && best != cur && l < best)) {
                        best_gain = g;
                        best = l;
                    }
                }
                for (uint32_t l : t) a[l] = 0.f;
                next[v] = best;
                moved += best != cur;
            }
            moved_blk[static_cast<size_t>(b)] = moved;
        });
        lab.swap(next);
        int64_t moved = 0;
        for (int64_t m : moved_blk) moved += m;
        if (plan_timing_enabled()) std::fprintf(stderr, "cluster: sweep %d moved %lld of %lld\n", sweep + 1, static_cast<long long>(moved), static_cast<long long>(n));
        if (moved * 100 < 3 * n) break;  // < 3 % of the vertices moved: what later sweeps add is within the noise of the launch time
        if (sweep > 0 && moved > prev_moved) return;  // more moves than the sweep before: the labels are chasing hubs, not settling (R-MAT): keep the walk's order
        prev_moved = moved;
    }
    // labels in the order of their members' mean old position; members keep their old order
    std::vector<double> sum_pos(L, 0.0);
    std::vector<uint32_t> cnt(L, 0u), inv(static_cast<size_t>(n));
    for (int64_t v = 0; v < n; ++v) {
        sum_pos[lab[v]] += rank[v];
        ++cnt[lab[v]];
        inv[rank[v]] = static_cast<uint32_t>(v);
    }
    std::vector<uint32_t> by_pos;
    for (uint32_t l = 0; l < L; ++l)
        if (cnt[l]) by_pos.push_back(l);
    std::sort(by_pos.begin(), by_pos.end(), [&](uint32_t x, uint32_t y) {
        const double mx = sum_pos[x] / cnt[x], my = sum_pos[y] / cnt[y];
        return mx < my || (mx == my && x < y);
    });
    std::vector<uint32_t> first(L, 0u);
    uint32_t at = 0;
    for (uint32_t l : by_pos) {
        first[l] = at;
        at += cnt[l];
    }
    std::vector<uint32_t> moved_rank(static_cast<size_t>(n));
    for (int64_t pos = 0; pos < n; ++pos) {
        const uint32_t v = inv[pos];
        moved_rank[v] = first[lab[v]]++;
    }
    // The moves pay on graphs that HAVE communities (all the GNN shapes: +30-50 % of the edges within 2048 positions).  On a
    // graph without them -- an R-MAT / Kronecker graph: labels collapse around the hubs -- they make the order worse (share
    // within 2048 positions 0.105 -> 0.068 on a scale-18 R-MAT).  So the result is held to that yardstick and only kept when
    // it is better than the walk's by a margin.
    constexpr uint32_t kWindow = 2048;
    const int64_t before = edges_within(n, rowPtr, col, rank, kWindow), after = edges_within(n, rowPtr, col, moved_rank, kWindow);
    if (plan_timing_enabled()) std::fprintf(stderr, "cluster: (sampled) edges within %u positions: %lld -> %lld\n", kWindow, static_cast<long long>(before), static_cast<long long>(after));
    if (after * 100 > before * 102) rank.swap(moved_rank);
}

}  // namespace

int order_cluster_host(int64_t n, const uint32_t *rowPtr, const uint32_t *col, std::vector<uint32_t> &rank, const flex_cluster_tuning *tuning) {
    const flex_cluster_tuning tn = tuning ? *tuning : flex_cluster_tuning{};
    if (n == 0) {
        rank.clear();
        return FLEX_OK;
    }
    for (uint32_t e = 0; e < rowPtr[n]; ++e)
        if (col[e] >= n) return FLEX_ERR_INVALID;
    try {
        Clusterer c(n, rowPtr, col);
        if (tn.batch > 0) c.batch = tn.batch;
        c.run(32);
        c.order(rank);
        if (!tn.no_refine) {
            const auto t0 = std::chrono::steady_clock::now();
            refine_by_label_moves(n, rowPtr, col, c.deg0, rank, tn);
            if (plan_timing_enabled())
                std::fprintf(stderr, "cluster: vertex moves %.2f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        }
    } catch (const std::bad_alloc &) {
        return FLEX_ERR_NOMEM;
    }
    return FLEX_OK;
}

}  // namespace flex

extern "C" int flex_order_cluster(const flex_csr *A, uint32_t *rank) { return flex_order_cluster_ex(A, nullptr, rank); }

extern "C" int flex_order_cluster_ex(const flex_csr *A, const flex_cluster_tuning *tuning, uint32_t *rank) try {
    if (!rank) return FLEX_ERR_INVALID;
    if (tuning && (tuning->batch < 0 || tuning->no_refine < 0 || tuning->stretch < 0 || tuning->sweeps < 0 || tuning->stride < 0)) return FLEX_ERR_INVALID;
    int rc = flex::validate_csr(A);
    if (rc) return rc;
    if (A->m != A->n) return FLEX_ERR_INVALID;
    std::vector<uint32_t> r;
    rc = flex::order_cluster_host(A->m, A->rowPtr, A->col, r, tuning);
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
