// synth.cpp -- deterministic synthetic graphs with the shapes the reference's README
// lists (README.md:13-20).  Only pubmed.csv ships with the reference; flickr / reddit /
// amazon / yelp / ppi are absent, so benchmarks use stand-ins with the exact n and nnz:
// symmetric, one self-loop per row (the reference's tilers need non-empty rows and a
// diagonal entry, mat.cu:1207, 718), sorted columns, Chung-Lu power-law degrees, planted
// communities (with a "near" ring so reordering has locality to find) and, optionally, a
// random vertex relabel so that the natural order is not accidentally banded.
// The result depends only on the parameters, never on the thread count.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <thread>
#include <stdexcept>
#include <vector>

#include "host_parallel.h"
#include "internal.h"

namespace {

using flex::parallel_chunks;

inline uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() { return mix64(s++); }
    double uni() { return static_cast<double>(next() >> 11) * (1.0 / 9007199254740992.0); }  // [0,1)
};

// sort + unique of 64-bit keys whose high word is < n: bucket by high word, sort buckets in parallel
void sort_unique(std::vector<uint64_t> &keys, uint64_t n) {
    if (keys.size() < (1u << 16)) {
        std::sort(keys.begin(), keys.end());
        keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
        return;
    }
    const int nb = 256;
    auto bucket_of = [&](uint64_t k) { return static_cast<int>(((k >> 32) * nb) / n); };
    std::vector<size_t> cnt(nb + 1, 0);
    for (uint64_t k : keys) ++cnt[bucket_of(k) + 1];
    for (int b = 0; b < nb; ++b) cnt[b + 1] += cnt[b];
    std::vector<uint64_t> tmp(keys.size());
    {
        std::vector<size_t> cur(cnt.begin(), cnt.end() - 1);
        for (uint64_t k : keys) tmp[cur[bucket_of(k)]++] = k;
    }
    std::vector<size_t> uniq(nb, 0);
    parallel_chunks(nb, [&](int64_t b) {
        auto beg = tmp.begin() + cnt[b], end = tmp.begin() + cnt[b + 1];
        std::sort(beg, end);
        uniq[b] = static_cast<size_t>(std::unique(beg, end) - beg);
    });
    size_t o = 0;
    for (int b = 0; b < nb; ++b) {
        std::copy(tmp.begin() + cnt[b], tmp.begin() + cnt[b] + uniq[b], keys.begin() + o);
        o += uniq[b];
    }
    keys.resize(o);
}

}  // namespace

extern "C" int flex_synth_preset(const char *name, int scale, flex_synth_params *out) {
    if (!name || !out || scale < 1) return FLEX_ERR_INVALID;
    struct Preset {
        const char *name;
        int64_t n, nnz;
        double alpha;
        int64_t community;
        double p_in, p_near;
        int gcn, directed;
    };
    // shapes: README.md:13-20 (GNN graphs) and SURVEY 8(d) (the two SuiteSparse matrices that
    // data/SuiteSparse/prepare_mtx_data.sh fetches); structure parameters are this project's choice
    static const Preset table[] = {
        {"amazon", 1569960, 264339468, 2.1, 4096, 0.60, 0.25, 0, 0},
        {"flickr", 89250, 989006, 2.3, 256, 0.55, 0.25, 1, 0},
        {"ppi", 14755, 458973, 2.4, 128, 0.60, 0.20, 1, 0},
        {"pubmed", 19717, 108365, 2.6, 64, 0.60, 0.25, 1, 0},
        {"reddit", 232965, 23446803, 2.1, 2048, 0.60, 0.25, 1, 0},
        {"soc-sign-epinions", 131828, 841372, 2.2, 128, 0.40, 0.20, 0, 1},
        {"wiki-vote", 8297, 103689, 2.2, 64, 0.30, 0.20, 0, 1},
        {"yelp", 716847, 13954819, 2.2, 512, 0.55, 0.25, 1, 0},
    };
    for (size_t i = 0; i < sizeof table / sizeof table[0]; ++i) {
        if (std::strcmp(name, table[i].name) != 0) continue;
        const Preset &t = table[i];
        int64_t nnz = t.nnz;
        if (!t.directed && ((nnz - t.n) & 1)) --nnz;  // symmetric + one self loop per row needs nnz - n even
        *out = flex_synth_params{t.n * scale, nnz * scale, t.alpha, t.community, t.p_in, t.p_near, 8, 1, t.gcn,
                                 t.directed, 0xF1E0ull + i};
        return FLEX_OK;
    }
    return FLEX_ERR_INVALID;
}

extern "C" int flex_synth_graph(const flex_synth_params *p, flex_host_csr *out) try {
    if (!p || !out) return FLEX_ERR_INVALID;
    std::memset(out, 0, sizeof *out);
    const int64_t n = p->n, nnz = p->nnz;
    const bool directed = p->directed != 0;
    if (n <= 0 || n >= INT32_MAX || nnz >= (int64_t(1) << 32)) return FLEX_ERR_INVALID;
    if (directed ? nnz < 0 : (nnz < n || ((nnz - n) & 1))) return FLEX_ERR_INVALID;
    const int64_t E = directed ? nnz : (nnz - n) / 2;
    if (n < 2 ? E > 0 : static_cast<double>(E) > 0.25 * static_cast<double>(n) * (n - 1)) return FLEX_ERR_INVALID;
    const double alpha = p->alpha > 1.5 ? p->alpha : 2.1;
    Rng rng(mix64(p->seed) ^ 0xF1E0);

    // expected-degree weights: w ~ rank^(-1/(alpha-1)), ranks dealt to vertices at random
    std::vector<uint32_t> perm(static_cast<size_t>(n));
    std::iota(perm.begin(), perm.end(), 0u);
    for (int64_t i = n - 1; i > 0; --i) std::swap(perm[i], perm[rng.next() % static_cast<uint64_t>(i + 1)]);
    std::vector<double> w(static_cast<size_t>(n));
    const double ex = -1.0 / (alpha - 1.0);
    for (int64_t i = 0; i < n; ++i) w[i] = std::pow(static_cast<double>(perm[i]) + 1.0, ex);
    // cap the heaviest expected degree (real GNN graphs: reddit's hub has ~2e4 neighbours)
    const double avgdeg = static_cast<double>(nnz) / n;
    const double capdeg = std::min(static_cast<double>(n) / 8.0, std::max(1000.0, 200.0 * avgdeg));
    for (int it = 0; it < 4 && E > 0; ++it) {
        const double W = std::accumulate(w.begin(), w.end(), 0.0);
        const double wcap = capdeg * W / (2.0 * E);
        for (auto &x : w) x = std::min(x, wcap);
    }
    std::vector<double> P(static_cast<size_t>(n) + 1, 0.0);
    for (int64_t i = 0; i < n; ++i) P[i + 1] = P[i] + w[i];

    // communities = contiguous id ranges of random size around p->community
    std::vector<uint32_t> cstart{0};
    if (p->community > 0) {
        while (static_cast<int64_t>(cstart.back()) < n) {
            const double f = std::exp2(2.0 * rng.uni() - 1.0);
            const int64_t sz = std::max<int64_t>(4, static_cast<int64_t>(p->community * f));
            cstart.push_back(static_cast<uint32_t>(std::min<int64_t>(n, cstart.back() + sz)));
        }
    } else {
        cstart.push_back(static_cast<uint32_t>(n));
    }
    const int64_t ncomm = static_cast<int64_t>(cstart.size()) - 1;
    std::vector<uint32_t> comm_of(static_cast<size_t>(n));
    for (int64_t c = 0; c < ncomm; ++c)
        for (uint32_t v = cstart[c]; v < cstart[c + 1]; ++v) comm_of[v] = static_cast<uint32_t>(c);

    auto pick = [&](Rng &r, uint32_t lo, uint32_t hi) {  // vertex in [lo,hi) with probability ~ w
        const double x = P[lo] + r.uni() * (P[hi] - P[lo]);
        const auto it = std::upper_bound(P.begin() + lo + 1, P.begin() + hi, x);
        return static_cast<uint32_t>(it - P.begin() - 1);
    };
    const int win = p->near_window > 0 ? p->near_window : 8;
    auto generate = [&](int64_t count, int round, double p_in, double p_near, std::vector<uint64_t> &keys) {
        constexpr int64_t kChunk = 1 << 16;
        const int64_t nchunks = (count + kChunk - 1) / kChunk;
        keys.assign(static_cast<size_t>(count), ~0ull);
        parallel_chunks(nchunks, [&](int64_t c) {
            Rng r(mix64(p->seed * 0x100000001B3ull + static_cast<uint64_t>(round)) ^ mix64(static_cast<uint64_t>(c) + 77));
            const int64_t b = c * kChunk, e = std::min(count, b + kChunk);
            for (int64_t i = b; i < e; ++i) {
                const uint32_t u = pick(r, 0, static_cast<uint32_t>(n));
                const double mode = r.uni();
                uint32_t lo = 0, hi = static_cast<uint32_t>(n);
                if (mode < p_in) {
                    lo = cstart[comm_of[u]];
                    hi = cstart[comm_of[u] + 1];
                } else if (mode < p_in + p_near) {
                    const int64_t cu = comm_of[u];
                    lo = cstart[std::max<int64_t>(0, cu - win)];
                    hi = cstart[std::min<int64_t>(ncomm, cu + win + 1)];
                }
                uint32_t v = u;
                for (int tries = 0; tries < 4 && v == u; ++tries) v = pick(r, lo, hi);
                if (v == u) continue;  // stays ~0 -> dropped
                keys[i] = (static_cast<uint64_t>(std::min(u, v)) << 32) | std::max(u, v);
            }
        });
        keys.erase(std::remove(keys.begin(), keys.end(), ~0ull), keys.end());
        sort_unique(keys, static_cast<uint64_t>(n));
    };

    // rounds of candidates until exactly E distinct undirected edges exist
    std::vector<uint64_t> edges, cand, fresh;
    double p_in = p->community > 0 ? p->p_in : 0.0, p_near = p->community > 0 ? p->p_near : 0.0;
    for (int round = 0; static_cast<int64_t>(edges.size()) < E; ++round) {
        if (round >= 200) return FLEX_ERR_UNSUPPORTED;  // cannot place that many distinct edges
        const int64_t need = E - static_cast<int64_t>(edges.size());
        generate(need + need / 16 + 1024, round, p_in, p_near, cand);
        fresh.clear();
        std::set_difference(cand.begin(), cand.end(), edges.begin(), edges.end(), std::back_inserter(fresh));
        if (static_cast<int64_t>(fresh.size()) > need) {  // keep a hash-random subset of exactly `need`
            std::sort(fresh.begin(), fresh.end(), [](uint64_t a, uint64_t b) { return mix64(a) < mix64(b); });
            fresh.resize(static_cast<size_t>(need));
            std::sort(fresh.begin(), fresh.end());
        }
        const size_t old = edges.size();
        edges.insert(edges.end(), fresh.begin(), fresh.end());
        std::inplace_merge(edges.begin(), edges.begin() + old, edges.end());
        if (round >= 2) {  // saturated communities: push the remaining demand outwards
            p_in *= 0.7;
            p_near *= 0.85;
        }
    }
    cand = std::vector<uint64_t>();
    fresh = std::vector<uint64_t>();

    // optional relabel
    std::vector<uint32_t> relabel(static_cast<size_t>(n));
    std::iota(relabel.begin(), relabel.end(), 0u);
    if (p->shuffle) {
        Rng r2(mix64(p->seed ^ 0xABCDEF));
        for (int64_t i = n - 1; i > 0; --i) std::swap(relabel[i], relabel[r2.next() % static_cast<uint64_t>(i + 1)]);
    }

    // CSR: degree count -> scatter -> per-row sort
    out->m = out->n = static_cast<int32_t>(n);
    out->nnz = nnz;
    out->rowPtr = static_cast<uint32_t *>(std::calloc(static_cast<size_t>(n) + 1, sizeof(uint32_t)));
    out->col = static_cast<uint32_t *>(std::malloc(static_cast<size_t>(nnz) * sizeof(uint32_t)));
    out->vals = static_cast<float *>(std::malloc(static_cast<size_t>(nnz) * sizeof(float)));
    if (!out->rowPtr || !out->col || !out->vals) {
        flex_host_csr_free(out);
        return FLEX_ERR_NOMEM;
    }
    uint32_t *rp = out->rowPtr;
    // directed stand-ins keep each edge in one direction chosen by a hash of the pair
    auto flipped = [&](uint64_t key) { return directed && (mix64(key ^ p->seed) & 1); };
    if (!directed)
        for (int64_t i = 0; i < n; ++i) rp[i + 1] = 1;  // self loop
    for (uint64_t key : edges) {
        const uint32_t a = relabel[key >> 32], b = relabel[key & 0xFFFFFFFFu];
        if (!directed) {
            ++rp[a + 1];
            ++rp[b + 1];
        } else {
            ++rp[(flipped(key) ? b : a) + 1];
        }
    }
    for (int64_t i = 0; i < n; ++i) rp[i + 1] += rp[i];
    {
        std::vector<uint32_t> cur(rp, rp + n);
        if (!directed)
            for (int64_t i = 0; i < n; ++i) out->col[cur[i]++] = static_cast<uint32_t>(i);
        for (uint64_t key : edges) {
            const uint32_t a = relabel[key >> 32], b = relabel[key & 0xFFFFFFFFu];
            if (!directed) {
                out->col[cur[a]++] = b;
                out->col[cur[b]++] = a;
            } else if (flipped(key)) {
                out->col[cur[b]++] = a;
            } else {
                out->col[cur[a]++] = b;
            }
        }
    }
    edges = std::vector<uint64_t>();
    constexpr int64_t kRows = 4096;
    const uint64_t vseed = mix64(p->seed ^ 0x5EED);
    parallel_chunks((n + kRows - 1) / kRows, [&](int64_t c) {
        const int64_t b = c * kRows, e = std::min(n, b + kRows);
        for (int64_t i = b; i < e; ++i) {
            std::sort(out->col + rp[i], out->col + rp[i + 1]);
            const double di = rp[i + 1] - rp[i];
            for (uint32_t z = rp[i]; z < rp[i + 1]; ++z) {
                const uint32_t j = out->col[z];
                if (p->gcn_norm) {
                    out->vals[z] = static_cast<float>(1.0 / std::sqrt(di * std::max<double>(1.0, rp[j + 1] - rp[j])));
                } else {  // U(-1,1), symmetric in (i,j)
                    const uint64_t a = std::min<uint64_t>(i, j), bb = std::max<uint64_t>(i, j);
                    const uint64_t h = mix64(vseed ^ (a << 32 | bb));
                    out->vals[z] = static_cast<float>(static_cast<double>(h >> 11) * (2.0 / 9007199254740992.0) - 1.0);
                }
            }
        }
    });
    out->uni_nb = 0;
    for (int64_t i = 0; i < n; ++i) out->uni_nb += (rp[i + 1] - rp[i] == 1);
    out->c = 100;
    if (!directed) {  // symmetric with symmetric values and a self loop on every row, by construction
        out->n_edges_one_way = out->n_edges_asymmetric = 0;
        out->n_nodes_z_out = out->n_nodes_z_in = out->n_nodes_z_deg = 0;
        out->is_directed = 0;
        return FLEX_OK;
    }
    // one direction per pair: every edge is one-way; zero-degree counts from the two degree vectors
    out->n_edges_one_way = nnz;
    out->n_edges_asymmetric = 0;
    out->is_directed = nnz > 0;
    std::vector<uint8_t> has_in(static_cast<size_t>(n), 0);
    for (int64_t z = 0; z < nnz; ++z) has_in[out->col[z]] = 1;
    for (int64_t i = 0; i < n; ++i) {
        const bool z_out = rp[i + 1] == rp[i], z_in = !has_in[i];
        out->n_nodes_z_out += z_out;
        out->n_nodes_z_in += z_in;
        out->n_nodes_z_deg += (z_out && z_in);
    }
    return FLEX_OK;
} catch (const std::bad_alloc &) {
    return FLEX_ERR_NOMEM;
} catch (const std::length_error &) {
    return FLEX_ERR_NOMEM;
} catch (...) {  // nothing crosses the C ABI as an exception
    return FLEX_ERR_INVALID;
}
