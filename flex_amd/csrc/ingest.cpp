// ingest.cpp -- CSV -> CSR ingest and the host-side dense operand, C ABI half.
//
// ≙ DataLoader::DataLoader (DataLoader.cu:9-124) and the cpuX fill of
// DataLoader::cuda_alloc_cpy (DataLoader.cu:198-209).  Same results as the
// reference (rowPtr/col/vals, uni_nb, symmetry and zero-degree statistics, the
// amazon.csv rule), but the file is parsed in one pass over a single buffer and
// the statistics come from a counting-sort transpose instead of the reference's
// vector<map<int,float>> (tens of GB at Amazon scale).
#include <algorithm>
#include <charconv>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "internal.h"

namespace {

struct Line {
    const char *b = nullptr, *e = nullptr;
};

bool read_all(const char *path, std::vector<char> &buf) {
    FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (sz < 0) { std::fclose(f); return false; }
    buf.resize(static_cast<size_t>(sz) + 1);
    const size_t got = std::fread(buf.data(), 1, static_cast<size_t>(sz), f);
    std::fclose(f);
    buf.resize(got + 1);
    buf[got] = '\n';
    return true;
}

// Tokens as `while(getline(ss, word, ','))` yields them: a trailing comma adds none.
template <typename F>
bool for_each_token(Line ln, F &&fn) {
    const char *p = ln.b;
    while (p < ln.e) {
        const char *q = static_cast<const char *>(std::memchr(p, ',', static_cast<size_t>(ln.e - p)));
        if (!q) q = ln.e;
        if (!fn(p, q)) return false;
        p = q + 1;
    }
    return true;
}

bool parse_u32(const char *b, const char *e, uint32_t &out) {  // std::stoi semantics: leading blanks, sign
    while (b < e && (*b == ' ' || *b == '\t')) ++b;
    if (b < e && *b == '+') ++b;
    long long v = 0;
    auto r = std::from_chars(b, e, v);
    if (r.ec != std::errc() || r.ptr == b) return false;
    out = static_cast<uint32_t>(v);
    return true;
}

int classes_by_name(const std::string &name) {  // DataLoader.cu:62-84
    static const std::pair<const char *, int> table[] = {
        {"polblogs.csv", 2}, {"cora.csv", 7},   {"citeseer.csv", 6}, {"pubmed.csv", 3}, {"ppi.csv", 121},
        {"reddit.csv", 41},  {"flickr.csv", 7}, {"yelp.csv", 100},   {"amazon.csv", 107}};
    for (auto &t : table)
        if (name == t.first) return t.second;
    return 100;
}

// is_directed, n_edges_one_way, n_edges_asymmetric, zero-degree counts (DataLoader.cu:86-115)
int graph_statistics(flex_host_csr *a) {
    const int64_t m = a->m, nnz = a->nnz;
    std::vector<uint32_t> tptr(static_cast<size_t>(m) + 1, 0), tsrc(static_cast<size_t>(nnz));
    std::vector<float> tval(static_cast<size_t>(nnz));
    for (int64_t e = 0; e < nnz; ++e) {
        if (a->col[e] >= static_cast<uint32_t>(m)) return FLEX_ERR_INVALID;
        ++tptr[a->col[e] + 1];
    }
    for (int64_t r = 0; r < m; ++r) tptr[r + 1] += tptr[r];
    {
        std::vector<uint32_t> cur(tptr.begin(), tptr.end() - 1);
        for (int64_t r = 0; r < m; ++r)
            for (uint32_t e = a->rowPtr[r]; e < a->rowPtr[r + 1]; ++e) {
                const uint32_t pos = cur[a->col[e]]++;
                tsrc[pos] = static_cast<uint32_t>(r);  // sources arrive in ascending order
                tval[pos] = a->vals[e];
            }
    }
    for (int64_t d = 0; d < m; ++d)
        if (std::adjacent_find(tsrc.begin() + tptr[d], tsrc.begin() + tptr[d + 1]) != tsrc.begin() + tptr[d + 1])
            return FLEX_ERR_DUPLICATE;
    int64_t one_way = 0, asym = 0;
    for (int64_t r = 0; r < m; ++r) {
        const auto in_b = tsrc.begin() + tptr[r], in_e = tsrc.begin() + tptr[r + 1];
        for (uint32_t e = a->rowPtr[r]; e < a->rowPtr[r + 1]; ++e) {
            const auto it = std::lower_bound(in_b, in_e, a->col[e]);  // reverse edge col[e] -> r ?
            if (it == in_e || *it != a->col[e]) ++one_way;
            else if (tval[static_cast<size_t>(it - tsrc.begin())] != a->vals[e]) ++asym;
        }
    }
    a->n_edges_one_way = one_way;
    a->n_edges_asymmetric = asym;
    a->n_nodes_z_out = a->n_nodes_z_in = a->n_nodes_z_deg = 0;
    for (int64_t r = 0; r < m; ++r) {
        const bool z_out = a->rowPtr[r] == a->rowPtr[r + 1];
        const bool z_in = tptr[r] == tptr[r + 1];
        a->n_nodes_z_out += z_out;
        a->n_nodes_z_in += z_in;
        a->n_nodes_z_deg += (z_in && z_out);
    }
    a->is_directed = one_way != 0;
    return FLEX_OK;
}

}  // namespace

extern "C" {

void flex_host_csr_free(flex_host_csr *a) {
    if (!a) return;
    std::free(a->rowPtr);
    std::free(a->col);
    std::free(a->vals);
    std::memset(a, 0, sizeof *a);
}

int flex_csv_load(const char *path, flex_host_csr *out) {
    if (!path || !out) return FLEX_ERR_INVALID;
    std::memset(out, 0, sizeof *out);
    std::vector<char> buf;
    if (!read_all(path, buf)) return FLEX_ERR_IO;
    std::string name(path);
    if (auto s = name.find_last_of('/'); s != std::string::npos) name = name.substr(s + 1);

    Line ln[3];
    const char *p = buf.data(), *end = buf.data() + buf.size();
    for (int i = 0; i < 3 && p < end; ++i) {
        const char *q = static_cast<const char *>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
        ln[i].b = p;
        ln[i].e = q;
        while (ln[i].e > ln[i].b && ln[i].e[-1] == '\r') --ln[i].e;
        p = q + 1;
    }
    if (!ln[0].b || !ln[1].b) return FLEX_ERR_FORMAT;

    std::vector<uint32_t> rp, col;
    std::vector<float> vals;
    bool ok = for_each_token(ln[0], [&](const char *b, const char *e) {
        uint32_t v;
        if (!parse_u32(b, e, v)) return false;
        rp.push_back(v);
        return true;
    });
    ok = ok && for_each_token(ln[1], [&](const char *b, const char *e) {
        uint32_t v;
        if (!parse_u32(b, e, v)) return false;
        col.push_back(v);
        return true;
    });
    if (!ok || rp.empty()) return FLEX_ERR_FORMAT;

    if (name == "amazon.csv") {  // no value line: DataLoader.cu:36-46
        vals.resize(col.size());
        for (auto &v : vals) v = 2 * static_cast<float>(std::rand()) / static_cast<float>(RAND_MAX) - 1.0f;
    } else {
        if (!ln[2].b) return FLEX_ERR_FORMAT;
        vals.reserve(col.size());
        ok = for_each_token(ln[2], [&](const char *b, const char *e) {  // std::stof == strtof
            std::string tok(b, e);
            char *endp = nullptr;
            const float v = std::strtof(tok.c_str(), &endp);
            if (endp == tok.c_str()) return false;
            vals.push_back(v);
            return true;
        });
        if (!ok || vals.size() != col.size()) return FLEX_ERR_FORMAT;  // assert(col.size()==vals.size())
    }
    const size_t m = rp.size() - 1;
    if (m >= (size_t(1) << 31) || col.size() >= (size_t(1) << 32)) return FLEX_ERR_UNSUPPORTED;
    if (rp[0] != 0 || rp[m] != col.size()) return FLEX_ERR_FORMAT;
    for (size_t r = 0; r < m; ++r)
        if (rp[r] > rp[r + 1]) return FLEX_ERR_FORMAT;

    out->m = out->n = static_cast<int32_t>(m);
    out->nnz = static_cast<int64_t>(col.size());
    out->rowPtr = static_cast<uint32_t *>(std::malloc((m + 1) * sizeof(uint32_t)));
    out->col = static_cast<uint32_t *>(std::malloc(std::max<size_t>(1, col.size()) * sizeof(uint32_t)));
    out->vals = static_cast<float *>(std::malloc(std::max<size_t>(1, col.size()) * sizeof(float)));
    if (!out->rowPtr || !out->col || !out->vals) {
        flex_host_csr_free(out);
        return FLEX_ERR_NOMEM;
    }
    std::copy(rp.begin(), rp.end(), out->rowPtr);
    std::copy(col.begin(), col.end(), out->col);
    std::copy(vals.begin(), vals.end(), out->vals);
    out->uni_nb = 0;  // rows with exactly one nonzero, DataLoader.cu:26-29
    for (size_t r = 0; r < m; ++r) out->uni_nb += (rp[r + 1] - rp[r] == 1);
    out->c = classes_by_name(name);
    const int rc = graph_statistics(out);
    if (rc) flex_host_csr_free(out);
    return rc;
}

int flex_fill_dense_rand(float *hostB, int64_t n, int k) {
    if (!hostB || n < 0 || k <= 0) return FLEX_ERR_INVALID;
    for (int64_t i = 0; i < n * k; ++i)
        hostB[i] = 2 * static_cast<float>(std::rand()) / static_cast<float>(RAND_MAX) - 1.0f;
    return FLEX_OK;
}

}  // extern "C"
