// ingest.cpp -- CSV -> CSR ingest and the host-side dense operand, C ABI half.
//
// ≙ DataLoader::DataLoader (DataLoader.cu:9-124) and the cpuX fill of
// DataLoader::cuda_alloc_cpy (DataLoader.cu:198-209).  Same results as the
// reference (rowPtr/col/vals, uni_nb, symmetry and zero-degree statistics, the
// amazon.csv rule), but the file is parsed in one pass over a single buffer and
// the statistics come from a counting-sort transpose instead of the reference's
// vector<map<int,float>> (tens of GB at Amazon scale).
#include <algorithm>
#include <cctype>
#include <charconv>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <stdexcept>
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
    buf.resize(static_cast<size_t>(sz) + 2);
    const size_t got = std::fread(buf.data(), 1, static_cast<size_t>(sz), f);
    std::fclose(f);
    buf.resize(got + 2);
    buf[got] = '\n';      // the last line always ends
    buf[got + 1] = '\0';  // and strto* never runs off the buffer (callers treat size()-1 as the end)
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
    // std::stoi throws std::out_of_range past INT_MAX and the reference stores the result in an unsigned (DataLoader.cu:21-33):
    // a negative or >32-bit token must fail here, not wrap into a valid-looking index
    if (v < 0 || v > static_cast<long long>(UINT32_MAX)) return false;
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

int flex_csv_load(const char *path, flex_host_csr *out) try {
    if (!path || !out) return FLEX_ERR_INVALID;
    std::memset(out, 0, sizeof *out);
    std::vector<char> buf;
    if (!read_all(path, buf)) return FLEX_ERR_IO;
    std::string name(path);
    if (auto s = name.find_last_of('/'); s != std::string::npos) name = name.substr(s + 1);

    Line ln[3];
    const char *p = buf.data(), *end = buf.data() + buf.size() - 1;
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
} catch (const std::bad_alloc &) {
    return FLEX_ERR_NOMEM;
} catch (const std::length_error &) {
    return FLEX_ERR_NOMEM;
} catch (...) {  // nothing crosses the C ABI as an exception
    return FLEX_ERR_INVALID;
}

// -------------------------------------------------------------------------------------------
// MatrixMarket -> CSR, CSV writer, binary cache (data/SuiteSparse/mtx2csr.cc is the counterpart)

static int finish_host_csr(flex_host_csr *out, int64_t m, int64_t n, std::vector<uint32_t> &rp,
                           std::vector<uint32_t> &col, std::vector<float> &vals, const std::string &name) {
    if (m >= INT32_MAX || n >= INT32_MAX || col.size() >= (size_t(1) << 32)) return FLEX_ERR_UNSUPPORTED;
    out->m = static_cast<int32_t>(m);
    out->n = static_cast<int32_t>(n);
    out->nnz = static_cast<int64_t>(col.size());
    out->rowPtr = static_cast<uint32_t *>(std::malloc((static_cast<size_t>(m) + 1) * sizeof(uint32_t)));
    out->col = static_cast<uint32_t *>(std::malloc(std::max<size_t>(1, col.size()) * sizeof(uint32_t)));
    out->vals = static_cast<float *>(std::malloc(std::max<size_t>(1, col.size()) * sizeof(float)));
    if (!out->rowPtr || !out->col || !out->vals) {
        flex_host_csr_free(out);
        return FLEX_ERR_NOMEM;
    }
    std::copy(rp.begin(), rp.end(), out->rowPtr);
    std::copy(col.begin(), col.end(), out->col);
    std::copy(vals.begin(), vals.end(), out->vals);
    out->uni_nb = 0;
    for (int64_t r = 0; r < m; ++r) out->uni_nb += (rp[r + 1] - rp[r] == 1);
    out->c = classes_by_name(name);
    if (m != n) return FLEX_OK;  // the reference's statistics are defined for graphs (square) only
    const int rc = graph_statistics(out);
    // duplicate coordinates are legal in MatrixMarket files; they only make the statistics undefined
    return rc == FLEX_ERR_DUPLICATE ? FLEX_OK : rc;
}

int flex_mtx_load(const char *path, int sort_columns, flex_host_csr *out) try {
    if (!path || !out) return FLEX_ERR_INVALID;
    std::memset(out, 0, sizeof *out);
    std::vector<char> buf;
    if (!read_all(path, buf)) return FLEX_ERR_IO;
    const char *p = buf.data(), *end = buf.data() + buf.size() - 1;
    auto next_line = [&](Line &ln) {
        if (p >= end) return false;
        const char *q = static_cast<const char *>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
        if (!q) q = end;
        ln.b = p;
        ln.e = q;
        p = q + 1;
        return true;
    };
    Line ln;
    if (!next_line(ln)) return FLEX_ERR_FORMAT;
    std::string banner(ln.b, ln.e);
    for (auto &ch : banner) ch = static_cast<char>(std::tolower(static_cast<unsigned char>(ch)));
    if (banner.rfind("%%matrixmarket", 0) != 0 || banner.find("matrix") == std::string::npos ||
        banner.find("coordinate") == std::string::npos)
        return FLEX_ERR_FORMAT;  // mtx2csr.cc:72-79: only sparse coordinate matrices
    const bool pattern = banner.find("pattern") != std::string::npos;
    const bool cplx = banner.find("complex") != std::string::npos;
    const bool symmetric = banner.find(" symmetric") != std::string::npos || banner.find("hermitian") != std::string::npos;
    do {
        if (!next_line(ln)) return FLEX_ERR_FORMAT;
    } while (ln.b == ln.e || *ln.b == '%');
    long long m = 0, n = 0, nz = 0;
    {
        std::string sz(ln.b, ln.e);
        if (std::sscanf(sz.c_str(), "%lld %lld %lld", &m, &n, &nz) != 3 || m < 0 || n < 0 || nz < 0) return FLEX_ERR_FORMAT;
    }
    if (m >= INT32_MAX || n >= INT32_MAX || nz >= (1ll << 32)) return FLEX_ERR_UNSUPPORTED;
    // an entry line is at least "1 1\n": a header that promises more entries than the file can hold is refused
    // before anything is allocated for it
    if (nz > (end - p) / 4 + 1) return FLEX_ERR_FORMAT;
    if (symmetric && 2 * nz >= (1ll << 32)) return FLEX_ERR_UNSUPPORTED;
    std::vector<uint32_t> ri(static_cast<size_t>(nz)), ci(static_cast<size_t>(nz));
    std::vector<float> vv(static_cast<size_t>(nz));
    for (long long i = 0; i < nz; ++i) {
        do {
            if (!next_line(ln)) return FLEX_ERR_FORMAT;
        } while (ln.b == ln.e);
        char *q = nullptr;
        const long long r = std::strtoll(ln.b, &q, 10);
        const char *q2 = q;
        const long long c = std::strtoll(q2, &q, 10);
        // strto* skips newlines as white space: a number taken from beyond this line means the line is short
        if (q == q2 || q > ln.e || r < 1 || c < 1 || r > m || c > n) return FLEX_ERR_FORMAT;
        double v = 1.0;  // pattern files carry no value (mtx2csr.cc:133-137)
        if (!pattern) {
            const char *q3 = q;
            v = std::strtod(q3, &q);  // complex: the real part is kept, the imaginary one read and dropped
            if (q == q3 || q > ln.e) return FLEX_ERR_FORMAT;
            (void)cplx;
        }
        ri[i] = static_cast<uint32_t>(r - 1);
        ci[i] = static_cast<uint32_t>(c - 1);
        vv[i] = static_cast<float>(v);
    }
    // row counts, mirrored entries of symmetric files included (mtx2csr.cc:147-157)
    std::vector<uint32_t> rp(static_cast<size_t>(m) + 1, 0u);
    for (long long i = 0; i < nz; ++i) {
        ++rp[ri[i] + 1];
        if (symmetric && ri[i] != ci[i]) {
            if (ci[i] >= static_cast<uint32_t>(m) || ri[i] >= static_cast<uint32_t>(n)) return FLEX_ERR_FORMAT;
            ++rp[ci[i] + 1];
        }
    }
    for (long long r = 0; r < m; ++r) rp[r + 1] += rp[r];
    std::vector<uint32_t> col(rp[m]), cur(rp.begin(), rp.end() - 1);
    std::vector<float> vals(rp[m]);
    for (long long i = 0; i < nz; ++i) {  // file order inside each row, exactly as mtx2csr.cc:171-207
        col[cur[ri[i]]] = ci[i];
        vals[cur[ri[i]]++] = vv[i];
        if (symmetric && ri[i] != ci[i]) {
            col[cur[ci[i]]] = ri[i];
            vals[cur[ci[i]]++] = vv[i];
        }
    }
    if (sort_columns) {
        std::vector<std::pair<uint32_t, float>> row;
        for (long long r = 0; r < m; ++r) {
            row.clear();
            for (uint32_t e = rp[r]; e < rp[r + 1]; ++e) row.emplace_back(col[e], vals[e]);
            std::stable_sort(row.begin(), row.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
            for (uint32_t e = rp[r], i = 0; e < rp[r + 1]; ++e, ++i) {
                col[e] = row[i].first;
                vals[e] = row[i].second;
            }
        }
    }
    std::string name(path);
    if (auto s = name.find_last_of('/'); s != std::string::npos) name = name.substr(s + 1);
    return finish_host_csr(out, m, n, rp, col, vals, name);
} catch (const std::bad_alloc &) {
    return FLEX_ERR_NOMEM;
} catch (const std::length_error &) {
    return FLEX_ERR_NOMEM;
} catch (...) {  // nothing crosses the C ABI as an exception
    return FLEX_ERR_INVALID;
}

int flex_csv_save(const char *path, const flex_csr *A) try {
    if (!path) return FLEX_ERR_INVALID;
    int rc = flex::validate_csr(A);
    if (rc) return rc;
    FILE *f = std::fopen(path, "wb");
    if (!f) return FLEX_ERR_IO;
    for (int32_t i = 0; i <= A->m; ++i) std::fprintf(f, i < A->m ? "%u," : "%u", A->rowPtr[i]);
    std::fputc('\n', f);
    for (int64_t i = 0; i < A->nnz; ++i) std::fprintf(f, i + 1 < A->nnz ? "%u," : "%u", A->col[i]);
    std::fputc('\n', f);
    for (int64_t i = 0; i < A->nnz; ++i) std::fprintf(f, i + 1 < A->nnz ? "%.9g," : "%.9g", static_cast<double>(A->vals[i]));
    std::fputc('\n', f);
    return std::fclose(f) == 0 ? FLEX_OK : FLEX_ERR_IO;
} catch (const std::bad_alloc &) {
    return FLEX_ERR_NOMEM;
} catch (const std::length_error &) {
    return FLEX_ERR_NOMEM;
} catch (...) {  // nothing crosses the C ABI as an exception
    return FLEX_ERR_INVALID;
}

static const char kBinMagic[8] = {'F', 'L', 'E', 'X', 'C', 'S', 'R', '1'};

int flex_csr_save_bin(const char *path, const flex_csr *A) try {
    if (!path) return FLEX_ERR_INVALID;
    int rc = flex::validate_csr(A);
    if (rc) return rc;
    FILE *f = std::fopen(path, "wb");
    if (!f) return FLEX_ERR_IO;
    const int64_t hdr[3] = {A->m, A->n, A->nnz};
    bool ok = std::fwrite(kBinMagic, 1, 8, f) == 8 && std::fwrite(hdr, sizeof(int64_t), 3, f) == 3 &&
              std::fwrite(A->rowPtr, sizeof(uint32_t), static_cast<size_t>(A->m) + 1, f) == static_cast<size_t>(A->m) + 1 &&
              std::fwrite(A->col, sizeof(uint32_t), static_cast<size_t>(A->nnz), f) == static_cast<size_t>(A->nnz) &&
              std::fwrite(A->vals, sizeof(float), static_cast<size_t>(A->nnz), f) == static_cast<size_t>(A->nnz);
    ok = (std::fclose(f) == 0) && ok;
    return ok ? FLEX_OK : FLEX_ERR_IO;
} catch (const std::bad_alloc &) {
    return FLEX_ERR_NOMEM;
} catch (const std::length_error &) {
    return FLEX_ERR_NOMEM;
} catch (...) {  // nothing crosses the C ABI as an exception
    return FLEX_ERR_INVALID;
}

int flex_csr_load_bin(const char *path, flex_host_csr *out) try {
    if (!path || !out) return FLEX_ERR_INVALID;
    std::memset(out, 0, sizeof *out);
    FILE *f = std::fopen(path, "rb");
    if (!f) return FLEX_ERR_IO;
    char magic[8];
    int64_t hdr[3];
    if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, kBinMagic, 8) != 0 || std::fread(hdr, sizeof(int64_t), 3, f) != 3 ||
        hdr[0] < 0 || hdr[1] < 0 || hdr[2] < 0 || hdr[0] >= INT32_MAX || hdr[1] >= INT32_MAX || hdr[2] >= (int64_t(1) << 32)) {
        std::fclose(f);
        return FLEX_ERR_FORMAT;
    }
    // the header must describe exactly this file: nothing is allocated on the word of a corrupt header
    const long body = std::ftell(f);
    std::fseek(f, 0, SEEK_END);
    const long total = std::ftell(f);
    std::fseek(f, body, SEEK_SET);
    if (body < 0 || total < 0 || static_cast<int64_t>(total - body) != 4 * (hdr[0] + 1) + 8 * hdr[2]) {
        std::fclose(f);
        return FLEX_ERR_FORMAT;
    }
    std::vector<uint32_t> rp, col;
    std::vector<float> vals;
    try {
        rp.resize(static_cast<size_t>(hdr[0]) + 1);
        col.resize(static_cast<size_t>(hdr[2]));
        vals.resize(static_cast<size_t>(hdr[2]));
    } catch (...) {
        std::fclose(f);
        throw;
    }
    const bool ok = std::fread(rp.data(), sizeof(uint32_t), rp.size(), f) == rp.size() &&
                    std::fread(col.data(), sizeof(uint32_t), col.size(), f) == col.size() &&
                    std::fread(vals.data(), sizeof(float), vals.size(), f) == vals.size();
    std::fclose(f);
    if (!ok || rp[0] != 0 || rp.back() != col.size()) return FLEX_ERR_FORMAT;
    for (size_t r = 0; r + 1 < rp.size(); ++r)
        if (rp[r] > rp[r + 1]) return FLEX_ERR_FORMAT;
    for (uint32_t c : col)
        if (c >= static_cast<uint32_t>(hdr[1])) return FLEX_ERR_FORMAT;
    std::string name(path);
    if (auto s = name.find_last_of('/'); s != std::string::npos) name = name.substr(s + 1);
    return finish_host_csr(out, hdr[0], hdr[1], rp, col, vals, name);
} catch (const std::bad_alloc &) {
    return FLEX_ERR_NOMEM;
} catch (const std::length_error &) {
    return FLEX_ERR_NOMEM;
} catch (...) {  // nothing crosses the C ABI as an exception
    return FLEX_ERR_INVALID;
}

int flex_fill_dense_rand(float *hostB, int64_t n, int k) {
    if (!hostB || n < 0 || k <= 0) return FLEX_ERR_INVALID;
    for (int64_t i = 0; i < n * k; ++i)
        hostB[i] = 2 * static_cast<float>(std::rand()) / static_cast<float>(RAND_MAX) - 1.0f;
    return FLEX_OK;
}

static const char kPermMagic[8] = {'F', 'L', 'E', 'X', 'P', 'R', 'M', '1'};

uint64_t flex_csr_fingerprint(const flex_csr *A) {
    if (flex::validate_csr(A)) return 0;
    // FNV-1a over 64-bit words of (m, n, nnz, rowPtr, col): structure only, values do not change an ordering
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
    mix(static_cast<uint64_t>(A->m));
    mix(static_cast<uint64_t>(A->n));
    mix(static_cast<uint64_t>(A->nnz));
    for (int64_t i = 0; i <= A->m; ++i) mix(A->rowPtr[i]);
    for (int64_t e = 0; e < A->nnz; ++e) mix(A->col[e]);
    return h ? h : 1;
}

int flex_perm_save(const char *path, const uint32_t *rank, int64_t n, uint64_t fingerprint) {
    if (!path || !rank || n < 0) return FLEX_ERR_INVALID;
    FILE *f = std::fopen(path, "wb");
    if (!f) return FLEX_ERR_IO;
    const uint64_t hdr[2] = {static_cast<uint64_t>(n), fingerprint};
    bool ok = std::fwrite(kPermMagic, 1, 8, f) == 8 && std::fwrite(hdr, sizeof(uint64_t), 2, f) == 2 &&
              std::fwrite(rank, sizeof(uint32_t), static_cast<size_t>(n), f) == static_cast<size_t>(n);
    ok = (std::fclose(f) == 0) && ok;
    return ok ? FLEX_OK : FLEX_ERR_IO;
}

int flex_perm_load(const char *path, uint32_t *rank, int64_t n, uint64_t fingerprint) try {
    if (!path || !rank || n < 0) return FLEX_ERR_INVALID;
    FILE *f = std::fopen(path, "rb");
    if (!f) return FLEX_ERR_IO;
    char magic[8];
    uint64_t hdr[2];
    bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, kPermMagic, 8) == 0 &&
              std::fread(hdr, sizeof(uint64_t), 2, f) == 2 && hdr[0] == static_cast<uint64_t>(n) && hdr[1] == fingerprint &&
              std::fread(rank, sizeof(uint32_t), static_cast<size_t>(n), f) == static_cast<size_t>(n) && std::fgetc(f) == EOF;
    std::fclose(f);
    if (!ok) return FLEX_ERR_FORMAT;
    try {
        std::vector<uint8_t> seen(static_cast<size_t>(n), 0);
        for (int64_t i = 0; i < n; ++i) {
            if (rank[i] >= static_cast<uint64_t>(n) || seen[rank[i]]) return FLEX_ERR_FORMAT;
            seen[rank[i]] = 1;
        }
    } catch (const std::bad_alloc &) {
        return FLEX_ERR_NOMEM;
    }
    return FLEX_OK;
} catch (const std::bad_alloc &) {
    return FLEX_ERR_NOMEM;
} catch (const std::length_error &) {
    return FLEX_ERR_NOMEM;
} catch (...) {  // nothing crosses the C ABI as an exception
    return FLEX_ERR_INVALID;
}

}  // extern "C"
