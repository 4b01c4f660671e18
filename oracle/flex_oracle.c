/*
 * flex_oracle.c -- CPU restatement of the reference's SpMM hot path.
 * TEST INFRASTRUCTURE ONLY (see flex_oracle.h).  Build: oracle/Makefile
 * (gcc -O3 -ffp-contract=off: the reference's CPU loop is host code of a .cu
 * compiled by g++ -O3 without -march, i.e. separately rounded mul and add).
 */
#define _GNU_SOURCE
#include "flex_oracle.h"
#include <ctype.h>
#include <errno.h>
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ ingest */

static char *read_file(const char *path, size_t *len) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)sz + 2);
    if (!buf) { fclose(f); return NULL; }
    size_t got = fread(buf, 1, (size_t)sz, f);
    fclose(f);
    buf[got] = '\n';
    buf[got + 1] = 0;
    *len = got;
    return buf;
}

/* Count comma-separated tokens of one line the way
 * `while(getline(ss,word,','))` does (DataLoader.cu:21-23): a trailing comma
 * produces no extra token. */
static size_t count_tokens(const char *b, const char *e) {
    if (b == e) return 0;
    size_t n = 1;
    for (const char *p = b; p < e; ++p)
        if (*p == ',') ++n;
    if (e[-1] == ',') --n;
    return n;
}

static const char *basename_of(const char *path) {
    const char *s = strrchr(path, '/');
    return s ? s + 1 : path;
}

/* classes by file name, DataLoader.cu:62-84 */
static int classes_for(const char *name) {
    static const struct { const char *n; int c; } tab[] = {
        {"polblogs.csv", 2}, {"cora.csv", 7},   {"citeseer.csv", 6}, {"pubmed.csv", 3},
        {"ppi.csv", 121},    {"reddit.csv", 41}, {"flickr.csv", 7},  {"yelp.csv", 100},
        {"amazon.csv", 107}};
    for (size_t i = 0; i < sizeof tab / sizeof tab[0]; ++i)
        if (!strcmp(name, tab[i].n)) return tab[i].c;
    return 100;
}

static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}

/* graph statistics, DataLoader.cu:86-115, without the vector<map> (same results:
 * e_inv[dst] is the set of sources of dst, i.e. row dst of the transpose). */
static int graph_stats(oracle_csr *a) {
    const int64_t m = a->m, nnz = a->nnz;
    uint32_t *tp = (uint32_t *)calloc((size_t)m + 1, sizeof(uint32_t));
    uint32_t *tsrc = (uint32_t *)malloc((size_t)(nnz ? nnz : 1) * sizeof(uint32_t));
    float *tval = (float *)malloc((size_t)(nnz ? nnz : 1) * sizeof(float));
    uint32_t *fill = (uint32_t *)calloc((size_t)m + 1, sizeof(uint32_t));
    if (!tp || !tsrc || !tval || !fill) return -ENOMEM;
    for (int64_t e = 0; e < nnz; ++e) {
        if (a->col[e] >= (uint32_t)m) return -EDOM;
        tp[a->col[e] + 1]++;
    }
    for (int64_t r = 0; r < m; ++r) tp[r + 1] += tp[r];
    for (int64_t r = 0; r < m; ++r)
        for (uint32_t e = a->rowPtr[r]; e < a->rowPtr[r + 1]; ++e) {
            uint32_t d = a->col[e];
            uint32_t pos = tp[d] + fill[d]++;
            tsrc[pos] = (uint32_t)r; /* ascending r within each d by construction */
            tval[pos] = a->vals[e];
        }
    /* assert( e_inv[dst].count(r) == 0 ): duplicate (r,dst) */
    for (int64_t d = 0; d < m; ++d)
        for (uint32_t p = tp[d] + 1; p < tp[d + 1]; ++p)
            if (tsrc[p] == tsrc[p - 1]) return -EEXIST;
    a->n_edges_one_way = a->n_edges_asymmetric = 0;
    for (int64_t r = 0; r < m; ++r)
        for (uint32_t e = a->rowPtr[r]; e < a->rowPtr[r + 1]; ++e) {
            /* e_inv[r].count(col[e]): is there an edge col[e] -> r */
            uint32_t key = a->col[e];
            uint32_t lo = tp[r], hi = tp[r + 1];
            uint32_t *hit = (uint32_t *)bsearch(&key, tsrc + lo, hi - lo, sizeof(uint32_t), cmp_u32);
            if (!hit) a->n_edges_one_way++;
            else if (tval[hit - tsrc] != a->vals[e]) a->n_edges_asymmetric++;
        }
    a->n_nodes_z_out = a->n_nodes_z_in = a->n_nodes_z_deg = 0;
    for (int64_t r = 0; r < m; ++r) {
        int z_out = a->rowPtr[r] == a->rowPtr[r + 1];
        int z_in = tp[r] == tp[r + 1];
        a->n_nodes_z_out += z_out;
        a->n_nodes_z_in += z_in;
        a->n_nodes_z_deg += (z_in && z_out);
    }
    a->is_directed = a->n_edges_one_way != 0;
    free(tp); free(tsrc); free(tval); free(fill);
    return 0;
}

int oracle_csv_load(const char *path, oracle_csr *out, int rand_state_reset) {
    memset(out, 0, sizeof *out);
    if (rand_state_reset) srand(1);
    size_t len = 0;
    char *buf = read_file(path, &len);
    if (!buf) return -ENOENT;
    const char *name = basename_of(path);
    /* three lines: rowPtr, col, vals (DataLoader.cu:19-54) */
    const char *lb[3] = {0, 0, 0}, *le[3] = {0, 0, 0};
    const char *p = buf, *end = buf + len;
    for (int i = 0; i < 3 && p <= end; ++i) {
        const char *q = memchr(p, '\n', (size_t)(end - p) + 1);
        lb[i] = p;
        le[i] = q;
        while (le[i] > lb[i] && (le[i][-1] == '\r')) --le[i];
        p = q + 1;
        if (p > end) break;
    }
    if (!lb[0] || !lb[1]) { free(buf); return -EINVAL; }
    size_t n_rp = count_tokens(lb[0], le[0]);
    size_t n_col = count_tokens(lb[1], le[1]);
    if (n_rp < 1) { free(buf); return -EINVAL; }
    out->rowPtr = (uint32_t *)malloc(n_rp * sizeof(uint32_t));
    out->col = (uint32_t *)malloc((n_col ? n_col : 1) * sizeof(uint32_t));
    out->vals = (float *)malloc((n_col ? n_col : 1) * sizeof(float));
    if (!out->rowPtr || !out->col || !out->vals) { free(buf); return -ENOMEM; }
    char *q;
    p = lb[0];
    for (size_t i = 0; i < n_rp; ++i) { /* std::stoi */
        out->rowPtr[i] = (uint32_t)strtol(p, &q, 10);
        if (q == p) { free(buf); return -EINVAL; }
        p = q + 1;
    }
    out->uni_nb = 0; /* DataLoader.cu:26-29 */
    for (size_t i = 1; i < n_rp; ++i)
        if (out->rowPtr[i] - out->rowPtr[i - 1] == 1) out->uni_nb++;
    p = lb[1];
    for (size_t i = 0; i < n_col; ++i) {
        out->col[i] = (uint32_t)strtol(p, &q, 10);
        if (q == p) { free(buf); return -EINVAL; }
        p = q + 1;
    }
    if (!strcmp(name, "amazon.csv")) { /* DataLoader.cu:36-46 */
        for (size_t i = 0; i < n_col; ++i)
            out->vals[i] = 2 * (float)rand() / (float)RAND_MAX - 1.0f;
    } else {
        size_t n_val = lb[2] ? count_tokens(lb[2], le[2]) : 0;
        if (n_val != n_col) { free(buf); return -ERANGE; } /* assert(col.size()==vals.size()) */
        p = lb[2];
        for (size_t i = 0; i < n_val; ++i) { /* std::stof */
            out->vals[i] = strtof(p, &q);
            if (q == p) { free(buf); return -EINVAL; }
            p = q + 1;
        }
    }
    free(buf);
    out->m = out->n = (int64_t)n_rp - 1;
    out->nnz = (int64_t)n_col;
    if (out->rowPtr[out->m] != (uint32_t)out->nnz) return -ERANGE;
    for (int64_t r = 0; r < out->m; ++r)
        if (out->rowPtr[r] > out->rowPtr[r + 1]) return -ERANGE;
    out->c = classes_for(name);
    return graph_stats(out);
}

void oracle_csr_free(oracle_csr *a) {
    free(a->rowPtr); free(a->col); free(a->vals);
    memset(a, 0, sizeof *a);
}

/* ------------------------------------------------------------- B generator */

void oracle_gen_B(int64_t n, int k, float *B, int rand_state_reset) {
    if (rand_state_reset) srand(1);
    for (int64_t i = 0; i < n; ++i)
        for (int j = 0; j < k; ++j)
            B[i * k + j] = 2 * (float)rand() / (float)RAND_MAX - 1.0f;
}

/* -------------------------------------------------------------------- SpMM */

static void spmm_rows(int64_t r0, int64_t r1, const uint32_t *rowPtr, const uint32_t *col,
                      const float *vals, const float *B, float *C, int k) {
    for (int64_t r = r0; r < r1; ++r) {
        float *c = C + r * k;
        for (int j = 0; j < k; ++j) c[j] = 0.0f;
        for (uint32_t e = rowPtr[r]; e < rowPtr[r + 1]; ++e) {
            const float *b = B + (int64_t)k * col[e];
            const float v = vals[e];
            for (int j = 0; j < k; ++j) c[j] += b[j] * v;
        }
    }
}

void oracle_spmm(int64_t m, const uint32_t *rowPtr, const uint32_t *col, const float *vals,
                 const float *B, float *C, int k) {
    spmm_rows(0, m, rowPtr, col, vals, B, C, k);
}

typedef struct {
    int64_t r0, r1;
    const uint32_t *rowPtr, *col;
    const float *vals, *B;
    float *C;
    int k;
} mt_arg;

static void *mt_run(void *p) {
    mt_arg *a = (mt_arg *)p;
    spmm_rows(a->r0, a->r1, a->rowPtr, a->col, a->vals, a->B, a->C, a->k);
    return NULL;
}

void oracle_spmm_mt(int64_t m, const uint32_t *rowPtr, const uint32_t *col, const float *vals,
                    const float *B, float *C, int k, int nthreads) {
    if (nthreads <= 1 || m < nthreads) { oracle_spmm(m, rowPtr, col, vals, B, C, k); return; }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    mt_arg *args = (mt_arg *)malloc(sizeof(mt_arg) * (size_t)nthreads);
    /* nnz-balanced contiguous row chunks (cost = nnz + rows) */
    const double total = (double)rowPtr[m] + (double)m;
    int64_t r = 0;
    for (int t = 0; t < nthreads; ++t) {
        int64_t r0 = r;
        const double want = total * (t + 1) / nthreads;
        while (r < m && (double)rowPtr[r + 1] + (double)(r + 1) <= want) ++r;
        if (t == nthreads - 1) r = m;
        args[t] = (mt_arg){r0, r, rowPtr, col, vals, B, C, k};
        pthread_create(&th[t], NULL, mt_run, &args[t]);
    }
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    free(th); free(args);
}

/* ---------------------------------------------------------------- resCheck */

int64_t oracle_rescheck(const float *gold, const float *res, const uint32_t *orig_rowPtr,
                        int64_t m, int k, double *max_err_out, int32_t *max_err_row_nnz,
                        int64_t *gold_zeros) {
    int64_t count = 0, nz = 0;
    double max_err = 0;
    int32_t me_nnz = 0;
    for (int64_t r = 0; r < m; ++r) {
        const int row_nnz = (int)(orig_rowPtr[r + 1] - orig_rowPtr[r]);
        const double tol = (double)FLT_EPSILON * row_nnz * 4; /* flex.cu:4172 */
        for (int c = 0; c < k; ++c) {
            const int64_t idx = r * k + c;
            if (gold[idx] == 0) nz++;
            const double err = fabsf(gold[idx]) < 1
                                   ? fabs((double)gold[idx] - res[idx])
                                   : fabs(1.0 - (double)res[idx] / gold[idx]);
            if (err > max_err) { max_err = err; me_nnz = row_nnz; }
            if (err > tol) count++;
            /* NaN never compares > tol in the reference either; flag it here so a
             * NaN-producing kernel cannot slip through the checker. */
            if (err != err) count++;
        }
    }
    if (max_err_out) *max_err_out = max_err;
    if (max_err_row_nnz) *max_err_row_nnz = me_nnz;
    if (gold_zeros) *gold_zeros = nz;
    return count;
}

/* --------------------------------------------------------------------- RCM */

typedef struct { uint64_t key, val; } keyval;

static int cmp_deg_asc(const void *a, const void *b) { /* order_deg.cu:9-11 */
    const keyval *x = (const keyval *)a, *y = (const keyval *)b;
    if (x->val != y->val) return x->val < y->val ? -1 : 1;
    return (x->key > y->key) - (x->key < y->key);
}

static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}

int oracle_order_rcm(int64_t n, const uint32_t *rowPtr, const uint32_t *col, uint64_t *rank) {
    const int64_t e = rowPtr[n];
    /* Edgelist(dl): edges (i, col[j]) in CSR order, edgelist.cu:23-32.
     * compute_degrees: deg = degIn + degOut, edgelist.cu:86-103. */
    uint64_t *degOut = (uint64_t *)calloc((size_t)n + 1, sizeof(uint64_t));
    uint64_t *degIn = (uint64_t *)calloc((size_t)n + 1, sizeof(uint64_t));
    keyval *ts = (keyval *)malloc(sizeof(keyval) * (size_t)(n ? n : 1));
    uint64_t *rank_deg = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n ? n : 1));
    uint64_t *cd = (uint64_t *)calloc((size_t)n + 1, sizeof(uint64_t));
    uint64_t *fill = (uint64_t *)calloc((size_t)n + 1, sizeof(uint64_t));
    uint64_t *adj = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(e ? e : 1));
    uint64_t *order = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n ? n : 1));
    uint8_t *placed = (uint8_t *)calloc((size_t)n + 1, 1);
    if (!degOut || !degIn || !ts || !rank_deg || !cd || !fill || !adj || !order || !placed)
        return -ENOMEM;
    for (int64_t i = 0; i < n; ++i)
        for (uint32_t j = rowPtr[i]; j < rowPtr[i + 1]; ++j) {
            degOut[i]++;
            degIn[col[j]]++;
        }
    /* order_deg(h,false): rank by (deg ASC, node ASC), order_deg.cu:19-45 */
    for (int64_t u = 0; u < n; ++u) ts[u] = (keyval){(uint64_t)u, degIn[u] + degOut[u]};
    qsort(ts, (size_t)n, sizeof(keyval), cmp_deg_asc);
    for (int64_t u = 0; u < n; ++u) rank_deg[ts[u].key] = (uint64_t)u;
    /* Dadjlist g(h, rank_deg): out-adjacency in relabelled ids, neighbours sorted,
     * adjlist.cu:62-73 (build_from_edgelist_ranked -> sorted=true), 127-150 */
    for (int64_t u = 0; u < n; ++u) cd[rank_deg[u] + 1] = degOut[u];
    for (int64_t u = 0; u < n; ++u) cd[u + 1] += cd[u];
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t u = rank_deg[i];
        for (uint32_t j = rowPtr[i]; j < rowPtr[i + 1]; ++j) adj[cd[u] + fill[u]++] = rank_deg[col[j]];
    }
    for (int64_t u = 0; u < n; ++u)
        qsort(adj + cd[u], (size_t)(cd[u + 1] - cd[u]), sizeof(uint64_t), cmp_u64);
    /* algo_bfs(g, 0): algo_bfs.cu:11-39 */
    uint64_t head = 0, tail = 0;
    for (int64_t c = 0; c < n; ++c) {
        if (placed[c]) continue;
        order[tail++] = (uint64_t)c;
        placed[c] = 1;
        while (head < tail) {
            const uint64_t w = order[head++];
            for (uint64_t a = cd[w]; a < cd[w + 1]; ++a) {
                const uint64_t v = adj[a];
                if (placed[v]) continue;
                placed[v] = 1;
                order[tail++] = v;
            }
        }
    }
    /* rank_from_order (tools.cu:31-43) then reverse + compose (order_rcm.cu:28-31);
     * degOut is reused as rank_bfs */
    uint64_t *rank_bfs = degOut;
    for (int64_t i = 0; i < n; ++i) rank_bfs[order[i]] = (uint64_t)i;
    for (int64_t u = 0; u < n; ++u) rank[u] = (uint64_t)n - 1 - rank_bfs[rank_deg[u]];
    free(degOut); free(degIn); free(ts); free(rank_deg); free(cd); free(fill);
    free(adj); free(order); free(placed);
    return 0;
}

typedef struct { uint32_t dst; float val; } dstval;
static int cmp_dst(const void *a, const void *b) {
    const dstval *x = (const dstval *)a, *y = (const dstval *)b;
    return (x->dst > y->dst) - (x->dst < y->dst);
}

void oracle_perm_csr(int64_t n, const uint32_t *rowPtr, const uint32_t *col, const float *vals,
                     const uint64_t *rank, int32_t *vo_mp, uint32_t *rowPtr2, uint32_t *col2,
                     float *vals2) {
    for (int64_t u = 0; u < n; ++u) vo_mp[rank[u]] = (int32_t)u; /* DataLoader.cu:747-750 */
    rowPtr2[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t v = vo_mp[i];
        rowPtr2[i + 1] = rowPtr2[i] + rowPtr[v + 1] - rowPtr[v];
    }
    uint32_t maxd = 0;
    for (int64_t i = 0; i < n; ++i)
        if (rowPtr[i + 1] - rowPtr[i] > maxd) maxd = rowPtr[i + 1] - rowPtr[i];
    dstval *perm = (dstval *)malloc(sizeof(dstval) * (size_t)(maxd ? maxd : 1));
    for (int64_t s = 0; s < n; ++s) { /* DataLoader.cu:758-779 */
        const uint64_t s2 = rank[s];
        const uint32_t d = rowPtr[s + 1] - rowPtr[s];
        for (uint32_t i = 0; i < d; ++i)
            perm[i] = (dstval){(uint32_t)rank[col[rowPtr[s] + i]], vals[rowPtr[s] + i]};
        qsort(perm, d, sizeof(dstval), cmp_dst);
        for (uint32_t i = 0; i < d; ++i) {
            col2[rowPtr2[s2] + i] = perm[i].dst;
            vals2[rowPtr2[s2] + i] = perm[i].val;
        }
    }
    free(perm);
}

/* ------------------------------------------------------------------ Gorder */
/* complete_gorder(h, window): order_gorder.cu:13-31 = RCM, Badjlist in RCM ids
 * (adjlist.cu:156-185), order_gorder (order_gorder.cu:35-84), move_window
 * (order_gorder.cu:88-143) over the UnitHeap of unitheap.cu.  Restated literally,
 * quirks included (ReConstruct links indices 0..heapsize-1, unitheap.cu:35-38). */

typedef struct { int key; uint64_t prev, next; } uh_elem;
typedef struct { uint64_t first, second; } uh_head;
typedef struct {
    int *update;
    uh_elem *ll;
    uh_head *hd;
    size_t hd_size, n;
    size_t heapsize;
    uint64_t top, huge, none;
    int infty;
    int failed;
} unitheap;

static void uh_header_resize(unitheap *h, size_t sz) {
    if (sz <= h->hd_size) return;
    h->hd = (uh_head *)realloc(h->hd, sz * sizeof(uh_head));
    for (size_t i = h->hd_size; i < sz; ++i) h->hd[i].first = h->hd[i].second = h->none;
    h->hd_size = sz;
}

static void uh_init(unitheap *h, uint64_t size) { /* unitheap.cu:17-23 */
    memset(h, 0, sizeof *h);
    h->infty = 0x7fffffff / 2;
    h->n = size;
    h->none = size + 2;
    h->huge = (uint64_t)sqrt((double)size);
    h->ll = (uh_elem *)malloc(sizeof(uh_elem) * (size ? size : 1));
    h->update = (int *)malloc(sizeof(int) * (size ? size : 1));
    for (uint64_t i = 0; i < size; ++i) {
        h->ll[i].key = h->infty;
        h->ll[i].prev = h->ll[i].next = h->none;
        h->update[i] = h->infty;
    }
}

static void uh_free(unitheap *h) { free(h->ll); free(h->update); free(h->hd); }

static unitheap *g_sort_heap;
static int uh_cmp_desc(const void *a, const void *b) { /* unitheap.cu:40-42 */
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    int kx = g_sort_heap->ll[x].key, ky = g_sort_heap->ll[y].key;
    if (kx != ky) return kx > ky ? -1 : 1;
    return (x > y) - (x < y);
}

static void uh_reconstruct(unitheap *h) { /* unitheap.cu:34-70 */
    size_t hs = h->heapsize;
    uint64_t *g = (uint64_t *)malloc(sizeof(uint64_t) * (hs ? hs : 1));
    for (size_t i = 0; i < hs; ++i) g[i] = i;
    g_sort_heap = h;
    qsort(g, hs, sizeof(uint64_t), uh_cmp_desc);
    h->top = g[0];
    int cur = h->ll[h->top].key;
    uh_header_resize(h, (size_t)10 * cur + 1);
    h->hd[cur].first = h->top;
    for (size_t i = 0; i < hs; ++i) {
        uint64_t v = g[i];
        h->ll[v].prev = i > 0 ? g[i - 1] : h->none;
        h->ll[v].next = i + 1 < hs ? g[i + 1] : h->none;
        int key = h->ll[v].key;
        if (key != cur) {
            h->hd[cur].second = g[i - 1];
            h->hd[key].first = g[i];
            cur = key;
        }
    }
    h->hd[cur].second = g[hs - 1];
    free(g);
}

static void uh_erase_key(unitheap *h, uint64_t idx, uint64_t next, uint64_t prev) { /* unitheap.cu:76-84 */
    int key = h->ll[idx].key;
    if (h->hd[key].first == h->hd[key].second) h->hd[key].first = h->hd[key].second = h->none;
    else if (idx == h->hd[key].first) h->hd[key].first = next;
    else if (idx == h->hd[key].second) h->hd[key].second = prev;
}

static void uh_delete(unitheap *h, uint64_t idx) { /* unitheap.cu:161-178 */
    h->update[idx] = h->infty;
    uint64_t prev = h->ll[idx].prev, next = h->ll[idx].next;
    if (prev != h->none) h->ll[prev].next = next;
    if (next != h->none) h->ll[next].prev = prev;
    uh_erase_key(h, idx, next, prev);
    if (h->top == idx) h->top = next;
    h->ll[idx].prev = h->ll[idx].next = h->none;
    h->heapsize--;
}

static void uh_decrease_top(unitheap *h) { /* unitheap.cu:107-158 */
    const uint64_t top = h->top, next = h->ll[top].next;
    if (next == h->none) return;
    const int key = h->ll[top].key;
    const int leftover = h->update[top] / 2;
    const int new_key = key + h->update[top] - leftover;
    if (-h->update[top] > key) { h->failed = 1; return; }
    if (new_key >= h->ll[next].key) return;
    h->update[top] = leftover;
    uint64_t level_tail = h->hd[key].second;
    uint64_t next_level = h->ll[level_tail].next;
    while (next_level != h->none && h->ll[next_level].key >= new_key) {
        level_tail = h->hd[h->ll[next_level].key].second;
        next_level = h->ll[level_tail].next;
    }
    h->ll[next].prev = h->none;
    h->ll[top].prev = level_tail;
    h->ll[top].next = next_level;
    h->ll[level_tail].next = top;
    if (next_level != h->none) h->ll[next_level].prev = top;
    uh_erase_key(h, top, next, h->none);
    if (new_key < 0) { h->failed = 1; return; }
    h->ll[top].key = new_key;
    h->hd[new_key].second = top;
    if (h->hd[new_key].first == h->none) h->hd[new_key].first = top;
    h->top = next;
}

static uint64_t uh_extract_max(unitheap *h) { /* unitheap.cu:90-104 */
    uint64_t tmptop;
    do {
        tmptop = h->top;
        if (h->update[h->top] < 0) uh_decrease_top(h);
        if (h->failed) return h->none;
    } while (h->top != tmptop);
    uh_delete(h, h->top == tmptop ? tmptop : tmptop); /* DeleteElement(top): top == tmptop here */
    return tmptop;
}

static void uh_increment_key(unitheap *h, uint64_t idx) { /* unitheap.cu:195-224 */
    const uint64_t level_head = h->hd[h->ll[idx].key].first;
    const uint64_t prev = h->ll[idx].prev, next = h->ll[idx].next;
    if (level_head != idx) {
        h->ll[prev].next = next;
        if (next != h->none) h->ll[next].prev = prev;
        uint64_t prev_level = h->ll[level_head].prev;
        h->ll[idx].prev = prev_level;
        h->ll[idx].next = level_head;
        h->ll[level_head].prev = idx;
        if (prev_level != h->none) h->ll[prev_level].next = idx;
    }
    uh_erase_key(h, idx, next, prev);
    int key = ++h->ll[idx].key;
    h->hd[key].second = idx;
    if (h->hd[key].first == h->none) {
        h->hd[key].first = idx;
        if (key > h->ll[h->top].key) h->top = idx;
    }
    if ((size_t)key + 4 >= h->hd_size) uh_header_resize(h, (size_t)(h->hd_size * 1.5));
}

static void uh_lazy_increment(unitheap *h, uint64_t idx, int up) { /* unitheap.cu:185-193 */
    if (h->update[idx] == h->infty) return;
    if (h->update[idx] == 0 && up > 0) uh_increment_key(h, idx);
    else {
        h->update[idx] += up;
        if (-h->update[idx] > h->ll[idx].key) h->failed = 1;
    }
}

typedef struct { uint64_t n; uint64_t *cd; uint64_t *adj; } badj; /* out lists [0,n), in lists [n,2n) */

static void gorder_move_window(const badj *g, unitheap *h, uint64_t new_node, uint64_t old_node,
                               uint64_t *tmp_old, uint64_t *tmp_new) { /* order_gorder.cu:88-143 */
    const uint64_t n = g->n;
#define OUT_DEG(u) (g->cd[(u) + 1] - g->cd[u])
    const uint64_t *old_p = g->adj + g->cd[old_node + n], *old_e = g->adj + g->cd[old_node + 1 + n];
    const uint64_t *new_p = g->adj + g->cd[new_node + n], *new_e = g->adj + g->cd[new_node + 1 + n];
    if (old_node == new_node) old_p = old_e;
    else if (OUT_DEG(old_node) <= h->huge)
        for (uint64_t a = g->cd[old_node]; a < g->cd[old_node + 1]; ++a) uh_lazy_increment(h, g->adj[a], -1);
    size_t n_old = 0, n_new = 0;
    for (;;) {
        int factor = -1;
        if (old_p >= old_e) {
            if (new_p >= new_e) break;
            factor = 1;
        } else if (new_p < new_e) {
            if (*new_p == *old_p) { old_p++; new_p++; continue; }
            if (*new_p < *old_p) factor = 1;
        }
        if (factor == -1) {
            if (OUT_DEG(*old_p) <= h->huge) tmp_old[n_old++] = *old_p;
            old_p++;
        } else {
            if (OUT_DEG(*new_p) <= h->huge) tmp_new[n_new++] = *new_p;
            new_p++;
        }
    }
    for (size_t i = 0; i < n_old; ++i) {
        const uint64_t parent = tmp_old[i];
        uh_lazy_increment(h, parent, -1);
        for (uint64_t a = g->cd[parent]; a < g->cd[parent + 1]; ++a)
            if (g->adj[a] != old_node) uh_lazy_increment(h, g->adj[a], -1);
    }
    if (OUT_DEG(new_node) <= h->huge)
        for (uint64_t a = g->cd[new_node]; a < g->cd[new_node + 1]; ++a) uh_lazy_increment(h, g->adj[a], +1);
    for (size_t i = 0; i < n_new; ++i) {
        const uint64_t parent = tmp_new[i];
        uh_lazy_increment(h, parent, +1);
        for (uint64_t a = g->cd[parent]; a < g->cd[parent + 1]; ++a)
            if (g->adj[a] != new_node) uh_lazy_increment(h, g->adj[a], +1);
    }
#undef OUT_DEG
}

int oracle_order_gorder(int64_t n, const uint32_t *rowPtr, const uint32_t *col, uint64_t window, uint64_t *rank) {
    if (n == 0) return 0;
    const uint64_t e = rowPtr[n];
    uint64_t *rank_rcm = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n);
    int rc = oracle_order_rcm(n, rowPtr, col, rank_rcm);
    if (rc) { free(rank_rcm); return rc; }
    /* Badjlist g(h, rank_rcm): adjlist.cu:156-185, neighbours sorted (adjlist.cu:62-73) */
    badj g;
    g.n = (uint64_t)n;
    g.cd = (uint64_t *)calloc((size_t)2 * n + 1, sizeof(uint64_t));
    g.adj = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(e ? 2 * e : 1));
    uint64_t *fill = (uint64_t *)calloc((size_t)2 * n + 1, sizeof(uint64_t));
    for (int64_t i = 0; i < n; ++i)
        for (uint32_t j = rowPtr[i]; j < rowPtr[i + 1]; ++j) {
            g.cd[rank_rcm[i] + 1]++;
            g.cd[rank_rcm[col[j]] + n + 1]++;
        }
    for (int64_t u = 0; u < 2 * n; ++u) g.cd[u + 1] += g.cd[u];
    for (int64_t i = 0; i < n; ++i)
        for (uint32_t j = rowPtr[i]; j < rowPtr[i + 1]; ++j) {
            const uint64_t u = rank_rcm[i], v = rank_rcm[col[j]];
            g.adj[g.cd[u] + fill[u]++] = v;
            g.adj[g.cd[v + n] + fill[v + n]++] = u;
        }
    for (int64_t u = 0; u < 2 * n; ++u) qsort(g.adj + g.cd[u], (size_t)(g.cd[u + 1] - g.cd[u]), sizeof(uint64_t), cmp_u64);
    /* order_gorder: order_gorder.cu:35-84 */
    unitheap h;
    uh_init(&h, (uint64_t)n);
    uint64_t *order = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n);
    uint64_t *isolates = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n);
    uint64_t *tmp_old = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n + 1));
    uint64_t *tmp_new = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n + 1));
    size_t n_order = 0, n_iso = 0;
    for (int64_t u = 0; u < n; ++u) {
        const uint64_t dout = g.cd[u + 1] - g.cd[u], din = g.cd[u + 1 + n] - g.cd[u + n];
        if (dout + din == 0) isolates[n_iso++] = (uint64_t)u;
        else { /* InsertElement, unitheap.cu:24-29 */
            h.ll[u].key = (int)din;
            h.update[u] = -(int)din;
            h.heapsize++;
        }
    }
    rc = 0;
    /* With an isolated vertex the reference's ReConstruct links indices 0..heapsize-1 instead of the
     * inserted ids (unitheap.cu:35-38), picks up a key of INT_MAX/2 and sizes Header to 10x that:
     * it cannot complete.  Reported as a domain error instead of reproducing the crash. */
    if (n_iso > 0) rc = -EDOM;
    if (h.heapsize > 0 && !rc) {
        uh_reconstruct(&h);
        const uint64_t hub = h.top;
        order[n_order++] = hub;
        uh_delete(&h, hub);
        gorder_move_window(&g, &h, hub, hub, tmp_old, tmp_new);
        while (h.heapsize > 0 && !h.failed) {
            const uint64_t nn = uh_extract_max(&h);
            if (h.failed || nn >= (uint64_t)n) { rc = -EDOM; break; }
            order[n_order++] = nn;
            uint64_t old_node = nn;
            if (n_order > window) old_node = order[n_order - window - 1];
            gorder_move_window(&g, &h, nn, old_node, tmp_old, tmp_new);
        }
        if (h.failed) rc = -EDOM;
    }
    for (size_t i = 0; i < n_iso && n_order < (size_t)n; ++i) order[n_order++] = isolates[i];
    if (!rc && n_order != (size_t)n) rc = -EDOM;
    if (!rc) {
        uint64_t *rank_g = fill; /* reuse: rank_from_order */
        for (int64_t i = 0; i < n; ++i) rank_g[order[i]] = (uint64_t)i;
        for (int64_t u = 0; u < n; ++u) rank[u] = rank_g[rank_rcm[u]]; /* order_gorder.cu:26-29 */
    }
    uh_free(&h);
    free(order); free(isolates); free(tmp_old); free(tmp_new); free(fill); free(g.cd); free(g.adj); free(rank_rcm);
    return rc;
}

/* --------------------------------------------------------------------- DFS */
/* DataLoaderDFS, DataLoader.cu:324-395: iterative DFS from vertex 0 over the CSR's out-edges in
 * stored order, new ids in discovery order (rowPtr grows by one entry per discovery, :358-372),
 * next root = next vertex without an id (:377-379).  rank[old] = new. */
int oracle_order_dfs(int64_t n, const uint32_t *rowPtr, const uint32_t *col, uint64_t *rank) {
    uint8_t *seen = (uint8_t *)calloc((size_t)n + 1, 1);
    uint32_t *stack_v = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n + 1));
    uint32_t *stack_e = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n + 1));
    if (!seen || !stack_v || !stack_e) return -ENOMEM;
    uint64_t next_id = 0;
    for (int64_t root = 0; root < n; ++root) {
        if (seen[root]) continue;
        int64_t sp = 0;
        seen[root] = 1;
        rank[root] = next_id++;
        stack_v[sp] = (uint32_t)root;
        stack_e[sp++] = rowPtr[root];
        while (sp > 0) {
            const uint32_t u = stack_v[sp - 1];
            while (stack_e[sp - 1] < rowPtr[u + 1] && seen[col[stack_e[sp - 1]]]) stack_e[sp - 1]++;
            if (stack_e[sp - 1] == rowPtr[u + 1]) { sp--; continue; }
            const uint32_t v = col[stack_e[sp - 1]++];
            seen[v] = 1;
            rank[v] = next_id++;
            stack_v[sp] = v;
            stack_e[sp++] = rowPtr[v];
        }
    }
    free(seen); free(stack_v); free(stack_e);
    return 0;
}

/* ------------------------------------------------------------------ Rabbit */
/* DataLoaderRabbit, DataLoader.cu:455-655 with its compile-time options as the reference sets them (opt_iterative = true,
 * opt_hub_group = opt_hub_sort = false, cluster shyness 1): modularity-driven agglomeration in rounds.
 *   - every vertex keeps a map neighbour -> weight (std::map<int,int> dst_wht: iteration in ascending key); weight 1 per
 *     distinct neighbour, self loops left out, and for a directed graph (dl.is_directed) the reverse edge is added too
 *     (force_undirected, :516-531); deg = size of the map when the vertex's turn in the construction loop ends (:529), n_edges =
 *     sum of deg, two_m_inv = 1/(2 n_edges);
 *   - round: vertices of the round sorted by current deg (:545-546); u is skipped if it absorbed something this round
 *     (:554); its target is the neighbour with the largest  w - deg[d] * deg[u] * two_m_inv,  first maximum in ascending
 *     d (set_max is a strict >, common.h:101-107; start value -1), and only if that is > 0 (:556-560);
 *   - merge u into v (:563-583): deg[v] += deg[u]; every (d,w) of u except d == v is added to v's map and, where d's map
 *     holds u, that entry moves to v; u leaves v's map; dendrogram node (v's tree, u's tree) becomes v's tree (:586-588);
 *     v enters the next round once (:590-592);
 *   - order (:614-628, 633-648): vertices in index order that still own a tree, leaves left to right.
 * rank[old] = new.  One liberty: the reference sorts a round with ranges::sort, which leaves the order of equal degrees to
 * the library; here equal degrees keep the order they had in the list (a stable sort) -- the same rule in flex_order_rabbit. */
typedef struct { uint32_t key; int32_t w; } rb_ent;
typedef struct { rb_ent *e; uint32_t n, cap; } rb_map;

static int64_t rb_find(const rb_map *m, uint32_t key) { /* index of key, or -(insertion point)-1 */
    int64_t lo = 0, hi = (int64_t)m->n - 1;
    while (lo <= hi) {
        const int64_t mid = (lo + hi) / 2;
        if (m->e[mid].key == key) return mid;
        if (m->e[mid].key < key) lo = mid + 1; else hi = mid - 1;
    }
    return -lo - 1;
}
static int rb_add(rb_map *m, uint32_t key, int32_t w) { /* m[key] += w (creating it at 0) */
    int64_t i = rb_find(m, key);
    if (i >= 0) { m->e[i].w += w; return 0; }
    i = -i - 1;
    if (m->n == m->cap) {
        const uint32_t cap = m->cap ? m->cap * 2 : 4;
        rb_ent *e = (rb_ent *)realloc(m->e, sizeof(rb_ent) * cap);
        if (!e) return -ENOMEM;
        m->e = e; m->cap = cap;
    }
    memmove(m->e + i + 1, m->e + i, sizeof(rb_ent) * (m->n - (size_t)i));
    m->e[i].key = key; m->e[i].w = w; m->n++;
    return 0;
}
static void rb_erase(rb_map *m, uint32_t key) {
    const int64_t i = rb_find(m, key);
    if (i < 0) return;
    memmove(m->e + i, m->e + i + 1, sizeof(rb_ent) * (m->n - (size_t)i - 1));
    m->n--;
}

int oracle_order_rabbit(int64_t n, const uint32_t *rowPtr, const uint32_t *col, int is_directed, uint64_t *rank) {
    if (n == 0) return 0;
    rb_map *g = (rb_map *)calloc((size_t)n, sizeof(rb_map));
    int64_t *deg = (int64_t *)calloc((size_t)n, sizeof(int64_t));
    int32_t *round_of = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    /* dendrogram: node ids 0..n-1 are leaves, n+u is the cluster node made when u was merged away */
    int64_t *lch = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)n), *rch = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)n);
    int64_t *tree = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    uint32_t *cur = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n), *nxt = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n);
    uint32_t *tmp = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n);
    int64_t *stack = (int64_t *)malloc(sizeof(int64_t) * (2 * (size_t)n + 2));
    if (!g || !deg || !round_of || !lch || !rch || !tree || !cur || !nxt || !tmp || !stack) return -ENOMEM;
    int rc = 0;
    int64_t n_edges = 0;
    for (int64_t v = 0; v < n && !rc; ++v) {
        for (uint32_t e = rowPtr[v]; e < rowPtr[v + 1] && !rc; ++e) {
            const uint32_t d = col[e];
            if (d == (uint32_t)v) continue;
            int64_t i = rb_find(&g[v], d);
            if (i < 0) rc = rb_add(&g[v], d, 1); /* dst_wht[d] = 1 (an assignment: duplicates do not add up) */
            if (is_directed && !rc && rb_find(&g[d], (uint32_t)v) < 0) rc = rb_add(&g[d], (uint32_t)v, 1);
        }
        /* vo.deg = vo.dst_wht.size() INSIDE the construction loop (:529-530): v's own edges plus the reverse edges that the
         * vertices before it inserted -- for a directed graph the reverse edges of later vertices are in the map by the
         * time the rounds start, but never in deg / n_edges.  Symmetric inputs are unaffected. */
        deg[v] = g[v].n;
        n_edges += deg[v];
    }
    for (int64_t v = 0; v < n; ++v) {
        tree[v] = v;
        lch[v] = rch[v] = -1;
        cur[v] = (uint32_t)v;
    }
    const double two_m_inv = 1.0 / (double)(2 * n_edges);
    size_t n_cur = (size_t)n;
    for (int32_t round = 1; n_cur > 0 && !rc; ++round) {
        /* stable sort of the round's vertices by current degree (bottom-up merge sort) */
        for (size_t w = 1; w < n_cur; w *= 2) {
            for (size_t lo = 0; lo < n_cur; lo += 2 * w) {
                const size_t mid = lo + w < n_cur ? lo + w : n_cur, hi = lo + 2 * w < n_cur ? lo + 2 * w : n_cur;
                size_t a = lo, b = mid, o = lo;
                while (a < mid && b < hi) tmp[o++] = deg[cur[b]] < deg[cur[a]] ? cur[b++] : cur[a++];
                while (a < mid) tmp[o++] = cur[a++];
                while (b < hi) tmp[o++] = cur[b++];
            }
            uint32_t *sw = cur; cur = tmp; tmp = sw;
        }
        size_t n_nxt = 0;
        for (size_t idx = 0; idx < n_cur && !rc; ++idx) {
            const uint32_t u = cur[idx];
            if (round_of[u] == round) continue;
            double dq_max = -1.0;
            int64_t v = -1;
            const double dv_2m = (double)deg[u] * two_m_inv;
            for (uint32_t i = 0; i < g[u].n; ++i) {
                const double q = (double)g[u].e[i].w - (double)deg[g[u].e[i].key] * dv_2m;
                if (q > dq_max) { dq_max = q; v = g[u].e[i].key; }
            }
            if (dq_max <= 0.0) continue;
            deg[v] += deg[u];
            for (uint32_t i = 0; i < g[u].n && !rc; ++i) {
                const uint32_t d = g[u].e[i].key;
                if (d == (uint32_t)v) continue;
                rc = rb_add(&g[v], d, g[u].e[i].w);
                const int64_t j = rb_find(&g[d], u);
                if (j < 0 || rc) continue;
                const int32_t wu = g[d].e[j].w;
                rc = rb_add(&g[d], (uint32_t)v, wu);
                rb_erase(&g[d], u);
            }
            rb_erase(&g[v], u);
            lch[n + u] = tree[v];
            rch[n + u] = tree[u];
            tree[u] = -1;
            tree[v] = n + u;
            if (round_of[v] == round) continue;
            round_of[v] = round;
            nxt[n_nxt++] = (uint32_t)v;
        }
        uint32_t *sw = cur; cur = nxt; nxt = sw;
        n_cur = n_nxt;
    }
    uint64_t next_id = 0;
    for (int64_t v = 0; v < n && !rc; ++v) {
        if (tree[v] < 0) continue;
        int64_t sp = 0;
        stack[sp++] = tree[v];
        while (sp > 0) { /* leaves left to right */
            const int64_t t = stack[--sp];
            if (t < n) { rank[t] = next_id++; continue; }
            stack[sp++] = rch[t];
            stack[sp++] = lch[t];
        }
    }
    for (int64_t v = 0; v < n; ++v) free(g[v].e);
    free(g); free(deg); free(round_of); free(lch); free(rch); free(tree); free(cur); free(nxt); free(tmp); free(stack);
    return rc;
}

/* --------------------------------------------------------- MatrixMarket -> CSR */
/* mmio_allinone, data/SuiteSparse/mtx2csr.cc:57-247: coordinate entries read with fscanf in file
 * order ("%d %d %lg" real, "%d %d %lg %lg" complex keeping the real part, "%d %d %d" integer,
 * "%d %d" pattern -> 1.0), 1-based -> 0-based, symmetric/hermitian files mirrored; rows filled
 * in file order (columns NOT sorted).  Arrays are malloc'ed; free with free(). */
int oracle_mtx_load(const char *path, int64_t *m_out, int64_t *n_out, int64_t *nnz_out,
                    uint32_t **rowPtr_out, uint32_t **col_out, float **vals_out) {
    FILE *f = fopen(path, "r");
    if (!f) return -ENOENT;
    char line[1100], banner[64], mtx[64], crd[64], dtype[64], storage[64];
    if (!fgets(line, sizeof line, f) ||
        sscanf(line, "%63s %63s %63s %63s %63s", banner, mtx, crd, dtype, storage) != 5) { fclose(f); return -EINVAL; }
    for (char *p = mtx; *p; ++p) *p = (char)tolower(*p);
    for (char *p = crd; *p; ++p) *p = (char)tolower(*p);
    for (char *p = dtype; *p; ++p) *p = (char)tolower(*p);
    for (char *p = storage; *p; ++p) *p = (char)tolower(*p);
    if (strcmp(banner, "%%MatrixMarket") || strcmp(mtx, "matrix") || strcmp(crd, "coordinate")) { fclose(f); return -EINVAL; }
    const int isPattern = !strcmp(dtype, "pattern"), isReal = !strcmp(dtype, "real"),
              isComplex = !strcmp(dtype, "complex"), isInteger = !strcmp(dtype, "integer");
    const int isSym = !strcmp(storage, "symmetric") || !strcmp(storage, "hermitian");
    if (!(isPattern || isReal || isComplex || isInteger)) { fclose(f); return -EINVAL; }
    do {
        if (!fgets(line, sizeof line, f)) { fclose(f); return -EINVAL; }
    } while (line[0] == '%');
    int m, n, nz;
    if (sscanf(line, "%d %d %d", &m, &n, &nz) != 3) { fclose(f); return -EINVAL; }
    int *cnt = (int *)calloc((size_t)m + 1, sizeof(int));
    int *ri = (int *)malloc(sizeof(int) * (size_t)(nz ? nz : 1)), *ci = (int *)malloc(sizeof(int) * (size_t)(nz ? nz : 1));
    float *vv = (float *)malloc(sizeof(float) * (size_t)(nz ? nz : 1));
    for (int i = 0; i < nz; ++i) {
        int a, b, iv, got;
        double fv = 0, fim;
        if (isReal) got = fscanf(f, "%d %d %lg\n", &a, &b, &fv) == 3;
        else if (isComplex) got = fscanf(f, "%d %d %lg %lg\n", &a, &b, &fv, &fim) == 4;
        else if (isInteger) { got = fscanf(f, "%d %d %d\n", &a, &b, &iv) == 3; fv = iv; }
        else { got = fscanf(f, "%d %d\n", &a, &b) == 2; fv = 1.0; }
        if (!got) { fclose(f); free(cnt); free(ri); free(ci); free(vv); return -EINVAL; }
        a--; b--;
        cnt[a]++;
        ri[i] = a; ci[i] = b; vv[i] = (float)fv;
    }
    fclose(f);
    if (isSym)
        for (int i = 0; i < nz; ++i)
            if (ri[i] != ci[i]) cnt[ci[i]]++;
    uint32_t *rp = (uint32_t *)calloc((size_t)m + 1, sizeof(uint32_t));
    for (int r = 0; r < m; ++r) rp[r + 1] = rp[r] + (uint32_t)cnt[r];
    const uint32_t total = rp[m];
    uint32_t *col = (uint32_t *)malloc(sizeof(uint32_t) * (total ? total : 1));
    float *vals = (float *)malloc(sizeof(float) * (total ? total : 1));
    memset(cnt, 0, sizeof(int) * ((size_t)m + 1));
    for (int i = 0; i < nz; ++i) {
        uint32_t o = rp[ri[i]] + (uint32_t)cnt[ri[i]]++;
        col[o] = (uint32_t)ci[i]; vals[o] = vv[i];
        if (isSym && ri[i] != ci[i]) {
            o = rp[ci[i]] + (uint32_t)cnt[ci[i]]++;
            col[o] = (uint32_t)ri[i]; vals[o] = vv[i];
        }
    }
    free(cnt); free(ri); free(ci); free(vv);
    *m_out = m; *n_out = n; *nnz_out = total;
    *rowPtr_out = rp; *col_out = col; *vals_out = vals;
    return 0;
}
void oracle_free(void *p) { free(p); }
