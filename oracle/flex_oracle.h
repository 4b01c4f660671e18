/*
 * flex_oracle.h -- CPU restatement of the reference's SpMM hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under flex_amd/ may include, link or call
 * this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do,
 * and there only as the checker / reported baseline.
 *
 * Every function cites the reference lines (paths relative to the reference
 * tree) whose arithmetic it restates.
 *
 * PARITY UNPINNED.  The reference cannot be built in this image (it needs nvcc,
 * cuSPARSE, cuBLAS and the external "gp"/NPerf library; stand-in headers are not
 * written), and its tree holds no golden outputs, known-answer tests or fixtures
 * for this path: its checks are run-time asserts against cuSPARSE, whose output
 * nothing records.  What this restatement IS checked against:
 *   - the reference's own DATA files (data/pubmed.csv, data/a_mat.csv);
 *   - three numbers SURVEY.md 8(c)/3.3 recorded from a stub-header probe of the
 *     reference's host half (cpuX prefix, sum(C) on pubmed k=32, RCM bandwidth
 *     of pubmed): reproduced, but survey-recorded numbers are not reference-held
 *     fixtures and pin nothing;
 *   - scipy.sparse / numpy in float64 and a literal Python transcription of the
 *     8-line loop, which share no code with this file (tests/).
 * Rabbit, Gorder and DFS have no reference figure at all.
 */
#ifndef FLEX_ORACLE_H
#define FLEX_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_csr {
    int64_t m, n, nnz;
    uint32_t *rowPtr; /* m+1 */
    uint32_t *col;    /* nnz */
    float *vals;      /* nnz */
    /* graph statistics, DataLoader.cu:26-29,86-115 */
    int64_t uni_nb;
    int64_t n_edges_one_way, n_edges_asymmetric;
    int32_t n_nodes_z_out, n_nodes_z_in, n_nodes_z_deg;
    int32_t is_directed;
    int32_t c; /* classes by file name, DataLoader.cu:62-84 */
} oracle_csr;

/* DataLoader::DataLoader, DataLoader.cu:9-124 (CSV -> CSR + stats).
 * `rand_state_reset` != 0 calls srand(1) first (the reference never calls srand,
 * so a fresh process sees seed 1; tests reset explicitly to be order-independent).
 * Returns 0 or a negative errno-style code. */
int oracle_csv_load(const char *path, oracle_csr *out, int rand_state_reset);
void oracle_csr_free(oracle_csr *a);

/* DataLoader::cuda_alloc_cpy, DataLoader.cu:198-209: row-major B[i*k+j] =
 * 2*(float)rand()/(float)RAND_MAX - 1.0f in i-then-j order. */
void oracle_gen_B(int64_t n, int k, float *B, int rand_state_reset);

/* aspt/sspmm_128.cu:1412-1422: C=0; for nz in CSR order, for j: C[row*k+j] +=
 * B[k*col+j]*val.  fp32, product rounded then added (built -ffp-contract=off). */
void oracle_spmm(int64_t m, const uint32_t *rowPtr, const uint32_t *col,
                 const float *vals, const float *B, float *C, int k);

/* Row-parallel variant of the same arithmetic (identical per-element order, so
 * bit-identical results); used only as the multi-core CPU baseline in bench.py. */
void oracle_spmm_mt(int64_t m, const uint32_t *rowPtr, const uint32_t *col,
                    const float *vals, const float *B, float *C, int k, int nthreads);

/* resCheck, flex.cu:4154-4213.  row_nnz comes from the ORIGINAL ordering's rowPtr.
 * Returns the mismatch count; writes max error, nnz of the max-error row, and the
 * number of exactly-zero gold elements. */
int64_t oracle_rescheck(const float *gold, const float *res, const uint32_t *orig_rowPtr,
                        int64_t m, int k, double *max_err, int32_t *max_err_row_nnz,
                        int64_t *gold_zeros);

/* order_rcm(h, directed=true), order_rcm.cu:15-33 with order_deg.cu:19-45,
 * adjlist.cu:62-73,127-150, algo_bfs.cu:11-39, tools.cu:31-43, edgelist.cu:23-32.
 * rank[old] = new. */
int oracle_order_rcm(int64_t n, const uint32_t *rowPtr, const uint32_t *col, uint64_t *rank);

/* complete_gorder(h, window), order_gorder.cu:13-31: RCM first, then Gorder over the RCM-relabelled
 * graph with the lazy unit heap of unitheap.cu; rank[old] = new.  DataLoaderGorder uses window 3
 * (DataLoader.cu:808). */
int oracle_order_gorder(int64_t n, const uint32_t *rowPtr, const uint32_t *col, uint64_t window, uint64_t *rank);

/* mmio_allinone, data/SuiteSparse/mtx2csr.cc:57-247: MatrixMarket coordinate file -> CSR in the
 * reference's (file-order, unsorted) layout.  Arrays are malloc'ed: release with oracle_free. */
int oracle_mtx_load(const char *path, int64_t *m, int64_t *n, int64_t *nnz, uint32_t **rowPtr, uint32_t **col,
                    float **vals);
void oracle_free(void *p);

/* DataLoaderDFS ordering, DataLoader.cu:324-395: depth-first discovery order from vertex 0. */
int oracle_order_dfs(int64_t n, const uint32_t *rowPtr, const uint32_t *col, uint64_t *rank);
/* DataLoaderRabbit (DataLoader.cu:455-655), options as compiled in the reference; rank[old] = new */
int oracle_order_rabbit(int64_t n, const uint32_t *rowPtr, const uint32_t *col, int is_directed, uint64_t *rank);

/* DataLoaderRcm / DataLoaderGorder body, DataLoader.cu:741-779 / 815-850: given
 * rank[old]=new, build vo_mp[new]=old and the permuted CSR with columns mapped and
 * sorted ascending per row. Output arrays are caller-allocated (same sizes). */
void oracle_perm_csr(int64_t n, const uint32_t *rowPtr, const uint32_t *col, const float *vals,
                     const uint64_t *rank, int32_t *vo_mp, uint32_t *rowPtr2, uint32_t *col2,
                     float *vals2);

#ifdef __cplusplus
}
#endif
#endif
