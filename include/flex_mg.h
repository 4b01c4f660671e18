/*
 * flex_mg.h -- C ABI of libflex_mg.so: one SpMM row-sharded over the GPUs of a node, single process.
 *
 * New work: the reference is single-GPU (one cudaSetDevice, flex.cu:4137).  Rows of A (and of C) are
 * independent units and B is replicated, so: A is re-ordered on the host, cut into contiguous row
 * shards of equal cost (flex_shard_rows), every GPU gets a plan for its shard (flex_plan_create_rows
 * with the column map of the re-ordering, so the un-permuted B is used), B is broadcast ONCE over
 * RCCL/xGMI, and flex_mg_spmm launches the shards on their own streams with no collective and no
 * reduction on the data path.  (bench.py does the same with one process per GPU and torch.distributed.)
 */
#ifndef FLEX_MG_H
#define FLEX_MG_H
#include "flex_spmm.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct flex_mg flex_mg;

/* devices == NULL means 0..ngpus-1.  order = FLEX_ORDER_* applied to the whole matrix before sharding.
 * Returns FLEX_OK or a flex_status (FLEX_ERR_HIP also covers RCCL failures; see flex_mg_last_rccl). */
int flex_mg_create(flex_mg **out, const flex_csr *hostA, int k, int ngpus, const int *devices, unsigned order);
/* hostB: n x k row-major.  Copies it to the first GPU and broadcasts it to the others over RCCL;
 * *bcast_ms (may be NULL) receives the wall time of the broadcast alone. */
int flex_mg_set_B(flex_mg *h, const float *hostB, double *bcast_ms);
/* one SpMM on every GPU (asynchronous, each shard on its own stream) */
int flex_mg_spmm(flex_mg *h);
/* wait for all GPUs; *max_ms (may be NULL) = slowest shard's device time of the last flex_mg_spmm batch
 * (events around `reps` launches issued by flex_mg_time) */
int flex_mg_sync(flex_mg *h);
/* convenience: warmup + reps timed launches on all GPUs; per-step time of the slowest shard in microseconds */
int flex_mg_time(flex_mg *h, int warmup, int reps, double *us_per_step);
/* hostC: m x k row-major in ORIGINAL row order */
int flex_mg_get_C(flex_mg *h, float *hostC);
/* row boundaries (ngpus+1 entries, in the re-ordered numbering) and nnz per shard, for reports */
int flex_mg_shard_info(const flex_mg *h, int64_t *row_bounds, int64_t *shard_nnz);
int flex_mg_destroy(flex_mg *h);
int flex_mg_last_rccl(void);

#ifdef __cplusplus
}
#endif
#endif
