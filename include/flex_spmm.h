/*
 * flex_spmm.h -- C ABI of the MI355X-native SpMM engine (libflex_spmm.so).
 *
 * This is the drop-in boundary for the one hot path of guohaoqiang/Flex:
 *     C[m x k] = A[m x n, CSR, fp32] * B[n x k, row-major fp32]
 * The reference has no FFI layer; its seam is "host-side Mat/DataLoader objects
 * -> kernel launch" (flex.cu:4979-4988, 5057-5059, 5690-5693).  Each entry point
 * below names the reference code it replaces (paths relative to the reference
 * tree).  Plain pointers and sizes only; nothing here throws, every function
 * returns 0 (FLEX_OK) or a negative flex_status.  See INTEGRATION.md for the
 * reference-side binding.
 *
 * There is deliberately NO CPU SpMM in this library: a missing GPU or a missing
 * kernel is an error (FLEX_ERR_HIP / FLEX_ERR_UNSUPPORTED), never a fallback.
 * The reference's CPU loop (aspt/sspmm_128.cu:1415-1422) lives in oracle/ as
 * test infrastructure.
 */
#ifndef FLEX_SPMM_H
#define FLEX_SPMM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLEX_ABI_VERSION 3 /* 3: plan-time knobs leave the environment for the struct flex_plan_tuning, flex_plan_desc.tuning, flex_plan_get_tuning,
                              flex_order_cluster_ex, flex_set_host_threads; split rows are summed by a second launch by default.
                              2: flex_plan_info / flex_plan_stats grew; FLEX_PLAN_ROW_RANGE; flex_order_rabbit, flex_plan_measure_imbalance */

typedef enum flex_status {
    FLEX_OK = 0,
    FLEX_ERR_INVALID = -1,     /* bad argument (null, negative size, k<=0, unsorted rowPtr, col>=n) */
    FLEX_ERR_NOMEM = -2,       /* host allocation failed */
    FLEX_ERR_HIP = -3,         /* a HIP runtime call failed (≙ CUDA_CHECK, common.h:53-60); see flex_last_hip_error */
    FLEX_ERR_UNSUPPORTED = -4, /* shape outside what the kernels cover (nnz >= 2^32, m >= 2^31) */
    FLEX_ERR_IO = -5,          /* file could not be opened / read */
    FLEX_ERR_FORMAT = -6,      /* CSV does not parse (≙ stoi/stof throwing, assert(col.size()==vals.size()), DataLoader.cu:57) */
    FLEX_ERR_DUPLICATE = -7    /* duplicate (row,col) (≙ assert(e_inv[dst].count(r)==0), DataLoader.cu:97) */
} flex_status;

/* Host CSR view: the three vectors of DataLoader (DataLoader.cuh:32-34) + sizes (:70).
 * Indices are 32-bit unsigned exactly as in the reference. Not owned. */
typedef struct flex_csr {
    int32_t m, n;
    int64_t nnz;
    const uint32_t *rowPtr; /* m+1 */
    const uint32_t *col;    /* nnz, any order within a row, must be < n */
    const float *vals;      /* nnz */
} flex_csr;

/* HIP stream handle; identical to hipStream_t (pass torch's cuda_stream integer cast to a pointer). */
typedef struct ihipStream_t *flex_stream_t;

typedef struct flex_plan flex_plan;

/* flags for flex_plan_create: bits 0-3 = row schedule */
#define FLEX_ORDER_NATURAL 0u /* rows processed in the order given ("OVO") */
#define FLEX_ORDER_RCM 1u     /* rows scheduled in the reference's RCM order (order_rcm.cu:15-33);
                                 columns keep ORIGINAL ids, so B needs no permuteX pass and C
                                 comes out in original row order */
#define FLEX_ORDER_CLUSTER 2u /* rows scheduled community by community (agglomerative modularity
                                 clustering, ≙ DataLoaderRabbit, DataLoader.cu:453-655); same
                                 no-permutation contract as FLEX_ORDER_RCM */
#define FLEX_ORDER_GORDER 3u  /* rows scheduled in Gorder(window 3) order (≙ DataLoaderGorder, DataLoader.cu:789-857) */
#define FLEX_ORDER_MASK 0xFu
#define FLEX_PLAN_STATS 0x100u /* also collect flex_plan_stats while planning (one extra pass over the records) */
#define FLEX_PLAN_AUTOTUNE 0x200u /* measure instead of trusting the degree rule: plan the neighbouring column-tile
                                     widths too (same row schedule), time each on zero-filled operands of the real
                                     size, keep the fastest; likewise the other setting of tuning.bundle where the tile
                                     has row bundles and the caller left it to the rule.  Costs up to three extra plans
                                     and, for the duration of the call, device memory for one B and one C */
#define FLEX_PLAN_ROW_RANGE 0x1000u /* flex_plan_create_ex only: desc->row_begin/row_end name a row shard */
#define FLEX_PLAN_XCD_INTERLEAVE 0x2000u /* deal the chunks round-robin over the 8 XCDs instead of giving each XCD one
                                            contiguous eighth of the schedule.  For rows that arrive in a BFS-like order
                                            (a loader reordered by RCM / Gorder and planned as given): eighths of
                                            such an order run at different speeds.  FLEX_ORDER_RCM and FLEX_ORDER_GORDER
                                            imply it */

/* ≙ Mat::Mat + csr2_DiagTiling + alpha_transfer (mat.cu:7-31, 680-942, 268-293):
 * builds the row-panel plan for `hostA` and uploads it to `device`.  The reference's
 * pillar tiler is replaced by an nnz-balanced wave/row-panel planner. */
int flex_plan_create(flex_plan **out, const flex_csr *hostA, int k, int device, unsigned flags);

/* Same with strided dense operands: row r of B starts at dB + r*ldb, row r of C at dC + r*ldc (floats,
 * ldb >= k, ldc >= k); the ldc-k trailing floats of every C row are left untouched.  For callers whose
 * feature width is not a multiple of 32: a B row that is not a whole number of 128-byte cache lines costs
 * every gather an extra line (k=100 runs 50 % slower than k=128 on the reddit shape), so store k=100 with
 * ldb = ldc = 128.  No reference counterpart (the reference runs k = 32 and 128 only). */
int flex_plan_create_ld(flex_plan **out, const flex_csr *hostA, int k, int ldb, int ldc, int device, unsigned flags);

/* Same, for a CSR that a reordered loader already permuted (DataLoaderRcm &c.,
 * DataLoader.cu:723-857): row r' of hostA is original row vo_mp[r'] and column c' is
 * original column vo_mp[c'].  The plan folds both maps in at build time, so flex_spmm
 * still takes B and returns C in ORIGINAL order (the reference needs permuteX +
 * segVoMap for that: flex.cu:276-289, mat.cu:816-824). vo_mp==NULL means identity. */
int flex_plan_create_mapped(flex_plan **out, const flex_csr *hostA, const int32_t *vo_mp, int k,
                            int device, unsigned flags);

/* Plan for the row slice [row_begin,row_end) of hostA only (a shard of a row-sharded
 * multi-GPU run; the reference is single-GPU, flex.cu:4137).  flex_spmm then writes
 * (row_end-row_begin) x k into dC, slice row 0 first; dB is still the full B.
 * col_map (or NULL): column c of hostA reads B row col_map[c] -- pass the vo_mp of a
 * reordered CSR to keep using the un-permuted B.  Rows are scheduled in the order given
 * (flags must be FLEX_ORDER_NATURAL: reorder first, then shard). */
int flex_plan_create_rows(flex_plan **out, const flex_csr *hostA, int64_t row_begin, int64_t row_end,
                          const int32_t *col_map, int k, int device, unsigned flags);

/* Every option of the four entry points above in one call (they are thin wrappers over it), for the combinations
 * they do not name -- e.g. a row shard of a reordered matrix over padded storage.  Zero-initialise, set
 * struct_size = sizeof(flex_plan_desc), fill what is needed:
 *   row_begin/row_end  read only when flags has FLEX_PLAN_ROW_RANGE: the shard [row_begin,row_end), which may be
 *                      empty (the order bits must then be FLEX_ORDER_NATURAL); without the flag: all rows
 *   col_map            column c of A reads B row col_map[c] (NULL = c)
 *   row_map            row r of A writes C row row_map[r] (NULL = r - row_begin); all rows only, A square
 *   ldb/ldc            row strides of B and C in floats (0 = k) */
/* Plan-time tuning knobs (ABI 3; rounds 1-2 read them from the environment, which is process-global and racy).  Every
 * field: 0 = the planner's measured rule.  Nothing in the product sets them; they exist for the tests, the
 * soak and tools/.  flex_plan_get_tuning returns the values a plan was actually built with. */
typedef struct flex_cluster_tuning { /* flex_order_cluster_ex / FLEX_ORDER_CLUSTER plans */
    int32_t batch;     /* proposals per batch of the agglomeration (4096) */
    int32_t no_refine; /* 1: skip the second stage (vertex moves between stretches of the order) */
    int32_t stretch;   /* positions per stretch (1024), >= 16 */
    int32_t sweeps;    /* sweep limit (8) */
    int32_t stride;    /* largest sampling stride of a long row (4) */
} flex_cluster_tuning;
typedef struct flex_plan_tuning {
    int32_t lanes_per_nz;    /* G = 4 (k <= 16 only) / 8 / 16 / 32 / 64 lanes per record (column tile of 4G columns), capped by k */
    int32_t chunk_records;   /* chunk budget: records per wave */
    int32_t long_row;        /* rows longer than this are cut into pieces (chunk_records) */
    int32_t piece_records;   /* ... of about this many records (chunk_records) */
    int32_t row_cost;        /* records a row boundary counts for when chunks are cut (16) */
    int32_t xcd_slices;      /* 1: every XCD walks one contiguous slice of the schedule; 2: round-robin (rule: 1 except RCM / Gorder);
                                3: stretches of xcd_stretch workgroups dealt to the XCDs in turn (the eight XCDs walk ADJACENT stretches of the
                                schedule at any time, each still on its own stretch) */
    int32_t xcd_balance;     /* 2: slices cut by chunk count instead of by cost */
    int32_t chunk_cost, task_cost; /* cost model of the slice balancing (16, 2) */
    int32_t split_rows;      /* how the pieces of a split row are summed: 2 = by spmm_fixup_kernel after the launch (the default: it
                                needs nothing beyond stream order); 1 = inside the launch by the piece that arrives last (relaxed
                                agent atomics + sc1 stores and loads: measured on gfx950, not an architectural guarantee) */
    int32_t rec_nt;          /* record stream read with non-temporal loads: 1 on, 2 off */
    int32_t unroll;          /* 8: eight gathers in flight per wave on the narrow tiles too */
    int32_t two_d;           /* 1: rows are also cut by column panel (the 2-D schedule) */
    int32_t panel_kb;        /* ... panel = this many KiB of one column tile of B (2048) */
    int32_t seg_min;         /* ... runs shorter than this stay in the row's last piece (4) */
    int32_t mfma;            /* dense 32x32 tiles to the MFMA kernel: 1 always route, 2 never (rule: when a sample finds >= 10 % of nnz) */
    int32_t mfma_fill_pct;   /* ... tiles of at least this fill (60) */
    int32_t lds_extra;       /* bytes of idle LDS per workgroup (occupancy throttle; multiple of 16) */
    int32_t host_threads;    /* host threads of this call (rule: flex_set_host_threads, else the core count, at most 32) */
    flex_cluster_tuning cluster;
    /* the hot-block path (the matrix is SPLIT: nonzeros with reuse inside a block of rows are multiplied out of LDS-staged B panels by
       their own kernel after the flat kernel has done the rest) */
    int32_t blocks;           /* 1: split, 2: never (rule: k >= 64 and a very large input -- >= 983 040 rows of average degree >= 48 -- on which
                                 a sampled look finds >= 72 % of the nonzeros in columns that a block of 480 rows uses three times or more;
                                 operands that are not 16-byte aligned run through generic kernels, correct and slow, as for flat plans) */
    int32_t block_rounds;     /* rows per slot: 2, 4 or 8; a block is rounds x 60 rows (8; 4 / 2 while there are few blocks per CU) */
    int32_t block_panel_rows; /* B rows per LDS panel: a multiple of 4, at most 304 (304) */
    int32_t block_thr;        /* a column is hot (staged) when at least this many nonzeros of the block use it (3) */
    int32_t block_cap;        /* nonzeros per slot: a longer row is spread over ceil(len / cap) slots, summed through LDS (1.5 x the average degree) */
    int32_t block_ablate;     /* timing-only experiments, the RESULT IS WRONG: 1 no panel staging, 2 no panel work, 16 no read-modify-write of C, 32 record loads folded onto one line */
    int32_t tile_group;       /* multi-tile launches: workgroups per group -- every XCD's slice of the schedule is walked group by group, all
                                 column tiles of a group back to back, so that a group's records are re-read from the Infinity Cache rather
                                 than from HBM (0 = rule; 1 = off: one pass over the whole schedule per tile) */
    int32_t xcd_stretch;     /* xcd_slices = 3: workgroups (4 chunks each) per stretch (256) */
    int32_t far_first;       /* > 0: inside every task the records whose column lies more than this many schedule positions from the row come
                                FIRST (a wave's gathers return in order: with the likely L2 misses issued together, only those groups of
                                gathers wait for the fabric) (rule: see plan_build.cpp, fill_records) */
    int32_t bundle;          /* row bundles: tasks that hold up to 64 / lanes_per_nz SHORT rows side by side, one per record slot -- no
                                cross-slot reduction and one store per lane at the end instead of a reduction and a store per row
                                (≙ the reference's narrow kernel giving every thread its own row, flex.cu:81-118): 1 on, 2 off
                                (rule: on when the plan holds a chunk for every wave slot of the card, or its average degree is below 16
                                -- and then on the tiles of 4 or more slots per step: wider k runs the 16-lane tile instead of the
                                32-lane one; plan_build.cpp, bundle_rule) */
    int32_t bundle_len;      /* ... rows of at most this many nonzeros are candidates (12 on the tiles of 8 or 16 slots per step, 16 on the 4-slot tile) */
    int32_t reserved[5];     /* zero */
} flex_plan_tuning;

typedef struct flex_plan_desc {
    size_t struct_size;
    const flex_csr *A;
    int k, ldb, ldc, device;
    unsigned flags;
    int64_t row_begin, row_end;
    const int32_t *col_map, *row_map;
    const flex_plan_tuning *tuning; /* ABI 3; NULL = rules.  A caller built against ABI 2 passes the shorter struct_size */
} flex_plan_desc;
int flex_plan_create_ex(flex_plan **out, const flex_plan_desc *desc);
int flex_plan_get_tuning(const flex_plan *plan, flex_plan_tuning *out);

/* Process-wide cap on the worker threads of the planner, the orderings and the generator (0 = the core count, at most
 * 32 either way); returns the previous value.  For N ranks on one host: host cores / N.  Thread-safe; results never
 * depend on the thread count (tests/test_planner_host.py). */
int flex_set_host_threads(int n);

/* ≙ launch_prep + cudaMemset(C) + kernel<<<>>> (mat.cu:32-41, flex.cu:5057-5059).
 * dB: n x k row-major device fp32; dC: m x k row-major device fp32, fully overwritten
 * (alpha=1, beta=0 as in cuSpmm, flex.cu:5728-5729).  Asynchronous on `stream`;
 * no allocation, no host sync (safe to capture in a hipGraph).  dB/dC must be
 * 16-byte aligned when k % 4 == 0.
 * NOT re-entrant per plan: a plan owns the partial-sum workspace and arrival counters of its split
 * rows, so at most ONE launch of a given plan may be in flight -- do not enqueue the same plan on two
 * streams, or replay two graphs holding it, concurrently (successive launches on one stream are fine;
 * different plans are independent).  Checked for eager launches: a launch on a DIFFERENT stream while the stream of the
 * plan's latest launch still has work pending returns FLEX_ERR_INVALID and enqueues nothing (a stream query when the
 * stream changes: no cost for a plan that stays on its stream; conservative -- unrelated work queued behind the plan's
 * launch on the old stream counts as pending too).  Launches captured into a graph are not checked (nor are replays):
 * ordering those is the caller's.  Plans without split rows (flex_plan_info.n_partials == 0) hold no workspace and may
 * overlap freely. */
int flex_spmm(flex_plan *plan, const float *dB, float *dC, flex_stream_t stream);

/* ≙ alpha_freeMatGPU (mat.cuh:184-193). */
int flex_plan_destroy(flex_plan *plan);

typedef struct flex_plan_info {
    int32_t m, n, k, device;
    int64_t nnz;
    int64_t n_tasks;      /* tasks: rows + pieces of split rows + row bundles (the rows inside a bundle are not tasks of their own) */
    int64_t n_chunks;     /* schedule chunks (runs of tasks): one wave each, four to a workgroup */
    int64_t n_split_rows; /* rows long enough to be split over several waves */
    int64_t n_partials;   /* k-wide partial sums held in the workspace */
    int64_t device_bytes; /* HBM held by the plan */
    int32_t lanes_per_nz; /* G: lanes that cooperate on one nonzero (k/4 rounded up to a power of two) */
    int32_t order;        /* FLEX_ORDER_* actually applied */
    double plan_ms;       /* host time spent planning + uploading */
    int64_t n_slots;      /* chunk-table entries launched: n_chunks + the empty entries that pad the XCD slices */
    int32_t two_d;        /* 1: rows are cut by column panel as well (each XCD slice runs phase by phase), else 0 */
    int32_t panel_rows;   /* two_d: B rows per column panel (a power of two), else 0 */
    int64_t n_tiles;      /* dense 32x32 tiles of A routed to the MFMA kernel (0: the vector kernel does everything) */
    int64_t tile_nnz;     /* nonzeros held by those tiles */
    int64_t n_records;    /* (col,val) records the vector kernel streams per column tile: nnz - tile_nnz + padding */
    /* the hot-block path (0 everywhere when the plan has no blocks) */
    int64_t n_blocks;         /* blocks: one workgroup each (per 64-column tile) */
    int64_t block_rows;       /* rows that hold slots in a block (empty rows, and rows longer than a whole block of slots, hold none) */
    int64_t block_nnz;        /* nonzeros of those rows */
    int64_t block_hot_nnz;    /* nonzeros in the block image: they read their B row from an LDS panel and are NOT in the flat plan's records */
    int64_t block_hot_cols;   /* B rows staged, summed over blocks: block_hot_nnz / block_hot_cols = u, the reuse of a staged row */
    int64_t block_panels;     /* panels staged per column tile, summed over blocks */
    int64_t block_records;    /* records the hot kernel streams per 64-column tile: block_hot_nnz + padding */
    /* row bundles (0 when the plan has none) */
    int64_t n_bundles;        /* tasks that hold several short rows side by side */
    int64_t bundle_rows;      /* rows inside them */
} flex_plan_info;
int flex_plan_get_info(const flex_plan *plan, flex_plan_info *out);

/* ≙ Mat::alpha_stats_collect (mat.cu:944-1065) and the B-Re1 / B-Re2 columns of run()'s table
 * (flex.cu:5217-5223): how much B-row reuse the schedule exposes to each level of the machine,
 * and how evenly the work is cut.  "B rows" are distinct column ids; padding records excluded.
 * Needs FLEX_PLAN_STATS at plan creation, otherwise FLEX_ERR_UNSUPPORTED. */
typedef struct flex_plan_stats {
    int64_t records;       /* nnz + per-row padding to whole gather steps */
    int64_t cols_wave;     /* sum over chunks of distinct B rows  (≙ n_col_sum: reuse inside one unit of work) */
    int64_t cols_wg;       /* sum over workgroups (4 chunks, one CU's L1) */
    int64_t cols_xcd;      /* sum over the 8 XCD slices of the chunk table (≙ acc_col: reuse inside one L2) */
    double reuse_wave;     /* nnz / cols_wave  -- B-Re1: 1 = none, ideal = average degree */
    double reuse_wg;       /* nnz / cols_wg */
    double reuse_xcd;      /* nnz / cols_xcd   -- B-Re2 */
    double gather_bytes;   /* no-reuse gather model: 4(n+1) + 8 nnz + 4 nnz k + 4 n k   (u = 1) */
    double l2_bytes;       /* same with B rows fetched once per XCD: 4(n+1) + 8 records + 4 k cols_xcd + 4 m k */
    int64_t chunk_rec_max; /* largest chunk, records */
    double chunk_rec_mean;
    double chunk_imb_pct;  /* 100 max/mean - 100   (≙ "wp imb") */
    double xcd_imb_pct;    /* records per XCD slice, 100 max/mean - 100   (≙ "sm imb") */
    double split_nnz_pct;  /* share of nonzeros in rows cut into pieces (≙ share of work needing atomics) */
    double pad_pct;        /* 100 (records - nnz) / nnz */
    int64_t n_workgroups;
    /* block-density detector (32x32 tiles in schedule coordinates; ≙ tools/block_density.py of round 1, now in the planner) */
    double tile_nnz_pct_10; /* share of the nonzeros in tiles of fill >= 0.10 */
    double tile_nnz_pct_25; /* ... >= 0.25 */
    double tile_nnz_pct_50; /* ... >= 0.50 */
    double tile_mean_fill;  /* nnz / (1024 * non-empty tiles) */
    int64_t mfma_tiles;     /* tiles routed to the MFMA kernel */
    double mfma_nnz_pct;    /* share of the nonzeros they hold */
    /* reuse a workgroup could have ABOVE the L2 (ABI 3): a column is hot in a block of 480 schedule-consecutive rows when
       at least thr of the block's nonzeros use it -- those B rows could be staged in LDS once and used u times (≙ the `u` of the
       reference's cost model, flex.cu:5513-5528, at the scope of one CU).  Every 4th block is looked at. */
    double lds_hot_pct_2, lds_hot_pct_4; /* share of the nonzeros in hot columns, thr = 2 / 4 */
    double lds_u_2, lds_u_4;             /* hot nonzeros per staged B row */
} flex_plan_stats;
int flex_plan_get_stats(const flex_plan *plan, flex_plan_stats *out);

/* ≙ the per-SM imbalance column of run()'s table (flex.cu:27-79 stamps %smid + clock() per warp, flex.cu:5087-5126
 * turns them into "Imb"): ONE extra launch of the plan with the stamped twin of the SpMM kernel (every wave records the
 * 100 MHz constant clock at start and end and the CU it ran on: XCC id + HW_ID), reduced on the host.  dB / dC as for
 * flex_spmm (16-byte aligned, k % 4 == 0; dC receives the ordinary result); synchronises `stream`.  Not part of
 * flex_spmm: the product launch carries no stamps. */
typedef struct flex_imbalance {
    int64_t waves;            /* waves that ran (chunk-table entries x column tiles, minus padding entries) */
    int32_t cus_seen;         /* distinct CUs that ran at least one wave (256 on MI355X) */
    int32_t xcds_seen;
    double span_us;           /* first wave start -> last wave end */
    double cu_busy_imb_pct;   /* per CU: sum of its waves' lifetimes; 100 max/mean - 100   (≙ "Imb") */
    double cu_end_spread_pct; /* 100 (latest CU end - earliest CU end) / span: how long the first idle CU waits for the last */
    double xcd_busy_imb_pct;  /* the same sums per XCD */
    double xcd_end_spread_pct;
    double wave_us_mean, wave_us_max;
} flex_imbalance;
int flex_plan_measure_imbalance(flex_plan *plan, const float *dB, float *dC, flex_stream_t stream, flex_imbalance *out);

/* ≙ Kernel_Info / GPU_Info (flex.cu:4127-4142, 4933-4941: "Kernel %s: %d regs, %zd local, %zd B shared"): what
 * the kernel this plan launches (for 16-byte aligned dense operands) costs per wave and how many waves fit a CU. */
typedef struct flex_kernel_info {
    int32_t vgprs, sgprs;      /* per wave */
    int32_t lds_bytes;         /* static LDS per workgroup */
    int32_t scratch_bytes;     /* per lane */
    int32_t threads_per_block; /* 256: four independent waves */
    int32_t waves_per_cu;      /* resident waves the occupancy calculator allows */
} flex_kernel_info;
int flex_plan_kernel_info(const flex_plan *plan, flex_kernel_info *out);

/* ≙ the tiler round-trip self-check of csr2_DiagTiling (mat.cu:905-940): reads the plan's device image back
 * and verifies that it is a partition of the work -- chunks tile the tasks, tasks tile the records, every
 * record names a valid B row, every C row is written exactly once (directly or by one split row with
 * contiguous pieces), table padding is empty.  FLEX_ERR_FORMAT if any invariant fails.  Debug / test aid:
 * synchronous, O(plan size) host memory. */
int flex_plan_self_check(const flex_plan *plan);

/* What this box's HBM delivers, for the roofline's denominator (verify BW_peak with a
 * device-to-device copy and report both): GB/s of a read-only streaming pass and of a copy (bytes
 * read + bytes written) over `bytes`-sized buffers (use >= 1 GiB: the Infinity Cache is 256 MiB),
 * mean of `reps` launches each.  temporal = 0: non-temporal loads (HBM rate at any size beyond L2);
 * temporal = 1: ordinary loads, so buffers up to the Infinity Cache's size show ITS rate -- the roof
 * of gathers that miss L2 but whose B is cache-resident (reddit: B = 119 MB).  Allocates and frees
 * 2 x bytes on `device`; synchronises. */
int flex_hbm_probe(int device, int64_t bytes, int reps, int temporal, double *read_gbps, double *copy_gbps);

/* ≙ flexspmm_v9_permuteX (flex.cu:276-289): dst[r,:] = src[idx[r],:], n rows of k floats.
 * Not needed by flex_spmm (plans fold the permutation in); provided for callers that
 * keep the reference's B' ("shadow_b") layout. All pointers are device pointers. */
int flex_gather_rows(float *dst, const float *src, const int32_t *idx, int64_t n, int k,
                     flex_stream_t stream);

/* ---- host-side ingest / reordering (C++ in the reference: DataLoader.cu) ---- */

/* Owned host CSR + the statistics DataLoader computes (DataLoader.cu:26-29, 86-115). */
typedef struct flex_host_csr {
    int32_t m, n;
    int64_t nnz;
    uint32_t *rowPtr;
    uint32_t *col;
    float *vals;
    int64_t uni_nb;
    int64_t n_edges_one_way, n_edges_asymmetric;
    int32_t n_nodes_z_out, n_nodes_z_in, n_nodes_z_deg;
    int32_t is_directed;
    int32_t c;
} flex_host_csr;

/* ≙ DataLoader::DataLoader (DataLoader.cu:9-124): 3-line CSV -> CSR (+ amazon.csv rule:
 * no value line, vals = 2*rand()/RAND_MAX-1). */
int flex_csv_load(const char *path, flex_host_csr *out);
void flex_host_csr_free(flex_host_csr *a);

/* ≙ data/SuiteSparse/mtx2csr.cc:57-247 (mmio_allinone): MatrixMarket coordinate file -> CSR.
 * real / integer / pattern (value 1) / complex (real part); `symmetric` and `hermitian` files are
 * expanded to both triangles.  sort_columns = 0 keeps the reference's layout (entries of a row in
 * file order, which the reference's tilers mis-handle: mtx2csr.cc:171-195); 1 sorts each row by
 * column.  m != n is allowed; graph statistics are filled only for square matrices. */
int flex_mtx_load(const char *path, int sort_columns, flex_host_csr *out);

/* ≙ writeCSR2csv (mtx2csr.cc:249-268): the 3-line CSV that DataLoader reads.  Values are written
 * with 9 significant digits so that the file round-trips fp32 exactly (the reference's ofstream
 * default keeps 6). */
int flex_csv_save(const char *path, const flex_csr *A);

/* Binary CSR cache (new; avoids re-parsing GB-sized CSVs): little-endian
 * {magic "FLEXCSR1", int64 m, n, nnz} + rowPtr + col + vals. */
int flex_csr_save_bin(const char *path, const flex_csr *A);
int flex_csr_load_bin(const char *path, flex_host_csr *out);

/* Permutation cache (the reference recomputes every ordering on every run): `rank`
 * (rank[old] = new, n entries) as a small binary file keyed by a 64-bit fingerprint of the CSR
 * structure (flex_csr_fingerprint: n, nnz, rowPtr, col), so a stale file is refused, not applied.
 * flex_perm_load returns FLEX_ERR_IO if the file is absent, FLEX_ERR_FORMAT if it is not a
 * permutation of 0..n-1 or was written for another matrix. */
uint64_t flex_csr_fingerprint(const flex_csr *A);
int flex_perm_save(const char *path, const uint32_t *rank, int64_t n, uint64_t fingerprint);
int flex_perm_load(const char *path, uint32_t *rank, int64_t n, uint64_t fingerprint);

/* ≙ cpuX fill in DataLoader::cuda_alloc_cpy (DataLoader.cu:198-209), glibc rand() stream. */
int flex_fill_dense_rand(float *hostB, int64_t n, int k);

/* ≙ order_rcm(h) (order_rcm.cu:15-33): rank[old] = new. */
int flex_order_rcm(const flex_csr *A, uint32_t *rank);

/* ≙ complete_gorder(h, window) (order_gorder.cu:13-31; DataLoaderGorder uses window 3,
 * DataLoader.cu:808): RCM, then Gorder with the lazy unit heap.  FLEX_ERR_UNSUPPORTED for a graph
 * with an isolated vertex (the reference cannot order one either, unitheap.cu:35-38). */
int flex_order_gorder(const flex_csr *A, uint32_t window, uint32_t *rank);

/* ≙ the ordering of DataLoaderDFS (DataLoader.cu:324-395): depth-first pre-order from vertex 0. */
int flex_order_dfs(const flex_csr *A, uint32_t *rank);
/* ≙ DataLoaderRabbit (DataLoader.cu:455-655) as the reference compiles it (iterative rounds in degree order, no hub
 * grouping): modularity-driven agglomeration + left-to-right walk of the dendrograms.  is_directed = the loader's flag
 * (the clustering then runs on the undirected version).  rank[old] = new; identical to the oracle's restatement.
 * The engine's own community schedule is flex_order_cluster (parallel, built for 10^8 nonzeros). */
int flex_order_rabbit(const flex_csr *A, int is_directed, uint32_t *rank);

/* ≙ order_deg(h, desc) (order_deg.cu:19-45): rank by in+out degree, ties by vertex id. */
int flex_order_deg(const flex_csr *A, int descending, uint32_t *rank);

/* ≙ the clustering half of DataLoaderRabbit (DataLoader.cu:453-655): rank[old] = new such
 * that communities (and their sub-communities) are consecutive.  Two stages (cluster.cpp): modularity-driven agglomeration
 * + a walk of the merge forest, then a few sweeps of vertex moves between stretches of that order (kept only when they
 * put more edges within 2048 positions).  Deterministic: the same rank for any number of host threads. */
int flex_order_cluster(const flex_csr *A, uint32_t *rank);
int flex_order_cluster_ex(const flex_csr *A, const flex_cluster_tuning *tuning /* NULL = rules */, uint32_t *rank);

/* ≙ DataLoaderRcm body (DataLoader.cu:741-779): vo_mp[new]=old + permuted CSR, columns
 * ascending per row. Outputs caller-allocated with the sizes of A. */
int flex_perm_csr(const flex_csr *A, const uint32_t *rank, int32_t *vo_mp, uint32_t *rowPtr2,
                  uint32_t *col2, float *vals2);

/* ---- multi-GPU row sharding (new; the reference is single-GPU, flex.cu:4137) ---- */

/* Contiguous row ranges of about equal cost, cost(row) = nnz(row)*(4k+8) + 4k bytes
 * (gathered B bytes + records + the C row).  Writes nparts+1 row boundaries,
 * row_bounds[0]=0 .. row_bounds[nparts]=m.  Each rank then builds its own plan on
 * rows [row_bounds[r], row_bounds[r+1]) with flex_plan_create_rows. */
int flex_shard_rows(const flex_csr *A, int k, int nparts, int64_t *row_bounds);

/* ---- synthetic graphs with the README's shapes (README.md:13-20); data files are absent ---- */
typedef struct flex_synth_params {
    int64_t n;           /* vertices */
    int64_t nnz;         /* exact nnz incl. one self-loop per row; nnz-n must be even (symmetric) */
    double alpha;        /* power-law exponent of expected degrees (e.g. 2.1) */
    int64_t community;   /* mean planted-community size (0 = none) */
    double p_in;         /* fraction of edges kept inside a community */
    double p_near;       /* fraction of edges that go to one of the 2*near_window neighbouring communities */
    int32_t near_window; /* communities on either side counted as "near" */
    int32_t shuffle;     /* 1: random vertex relabel so the natural order is not banded */
    int32_t gcn_norm;    /* 1: vals = 1/sqrt(d_i d_j) (pubmed-like); 0: U(-1,1) */
    int32_t directed;    /* 1: every edge kept in ONE random direction, no self loops, so nnz = #edges and
                            rows may be empty (the SuiteSparse stand-ins); 0: symmetric + self loops */
    uint64_t seed;
} flex_synth_params;
int flex_synth_graph(const flex_synth_params *p, flex_host_csr *out);
/* Parameters of the stand-in for a named graph of the README / the SuiteSparse sweep
 * ("pubmed","flickr","reddit","ppi","yelp","amazon","wiki-vote","soc-sign-epinions"),
 * scaled to `scale` x vertices and nonzeros (weak-scaling runs). */
int flex_synth_preset(const char *name, int scale, flex_synth_params *out);

const char *flex_strerror(int status);
/* hipError_t of the last failed HIP call on this thread (0 if none) and its text. */
int flex_last_hip_error(void);
const char *flex_last_hip_error_string(void);
int flex_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FLEX_SPMM_H */
