// internal.h -- shared between the host planner and the HIP kernels of libflex_spmm.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>

#include "../../include/flex_spmm.h"

namespace flex {

// One split row: C[row,:] = sum of partial[first .. first+count) in that order.
struct SplitRow {
    uint32_t row, first, count;
};

// What the SpMM kernels read.  Passed by value as a kernel argument (the reference
// copies a ~270-byte Mat_POD into __constant__ memory instead: mat.cuh:18-65, mat.cu:32-41).
struct PlanView {
    const uint2 *rec;        // [nnz] {x = B-row byte offset (off32) or column id, y = value bits}, task order
    const uint32_t *t_beg;   // [n_tasks+1] first record of each task
    const uint32_t *t_dst;   // [n_tasks]   C row written by the task; MSB set -> partial slot id (the task is a PIECE), or, with
                             //             kBundleFlag as well, a BUNDLE: low bits = its first entry in the chunk's part of bd_rows
    const uint2 *t_aux;      // [n_tasks]   pieces: {index into `split` of the piece's row, #pieces of that row}; bundles: {first entry
                             //             in bd_rows, steps}; else {0,0}
    const uint4 *chunk;      // [n_chunks] {first task, #tasks (<= 63), first record, end record}; one wave per chunk
    float *partial;          // [n_partials][k] partial sums of split rows
    const SplitRow *split;   // [n_split] {C row, first partial, #pieces}
    uint32_t *split_cnt;     // [n_split][k-tiles] arrival counters, zero between launches
    uint32_t fused_fixup;    // 1: the last piece to finish sums the row inside the launch; 0: spmm_fixup_kernel does
    uint32_t n_chunks;
    int32_t k;
    int32_t ldb, ldc;        // floats between consecutive rows of B and of C (>= k; == k for dense operands)
    uint32_t xcd_remap;      // 1: remap workgroup ids so each XCD walks one contiguous slice of the schedule
    uint32_t lds_extra;      // bytes of unused dynamic LDS per workgroup (occupancy throttle, tuning only)
    uint32_t rec_nt;         // 1: the record stream is read with non-temporal loads
    uint32_t tile_group;     // 0: column tiles are the slow grid dimension (one pass over all chunks per tile).  Else: workgroups per group --
                             // every XCD slice of the chunk table is walked group by group, all column tiles of a group back to back, so that
                             // a group's records are re-read while they are still in the Infinity Cache (1-D grid of n_workgroups x tiles)
    uint64_t *trace;         // flex_plan_measure_imbalance: 3 words per (k-tile, chunk-table entry); diagnostic -DFLEX_TRACE builds: 12 per wave; else nullptr
    // Row bundles (plan_build.cpp, form_tasks): a task that holds up to S = 64 / G SHORT rows side by side, slot s of every step working
    // on row s -- no cross-slot reduction, one 16-byte store per lane at the end.  nullptr when the plan has none.
    const uint32_t *bd_rows; // per bundle S entries: C row of slot s | kBundleZero (the row holds no nonzero: zeros are stored), or kBundleNoRow
    const uint2 *chunk_bd;   // [chunk-table entries] {first entry in bd_rows, entries (a multiple of S, <= kBundleRowsPerChunk)} of the chunk's bundles
};


// What the dense-tile kernel reads (tile_kernels.hip): 32x32 blocks of A that left the record stream.
struct TileView {
    const float *a;            // [n_tiles][4][64][4] tile values in MFMA A-operand order: (q,lane,e) = A[lane&31][2(4q+e) + (lane>>5)]
    const uint32_t *boff;      // [n_tiles][32] B row of each tile column: byte offset (off32) or row id
    const uint32_t *mask;      // [n_tiles][32] bit j of word i: the tile holds an entry at (row i, column j) -- an explicit zero counts, an absent one does not
    const uint32_t *rt_ptr;    // [n_row_tiles+1] tiles of each listed row tile, in column order
    const uint32_t *rt_rows;   // [n_row_tiles][32] C row of each row of the row tile, 0xFFFFFFFF = none
    uint32_t n_row_tiles;
};

// ---- the hot-block path (block_kernels.hip, block_plan.cpp): LDS-level reuse of B for the nonzeros that have it.
// Round 4 design.  The matrix is SPLIT: a nonzero whose column is used by at least `thr` nonzeros of its BLOCK (R = rounds x 60
// schedule-consecutive rows) is HOT and lives in the block image below; every other nonzero -- and every row too long for a
// slot -- stays in the flat plan, whose kernel is the one that moves L2 misses at the fabric's rate.  flex_spmm runs the flat
// kernel first (it writes every row of C), then spmm_hot_kernel ADDS the hot part: one workgroup of 16 waves per (block, 64-column
// tile): 15 CONSUMER waves of 4 slots x 16 lanes + 1 LOADER wave.  A slot holds ONE C row per round in registers (the lane owns 4
// of the tile's 64 columns); the hot B rows (256 bytes per row and tile) are staged panel by panel into LDS by the loader wave
// (LDS-DMA, double-buffered) and every use is a ds_read_b128 that is bank-conflict free by construction (a 16-lane slot reads one
// whole 256-byte row = all 64 banks).  Records never touch LDS: a RUN (the <= 16 steps of one (wave, panel, round)) is one coalesced
// 512-byte load into a register pair a whole panel ahead, and step j's record reaches the 16 lanes of its slot by a DPP row
// broadcast (row_newbcast:j) -- the LDS pipe carries nothing but B.
#ifndef FLEX_BK_WAVES
#define FLEX_BK_WAVES 15
#endif
constexpr int kBkWaves = FLEX_BK_WAVES;                   // consumer waves per workgroup
constexpr int kBkSlots = 4;                               // slots per wave (16 lanes x float4 = one 64-column tile of one row)
constexpr int kBkTileCols = 64;                           // columns of C per pass
constexpr int kBkRowsPerRound = kBkWaves * kBkSlots;      // 60 row slots per round
constexpr int kBkMaxRounds = 8;
constexpr uint32_t kBkRunMax = 16;                        // steps of one run = lanes of a slot: what one DPP row holds
#ifndef FLEX_BK_NBUF       // experiment builds (make -C flex_amd/csrc block_variants) may vary the two; the product has one pair
#define FLEX_BK_NBUF 2     // measured (profiles/r04_hot_block_ring_probe.txt): three buffers of 200 rows with the loader two panels ahead are
#endif                     // SLOWER than two of 304 -- more panels mean more runs and barriers, and those, not staging latency, are what the kernel pays for
#ifndef FLEX_BK_PANEL_MAX
#define FLEX_BK_PANEL_MAX (FLEX_BK_NBUF == 2 ? 304 : 200)
#endif
constexpr uint32_t kBkNBuf = FLEX_BK_NBUF;                // panel buffers: the loader runs kBkNBuf - 1 panels ahead of the consumers
constexpr uint32_t kBkPanelMax = FLEX_BK_PANEL_MAX;       // B rows per LDS panel
constexpr uint32_t kBkRowBytes = 256;                     // one B row of one column tile
constexpr uint32_t kBkZeroRow = kBkPanelMax * kBkRowBytes; // byte offset, inside a panel buffer, of a row of zeros (padding records point at it)
constexpr uint32_t kBkBufBytes = kBkZeroRow + kBkRowBytes;
constexpr uint32_t kBkLdsHcol = kBkNBuf * kBkBufBytes;    // two scratch slots for the byte offsets of the panels about to be staged
constexpr uint32_t kBkLdsBytes = kBkLdsHcol + 2 * kBkPanelMax * 4;  // 158 592 of the CU's 163 840 (two buffers of 304 rows)
static_assert(kBkLdsBytes <= 163840 && kBkPanelMax % 4 == 0 && kBkPanelMax <= 256 + 48 && (kBkNBuf == 2 || kBkNBuf == 3), "the hot kernel's LDS image must fit one CU");
constexpr uint32_t kBkLdsNext = kBkMaxRounds * kBkRowsPerRound * kBkRowBytes;  // after the last panel: [slots] sums of later parts, then [slots] next part + 1
static_assert(kBkLdsNext + kBkMaxRounds * kBkRowsPerRound * 4 <= kBkLdsHcol, "the parts of long rows meet in the panel buffers");
constexpr uint32_t kBkEmptyRow = 0xFFFFFFFFu;             // brow entry of a slot that holds no row
constexpr uint32_t kBkMaxPanels = 63;                     // a wave holds its run counts one panel per lane (and looks one panel ahead)
// A long row holds several slots (its PARTS); they meet in LDS after the last panel.  link[slot]: bits 0-15 = 1 + the slot of the
// row's next part (0 = none), in the block's [round][wave][slot] numbering.
constexpr uint32_t kBkLinkOwner = 0x40000000u;            // this slot collects the chain that starts at its `next` and writes the row
constexpr uint32_t kBkLinkPart = 0x80000000u;             // this slot publishes its sum (and its `next`) for the owner

struct BlockView {
    const uint4 *hdr;        // [n_blocks] {panels | (some row has several parts) << 31, first entry in hcol, first word in cnt, words of cnt per wave (2 x panels)}
    const uint2 *wstart;     // [n_blocks][15] {first step of the wave's record stream, its steps}
    const uint32_t *cnt;     // per (block, wave, panel): a 64-bit word (two u32), byte r = the steps (<= 16) of the run (panel, round r)
    const uint32_t *hcol;    // per (block, panel): panel_rows byte offsets of the B rows staged (padded with a valid one)
    const uint32_t *brow;    // [n_blocks][rounds][15][4] C row the slot reads and writes (an owner); kBkEmptyRow = none (empty, or a later part of a row)
    const uint32_t *link;    // [n_blocks][rounds][15][4] kBkLinkOwner / kBkLinkPart | next part + 1; 0 for a row of one part
    const uint2 *rec;        // [steps][4] {byte offset inside the panel buffer, value bits}
    uint64_t n_rec;          // records in `rec` (loads past a wave's stream are clamped to the last one)
    uint32_t n_blocks, rounds, panel_rows;
    int32_t k, ldb, ldc;
    uint32_t xcd_remap;
    uint32_t ablate;         // timing-only (tools/probe_blocks.py; results are WRONG): 1 no panel DMA, 2 no panel work, 16 no read-modify-write of C, 32 every record load hits one line
    uint64_t *trace;         // -DFLEX_TRACE builds only (tools/trace_blocks.py): 8 cycle counters per (tile, block, wave); else nullptr
};

constexpr uint32_t kPartialFlag = 0x80000000u;
constexpr uint32_t kBundleFlag = 0x40000000u;       // in t_dst, together with kPartialFlag: the task is a bundle of rows
constexpr uint32_t kBundleZero = 0x80000000u;       // in bd_rows: store zeros (a row without nonzeros)
constexpr uint32_t kBundleNoRow = 0xFFFFFFFFu;      // in bd_rows: the slot holds no row
constexpr uint32_t kBundleRowsPerChunk = 128;       // a wave keeps its chunk's bd_rows entries in two registers per lane
constexpr uint32_t kBundleMinSlots = 4;             // bundles only on tiles with at least this many record slots per step (G <= 16)
constexpr int kWavesPerBlock = 4;  // 256-thread workgroups
constexpr int kXcds = 8;           // MI355X: 8 XCDs, each with a private 4 MiB L2

// per-thread record of the last HIP failure (flex_last_hip_error)
void note_hip_error(hipError_t e);

#define FLEX_HIP_TRY(expr)                          \
    do {                                            \
        hipError_t e_ = (expr);                     \
        if (e_ != hipSuccess) {                     \
            ::flex::note_hip_error(e_);             \
            return FLEX_ERR_HIP;                    \
        }                                           \
    } while (0)

// kernel launchers (spmm_kernels.hip)
int launch_spmm(const PlanView &v, int lanes_per_nz, bool off32, bool vec4, const float *dB, float *dC,
                hipStream_t s, int unroll = 0);
int launch_spmm_stamped(const PlanView &v, int lanes_per_nz, bool off32, const float *dB, float *dC, hipStream_t s);
int launch_fixup(const float *partial, const SplitRow *rows, uint32_t n_rows, int k, int ldc, float *dC,
                 hipStream_t s);
int kernel_attributes(int lanes_per_nz, bool off32, bool vec4, hipFuncAttributes *attr, int *waves_per_cu);
int launch_gather_rows(float *dst, const float *src, const int32_t *idx, int64_t n, int k, hipStream_t s);
int launch_tiles(const TileView &tv, bool off32, const float *dB, float *dC, int k, int ldb, int ldc, hipStream_t s);
int launch_blocks(const BlockView &bv, const float *dB, float *dC, hipStream_t s, bool vec4 = true);

// FLEX_PLAN_TIMING in the environment: phase times of the planner and the clustering on stderr.  The only environment
// variable the library reads; every tuning knob is a field of flex_plan_tuning (include/flex_spmm.h).
bool plan_timing_enabled();

// host-side helpers shared by the ABI files
int order_rcm_host(int64_t n, const uint32_t *rowPtr, const uint32_t *col, std::vector<uint32_t> &rank);
int order_cluster_host(int64_t n, const uint32_t *rowPtr, const uint32_t *col, std::vector<uint32_t> &rank,
                       const flex_cluster_tuning *tuning = nullptr);
int order_gorder_host(int64_t n, const uint32_t *rowPtr, const uint32_t *col, uint32_t window,
                      std::vector<uint32_t> &rank);
int validate_csr(const flex_csr *A);

}  // namespace flex
